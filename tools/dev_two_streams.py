#!/usr/bin/env python3
"""Dev harness (GPU box): does the aggregation gain from running the left-volume chain and the right-volume chain as
two independent streams of single-volume launches (one launch's tail under the other's body) instead of one
two-volume launch per iteration?  Kernels assembled from csrc/asm/cbca_prog_gen.py with the library's parameters,
programs from the library's builder.   python tools/dev_two_streams.py [--config cfg2] [--iters 16] [--skip]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", "mc-cnn-python_amd/csrc/asm", "tests/asmtools", "tools", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
import cbca_prog_gen as gen
import cbca_prog_ref as ref
import dev_prog_check as dpc
from bench import CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pipe", type=int, default=0, help="the program-managed ring window (widest unit); programs from the Python builder")
    ap.add_argument("--w", type=int, default=20); ap.add_argument("--minvgpr", type=int, default=0)
    args = ap.parse_args()
    hip.require_device()
    H, W, D = CONFIGS[args.config]
    Dp = sd.hwd_pitch(D)
    vpl = 2 if Dp <= 128 else 3 if (Dp <= 192 and Dp % 3 == 0) else 4
    outdir = os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm")
    mods = {}
    for skip in (False, True):
        g = gen.Gen(gen.Params(vpl=vpl, K=4, W=args.w, skip=skip, pipe=args.pipe, minvgpr=args.minvgpr)).build()
        base = os.path.join(outdir, "two_streams_v%d_%d_%d_%d_%d" % (vpl, int(skip), args.w, args.pipe, args.minvgpr))
        open(base + ".s", "w").write(g.render())
        import subprocess
        subprocess.check_call([dpc.LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                               base + ".s", "-o", base + ".o"])
        subprocess.check_call([dpc.LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
        mods[skip] = dpc.Module(base + ".hsaco", g.P.name())
        L = g.layout()
    Li, Ri, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(Li[:, :, 0]).cuda(), torch.from_numpy(Ri[:, :, 0]).cuda()
    sl, sr = sd.cross_arms_pair(dl, dr, 0.02, 14)
    progs = sd.cbca_prog_buffers(D, H, W, dl.device)
    sd.cbca_prog_build_pair(sl, sr, D, 14, progs)
    set_dwords = progs[0].numel() // 2
    meta = dict(band_rows=ref.band_rows_of(H, 4), ngroups=-(-W // 5), stride=ref.prog_stride_dwords(L))
    meta["band_groups"] = meta["band_rows"] // 4
    nchunks = -(-Dp // (64 * vpl))
    gt = torch.Generator(device="cuda").manual_seed(0)
    a = -torch.rand((H, W, Dp), device="cuda", generator=gt); b = torch.empty_like(a)
    c = a.flip(0).contiguous(); d = torch.empty_like(a)
    want = sd.cbca_prog_pair(a.clone(), torch.empty_like(a), sl, c.clone(), torch.empty_like(c), sr, progs, D, args.iters, 14)
    want_l, want_r = want[0][0].clone(), want[1][0].clone()
    pf = [progs[0][:set_dwords], progs[1][:set_dwords]]
    ps = [progs[0][set_dwords:], progs[1][set_dwords:]]
    if args.pipe or args.w != 20:      # another program format: the plain-Python builder
        Lp = dict(L, pix=4 * Dp)
        pf, ps = [], []
        for sup in (sl, sr):
            sup0 = sup.cpu().numpy().view(np.uint32).reshape(-1)[:H * W].reshape(H, W)
            for lst, sk in ((pf, False), (ps, True)):
                arr, m2 = ref.build_all(sup0, H, W, Lp, skip_unit=sk)
                lst.append(torch.from_numpy(arr.view(np.int32)).cuda())
        meta["stride"] = m2["stride"]
    grid2 = (8 * meta["band_groups"], meta["ngroups"], nchunks * 2)
    grid1 = (8 * meta["band_groups"], meta["ngroups"], nchunks)

    def is_skip(it):
        n = args.iters
        return it >= 1 and not (n % 2 == 0 and it == n - 1)

    def run_pair(x0, y0, x1, y1):
        for it in range(args.iters):
            sk = is_skip(it)
            pp = ps if sk else pf
            mods[sk].launch(grid2, dpc.kargs([x0, x1], [y0, y1], pp, [sl, sr], Dp, H, W, nchunks, meta))
            x0, y0, x1, y1 = y0, x0, y1, x1
        return x0, x1

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run_split(x0, y0, x1, y1):
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        for st, (x, y, sup, k) in ((s1, (x0, y0, sl, 0)), (s2, (x1, y1, sr, 1))):
            with torch.cuda.stream(st):
                for it in range(args.iters):
                    sk = is_skip(it)
                    pp = ps if sk else pf
                    mods[sk].launch(grid1, dpc.kargs([x, x], [y, y], [pp[k], pp[k]], [sup, sup], Dp, H, W, nchunks, meta))
                    x, y = y, x
            if k == 0:
                r0 = x
            else:
                r1 = x
        main.wait_stream(s1); main.wait_stream(s2)
        return r0, r1

    for name, fn in (("one two-volume launch per iteration", run_pair), ("two streams of single-volume launches", run_split)):
        r0, r1 = fn(a.clone(), b, c.clone(), d)
        torch.cuda.synchronize()
        ok = torch.equal(r0, want_l) and torch.equal(r1, want_r)
        ts = []
        for _ in range(args.reps):
            x0, x1 = a.clone(), c.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(x0, b, x1, d); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-42s %d iterations: %.4f ms (min %.4f) = %.4f ms per iteration   bit-identical to cbca_prog_pair: %s"
              % (name, args.iters, float(np.median(ts)), min(ts), float(np.median(ts)) / args.iters, ok), flush=True)


if __name__ == "__main__":
    main()

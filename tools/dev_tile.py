#!/usr/bin/env python3
"""Dev harness (GPU box), round 6: the aggregation kernel with the region rows shared through LDS by the NW waves of a
workgroup (csrc/asm/cbca_prog_gen.py --tile NW; programs from tests/asmtools/cbca_prog_ref.build_all_tiles) against the
shipped program-driven kernel: bit-identity and the time of match.py's 16-iteration aggregation as two chains of
one-volume launches on two streams (full, 14 x skip, full).
    python tools/dev_tile.py [--config cfg2] [--iters 16] [--nw 4] [--small-only]"""
import argparse
import os
import subprocess
import sys
import time

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


DEBUG = 0


def assemble(vpl, nw, skip, outdir):
    g = gen.Gen(gen.Params(vpl=vpl, K=4, W=20, skip=skip, tile=nw, debug=DEBUG)).build()
    base = os.path.join(outdir, "tile%d_v%d_%d_g%d" % (nw, vpl, int(skip), DEBUG))
    os.makedirs(outdir, exist_ok=True)
    open(base + ".s", "w").write(g.render())
    subprocess.check_call([dpc.LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                           base + ".s", "-o", base + ".o"])
    subprocess.check_call([dpc.LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
    return g, dpc.Module(base + ".hsaco", g.P.name())


def launch(mod, grid, block, karg_bytes):
    import ctypes
    buf = ctypes.create_string_buffer(karg_bytes, len(karg_bytes))
    size = ctypes.c_size_t(len(karg_bytes))
    extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p).value, 2,
                                   ctypes.cast(ctypes.pointer(size), ctypes.c_void_p).value, 3)
    rc = mod.hip.hipModuleLaunchKernel(mod.fn, grid[0], grid[1], grid[2], block, 1, 1, 0,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
    assert rc == 0, "hipModuleLaunchKernel -> %d" % rc


def tile_programs(sup, H, W, L, Dp):
    sup0 = sup.cpu().numpy().view(np.uint32).reshape(-1)[:H * W].reshape(H, W)
    out = []
    meta = None
    for sk in (False, True):
        t = time.time()
        arr, meta = ref.build_all_tiles(sup0, H, W, dict(L, pix=4 * Dp), skip_unit=sk)
        print("   programs (%s): %.1f s, %d steps, %d LDS slots, %d window units, longest %d of %d words"
              % ("skip" if sk else "full", time.time() - t, meta["steps"], meta["slots"], meta["units"], meta["longest"],
                 meta["stride"]), flush=True)
        out.append(torch.from_numpy(arr.view(np.int32)).cuda())
    return out, meta


def check_small(mods, L, nw, H, W, D, seed, n_iter=3, vpl=4):
    """n_iter iterations (full, skip.., full when even) of the tile kernels against cbca_prog_pair on a small shape."""
    Li = synthetic.make_pair(H, W, max(1, min(D, W - 2)), seed=seed)[0]
    img = torch.from_numpy(Li[:, :, 0]).cuda()
    sup = sd.cross_arms(img, 0.02, 14)
    Dp = sd.hwd_pitch(D)
    gt = torch.Generator(device="cuda").manual_seed(seed)
    a = (torch.rand((H, W, Dp), device="cuda", generator=gt) * 3 - 2).contiguous()
    progs = sd.cbca_prog_buffers(D, H, W, img.device)
    sd.cbca_prog_build_pair(sup, sup, D, 14, progs)
    (want, _), _ = sd.cbca_prog_pair(a.clone(), torch.full_like(a, float("nan")), sup, a.clone(), torch.empty_like(a), sup, progs,
                                     D, n_iter, 14)
    torch.cuda.synchronize()
    print("   reference done", flush=True)
    (pf, ps), meta = tile_programs(sup, H, W, L, Dp)
    nchunks = -(-Dp // (64 * vpl))
    grid = (8 * meta["band_groups"], meta["ntx"], nchunks)
    x, y = a.clone(), torch.full_like(a, float("nan"))
    for it in range(n_iter):
        sk = it >= 1 and not (n_iter % 2 == 0 and it == n_iter - 1)
        pp = ps if sk else pf
        launch(mods[sk], grid, 64 * nw, dpc.kargs([x, x], [y, y], [pp, pp], [sup, sup], Dp, H, W, nchunks, meta))
        torch.cuda.synchronize()
        print("   iteration %d (%s) done" % (it, "skip" if sk else "full"), flush=True)
        x, y = y, x
    torch.cuda.synchronize()
    ok = torch.equal(x[:, :, :D].view(torch.int32), want[:, :, :D].view(torch.int32))
    print("shape %dx%dx%d seed %d, %d iterations: %s" % (W, H, D, seed, n_iter, "bit-identical" if ok else "DIFFERENT"), flush=True)
    if not ok:
        d = (x[:, :, :D] != want[:, :, :D]) & ~(torch.isnan(x[:, :, :D]) & torch.isnan(want[:, :, :D]))
        idx = d.nonzero()[:6].tolist()
        print("   mismatching voxels: %d; first (y,x,d):" % int(d.sum()), idx, [float(x[tuple(i)]) for i in idx],
              [float(want[tuple(i)]) for i in idx])
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10); ap.add_argument("--nw", type=int, default=4)
    ap.add_argument("--small-only", action="store_true")
    ap.add_argument("--debug", type=int, default=0, help="generator debug bits (32: no LDS requests, 64: no LDS reads)")
    ap.add_argument("--vpl", type=int, default=4, help="disparities per lane (2: 128-disparity chunks, 512-byte LDS slots, 86 VGPRs)")
    args = ap.parse_args()
    global DEBUG
    DEBUG = args.debug
    hip.require_device()
    H, W, D = CONFIGS[args.config]
    Dp = sd.hwd_pitch(D)
    VPL = args.vpl
    assert Dp % VPL == 0 and VPL in (2, 4)
    outdir = os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm")
    mods = {}
    for skip in (False, True):
        g, mods[skip] = assemble(VPL, args.nw, skip, outdir)
        L = g.layout()
    print("tile kernel: %d waves per workgroup, %d VGPRs, %d bytes of LDS, %d bytes of code" % (args.nw, g.P.nvgpr, g.P.LDS_BYTES,
                                                                                               L["code_bytes"]), flush=True)
    ok = True
    for (h, w, d, seed, n) in ((24, 32, 256, 1, 3), (37, 61, 256, 2, 4), (40, 203, 300, 3, 5), (9, 44, 256, 4, 1)):
        ok &= check_small(mods, L, args.nw, h, w, d, seed, n, VPL)
    if args.small_only:
        print("ALL OK" if ok else "FAILED")
        sys.exit(0 if ok else 1)

    Li, Ri, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(Li[:, :, 0]).cuda(), torch.from_numpy(Ri[:, :, 0]).cuda()
    sl, sr = sd.cross_arms_pair(dl, dr, 0.02, 14)
    progs = sd.cbca_prog_buffers(D, H, W, dl.device)
    sd.cbca_prog_build_pair(sl, sr, D, 14, progs)
    gt = torch.Generator(device="cuda").manual_seed(0)
    a = -torch.rand((H, W, Dp), device="cuda", generator=gt); b = torch.empty_like(a)
    c = a.flip(0).contiguous(); d = torch.empty_like(a)
    n = args.iters
    tprogs, meta = [], None
    for sup in (sl, sr):
        pp, meta = tile_programs(sup, H, W, L, Dp)
        tprogs.append(pp)
    nchunks = -(-Dp // (64 * VPL))
    grid = (8 * meta["band_groups"], meta["ntx"], nchunks)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def is_skip(it, skipping):
        return skipping and it >= 1 and not (n % 2 == 0 and it == n - 1)

    def chains(kind, skipping, x0, y0, x1, y1):
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        res = [None, None]
        for st, (x, y, sup, k) in ((s1, (x0, y0, sl, 0)), (s2, (x1, y1, sr, 1))):
            with torch.cuda.stream(st):
                if kind == "prog":
                    x, y = sd.cbca_prog_chain(x, y, sup, progs[k], D, n, 14, skip_unit_regions=skipping)
                else:
                    for it in range(n):
                        sk = is_skip(it, skipping)
                        pp = tprogs[k][1 if sk else 0]
                        launch(mods[sk], grid, 64 * args.nw, dpc.kargs([x, x], [y, y], [pp, pp], [sup, sup], Dp, H, W, nchunks, meta))
                        x, y = y, x
                res[k] = x
        main.wait_stream(s1); main.wait_stream(s2)
        return res

    for skipping in (True, False):
        want = chains("prog", skipping, a.clone(), b, c.clone(), d)
        want = [w.clone() for w in want]
        for kind in ("prog", "tile"):
            r = chains(kind, skipping, a.clone(), b, c.clone(), d)
            torch.cuda.synchronize()
            same = all(torch.equal(r[i].view(torch.int32), want[i].view(torch.int32)) for i in range(2))
            ts = []
            for _ in range(args.reps):
                x0, x1 = a.clone(), c.clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); chains(kind, skipping, x0, b, x1, d); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print("%-28s %s  %d iterations on two streams: %.4f ms (min %.4f) = %.4f ms per iteration   bit-identical: %s"
                  % ("shipped patch kernel" if kind == "prog" else "tile kernel (%d waves)" % args.nw,
                     "full, 14 x skip, full" if skipping else "every iteration full ", n, float(np.median(ts)), min(ts),
                     float(np.median(ts)) / n, same), flush=True)


if __name__ == "__main__":
    main()

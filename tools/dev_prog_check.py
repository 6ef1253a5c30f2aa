#!/usr/bin/env python3
"""Dev harness (GPU box): launches the assembled program-driven aggregation kernel (csrc/asm/cbca_prog_gen.py) straight
through the HIP module API with programs built by the plain-Python builder (tests/asmtools/cbca_prog_ref.py), checks it
bit for bit against cbca_hwd_kernel and times both.
    python tools/dev_prog_check.py [--config cfg2] [--iters 10] [--w 12] [--small-only]"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", "mc-cnn-python_amd/csrc/asm", "tests/asmtools", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
import cbca_prog_gen as gen
import cbca_prog_ref as ref
from bench import CONFIGS

LLVM = "/opt/rocm/lib/llvm/bin"
K_DEFAULT = 2


RING = 0
MINVGPR = 0
PERSIST = False
PIPE = 0
NTLOAD = False
EARLY = False


def build_hsaco(vpl, w, outdir, nb=1, debug=0, pf=0, k=None):
    k = k or K_DEFAULT
    os.makedirs(outdir, exist_ok=True)
    g = gen.Gen(gen.Params(vpl=vpl, K=k, W=w, NB=nb, debug=debug, PF=pf, ring=RING, minvgpr=MINVGPR, persist=PERSIST, pipe=PIPE, ntload=NTLOAD, early=EARLY)).build()
    base = os.path.join(outdir, "cbca_prog_v%d_k%d_w%d_b%d_g%d_p%d_r%d_m%d_s%d_q%d_n%d_e%d" % (vpl, k, w, nb, debug, pf, RING, MINVGPR, int(PERSIST), PIPE, int(NTLOAD), int(EARLY)))
    if not os.path.exists(base + ".hsaco") or os.path.getmtime(base + ".hsaco") < os.path.getmtime(gen.__file__):
        open(base + ".s", "w").write(g.render())
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                               base + ".s", "-o", base + ".o"])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
    return g, base + ".hsaco"


class Module:
    def __init__(self, path, name):
        self.hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        self.mod = ctypes.c_void_p()
        rc = self.hip.hipModuleLoad(ctypes.byref(self.mod), path.encode())
        assert rc == 0, "hipModuleLoad -> %d" % rc
        self.fn = ctypes.c_void_p()
        rc = self.hip.hipModuleGetFunction(ctypes.byref(self.fn), self.mod, name.encode())
        assert rc == 0, "hipModuleGetFunction -> %d" % rc

    def launch(self, grid, karg_bytes, lds=0):
        buf = ctypes.create_string_buffer(karg_bytes, len(karg_bytes))
        size = ctypes.c_size_t(len(karg_bytes))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p).value, 2,
                                       ctypes.cast(ctypes.pointer(size), ctypes.c_void_p).value, 3)
        rc = self.hip.hipModuleLaunchKernel(self.fn, grid[0], grid[1], grid[2], 64, 1, 1, lds,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
        assert rc == 0, "hipModuleLaunchKernel -> %d" % rc


def grid_of(meta, nz):
    """The launch grid: one wave per patch and (image, chunk), or (persistent kernels) 8 x NWC x band_groups waves."""
    if not PERSIST:
        return (8 * meta["band_groups"], meta["ngroups"], nz)
    return (8 * nwc_of(meta) * meta["band_groups"], 1, 1)


def nwc_of(meta):
    per_cu = max(1, min(12, 160 // max(RING, 1)))          # waves a CU holds: LDS-limited
    return max(1, (32 * per_cu) // meta["band_groups"])


def kargs(ins, outs, progs, sups, Dp, H, W, nchunks, meta):
    k = np.zeros((0x80 if PERSIST else 0x60) // 4, np.uint32)
    if PERSIST:
        k[24] = nwc_of(meta)
    for i, t in enumerate(ins + outs + progs + sups):
        a = t.data_ptr()
        k[2 * i], k[2 * i + 1] = a & 0xffffffff, a >> 32
    k[16:24] = [Dp, H, W, nchunks, meta["band_rows"], meta["band_groups"], meta["stride"] * 4, meta["ngroups"]]
    return k.tobytes()


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def programs_for(sup, H, W, L, Dp=0):
    sup0 = sup.cpu().numpy().view(np.uint32).reshape(H, W)
    t = time.time()
    progs, meta = ref.build_all(sup0, H, W, dict(L, pix=4 * Dp))
    meta["build_s"] = time.time() - t
    return torch.from_numpy(progs.view(np.int32)).cuda(), meta


def check_shape(mod, g, L, H, W, D, seed, flat=False):
    P = g.P
    if flat:
        img = torch.zeros((H, W), device="cuda")
    else:
        Li = synthetic.make_pair(H, W, min(D, W - 2), seed=seed)[0]
        img = torch.from_numpy(Li[:, :, 0]).cuda()
    sup = sd.cross_arms(img, 0.02, 14)
    Dp = sd.hwd_pitch(D)
    gt = torch.Generator(device="cuda").manual_seed(seed)
    a = (torch.rand((H, W, Dp), device="cuda", generator=gt) * 3 - 2).contiguous()
    want, _ = sd.cbca_hwd(a, torch.full_like(a, float("nan")), sup, D, 1, 14)
    progs, meta = programs_for(sup, H, W, L, Dp)
    out = torch.full_like(a, float("nan"))
    nchunks = -(-Dp // (64 * P.VPL))
    mod.launch(grid_of(meta, nchunks), kargs([a, a], [out, out], [progs, progs], [sup, sup], Dp, H, W, nchunks, meta))
    torch.cuda.synchronize()
    ok = torch.equal(out[:, :, :D].nan_to_num(777.), want[:, :, :D].nan_to_num(777.))
    print("shape %dx%dx%d seed %d flat %d: %s  (longest program %d of %d)" % (H, W, D, seed, flat, ok, meta["longest"],
                                                                             meta["stride"]), flush=True)
    if not ok:
        d = (out[:, :, :D] != want[:, :, :D])
        idx = d.nonzero()[:5].tolist()
        print("   first mismatches (y,x,d):", idx, [float(out[tuple(i)]) for i in idx], [float(want[tuple(i)]) for i in idx])
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--w", type=int, default=12); ap.add_argument("--small-only", action="store_true")
    ap.add_argument("--nb", type=int, default=1); ap.add_argument("--skip-small", action="store_true")
    ap.add_argument("--experiments", action="store_true"); ap.add_argument("--pf", default="")
    ap.add_argument("--k", type=int, default=2); ap.add_argument("--order", type=int, default=0)
    ap.add_argument("--ring", type=int, default=0, help="experimental: window rows through an LDS ring of this many slots")
    ap.add_argument("--minvgpr", type=int, default=0); ap.add_argument("--persist", action="store_true")
    ap.add_argument("--pipe", type=int, default=0, help="the window is a program-managed ring of --w slots; widest unit")
    ap.add_argument("--ntload", action="store_true", help="non-temporal loads for rows of unit-region pixels")
    ap.add_argument("--vpl", type=int, default=0, help="disparities per lane of the full-size run (0: the library's rule)")
    ap.add_argument("--drop", default="", help="fl,fa: time ONLY the variant with these fractions of LOAD / ADD ops dropped "
                                               "(counter passes under rocprofv3: every launch of the run is that variant)")
    ap.add_argument("--early", action="store_true", help="experimental: the program's first op dispatched from a scalar load")
    args = ap.parse_args()
    global K_DEFAULT, RING, MINVGPR, PERSIST, PIPE, NTLOAD, EARLY
    K_DEFAULT, RING, MINVGPR, PERSIST, PIPE, NTLOAD = args.k, args.ring, args.minvgpr, args.persist, args.pipe, args.ntload
    EARLY = args.early
    hip.require_device()
    allok = True
    for vpl in (() if args.skip_small else (4,) if RING else (4, 2, 3)):   # (no 8-byte buffer_load ... lds on gfx950)
        g, path = build_hsaco(vpl, args.w, os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm"), args.nb)
        mod = Module(path, g.P.name())
        L = g.layout()
        shapes = {4: [(24, 32, 8, 0), (37, 61, 200, 2), (20, 300, 256, 6), (30, 47, 400, 7), (64, 130, 250, 3)],
                  2: [(40, 48, 16, 1), (50, 70, 2, 4), (33, 45, 128, 5)], 3: [(28, 66, 192, 8), (31, 50, 96, 9)]}[vpl]
        for (H, W, D, seed) in shapes:
            assert sd.hwd_pitch(D) % vpl == 0
            allok &= check_shape(mod, g, L, H, W, D, seed)
        if vpl == 4:
            allok &= check_shape(mod, g, L, 60, 80, 20, 0, flat=True)
    print("ALL OK" if allok else "MISMATCH", flush=True)
    if args.small_only:
        sys.exit(0 if allok else 1)
    H, W, D = CONFIGS[args.config]
    Dp = sd.hwd_pitch(D)
    vpl = args.vpl or (2 if Dp <= 128 else 3 if (Dp <= 192 and Dp % 3 == 0) else 4)
    g, path = build_hsaco(vpl, args.w, os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm"), args.nb)
    mod = Module(path, g.P.name()); L = g.layout()
    Li, Ri, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(Li[:, :, 0]).cuda(), torch.from_numpy(Ri[:, :, 0]).cuda()
    sl, sr = sd.cross_arms_pair(dl, dr, 0.02, 14)
    pl, meta = programs_for(sl, H, W, L, Dp)
    pr, meta_r = programs_for(sr, H, W, L, Dp)
    print("program build (python): %.1f s + %.1f s; longest %d / %d of stride %d dwords" % (
        meta["build_s"], meta_r["build_s"], meta["longest"], meta_r["longest"], meta["stride"]), flush=True)
    used = int((pl.view(-1, meta["stride"]) != 0).sum()) * 4
    print("program bytes used (left image): %.1f MB" % (used / 1e6))
    gt = torch.Generator(device="cuda").manual_seed(0)
    a = -torch.rand((H, W, Dp), device="cuda", generator=gt); b = torch.empty_like(a)
    c = a.clone(); d = torch.empty_like(a)
    nchunks = -(-Dp // (64 * vpl))
    want_l, _ = sd.cbca_hwd(a, torch.empty_like(a), sl, D, 1, 14)
    want_l = want_l.clone()
    want_r, _ = sd.cbca_hwd(c, torch.empty_like(c), sr, D, 1, 14)
    want_r = want_r.clone()
    if args.drop:
        fl_, fa_ = (float(x) for x in args.drop.split(","))
        loads = set(L["loadk"]) if PIPE else set(x for row in L["load"] for x in row[1:])
        special = loads | set(L["wait"]) | {L["end"], L["refill"]}
        nop = L["wait"][0] | (L["M0_SRC1"] << 16)
        ps = []
        for pt in (pl, pr):
            pn = pt.cpu().numpy().view(np.uint32).copy()
            if PIPE:      # only the op words (the second words of a chunk follow its 64 ops)
                opmask = (np.arange(pn.size).reshape(pn.shape) % 128) < 64
            else:
                opmask = np.ones(pn.shape, bool)
            off = pn & 0xffff
            isload = opmask & np.isin(off, list(loads)) & (pn != 0)
            isadd = opmask & (pn != 0) & ~np.isin(off, list(special))
            rng = np.random.default_rng(0)
            pn[isload & (rng.random(pn.shape) < fl_)] = nop
            pn[isadd & (rng.random(pn.shape) < fa_)] = nop
            ps.append(torch.from_numpy(pn.view(np.int32)).cuda())
        k2 = kargs([a, c], [b, d], ps, [sl, sr], Dp, H, W, nchunks, meta)
        ms = timeit(lambda: mod.launch(grid_of(meta, nchunks * 2), k2), args.iters)
        print("LOAD ops dropped %3.0f%%, ADD ops dropped %3.0f%%: %8.4f ms" % (100 * fl_, 100 * fa_, ms), flush=True)
        sys.exit(0)
    ka = kargs([a, c], [b, d], [pl, pr], [sl, sr], Dp, H, W, nchunks, meta)
    b.fill_(float("nan")); d.fill_(float("nan"))
    mod.launch(grid_of(meta, nchunks * 2), ka)
    torch.cuda.synchronize()
    print("full size bit-exact vs cbca_hwd: left %s right %s" % (torch.equal(b, want_l), torch.equal(d, want_r)), flush=True)
    vb = 4.0 * H * W * D
    ms = timeit(lambda: mod.launch(grid_of(meta, nchunks * 2), ka), args.iters)
    print("K=%d W=%d NB=%d pipe=%d regs=%d" % (g.P.K, args.w, args.nb, PIPE, g.P.nvgpr))
    print("cbca_prog pair      %8.4f ms  %6.1f GB/s (%.1f%% of 8 TB/s)" % (ms, 4 * vb / ms / 1e6, 4 * vb / ms / 1e6 / 80), flush=True)
    for pf in [int(x) for x in args.pf.split(",") if x]:
        g2, path2 = build_hsaco(vpl, args.w, os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm"), args.nb, 0, pf)
        m2 = Module(path2, g2.P.name())
        b.fill_(float("nan")); d.fill_(float("nan"))
        m2.launch((8 * meta["band_groups"], meta["ngroups"], nchunks * 2), ka)
        torch.cuda.synchronize()
        ok = torch.equal(b, want_l) and torch.equal(d, want_r)
        ms = timeit(lambda: m2.launch((8 * meta["band_groups"], meta["ngroups"], nchunks * 2), ka), args.iters)
        print("  scalar prefetch %2d columns ahead: %8.4f ms  bit-exact %s" % (pf, ms, ok), flush=True)
    if args.order:
        g2 = gen.Gen(gen.Params(vpl=vpl, K=args.k, W=args.w, NB=args.nb, order=1)).build()
        base = os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm", "cbca_prog_order1")
        open(base + ".s", "w").write(g2.render())
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + ".s", "-o", base + ".o"])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
        m2 = Module(base + ".hsaco", g2.P.name())
        grid1 = (8 * meta["ngroups"], meta["band_groups"], nchunks * 2)
        b.fill_(float("nan")); d.fill_(float("nan"))
        m2.launch(grid1, ka); torch.cuda.synchronize()
        ok = torch.equal(b, want_l) and torch.equal(d, want_r)
        ms = timeit(lambda: m2.launch(grid1, ka), args.iters)
        print("  row-group-major dispatch: %8.4f ms  bit-exact %s" % (ms, ok), flush=True)
    if args.experiments:
        grid = (8 * meta["band_groups"], meta["ngroups"], nchunks * 2)
        for lds in (20480, 40960, 81920):
            ms = timeit(lambda: mod.launch(grid, ka, lds), args.iters)
            print("  with %d B of LDS per wave (%d waves per SIMD at most): %8.4f ms" % (lds, 163840 // lds // 4, ms), flush=True)
        # timing only (wrong results): ops replaced by WAIT 0 / kernels without division or stores
        loads = set(L["loadk"]) if PIPE else set(x for row in L["load"] for x in row[1:])
        special = loads | set(L["wait"]) | {L["end"], L["refill"]}
        nop = L["wait"][0] | (L["M0_SRC1"] << 16)

        def variant(fl, fa, m=None, note=""):
            ps = []
            for pt in (pl, pr):
                pn = pt.cpu().numpy().view(np.uint32).copy()
                off = pn & 0xffff
                isload = np.isin(off, list(loads))
                isadd = (pn != 0) & ~np.isin(off, list(special))
                rng = np.random.default_rng(0)
                pn[isload & (rng.random(pn.shape) < fl)] = nop
                pn[isadd & (rng.random(pn.shape) < fa)] = nop
                ps.append(torch.from_numpy(pn.view(np.int32)).cuda())
            k2 = kargs([a, c], [b, d], ps, [sl, sr], Dp, H, W, nchunks, meta)
            ms = timeit(lambda: (m or mod).launch(grid, k2), args.iters)
            print("  LOAD ops dropped %3.0f%%, ADD ops dropped %3.0f%% %s: %8.4f ms" % (100 * fl, 100 * fa, note, ms), flush=True)
        variant(0.5, 0)
        variant(1.0, 0)
        variant(1.0, 1.0)
        variant(0.0, 1.0)
        for dbg, note in ((4, "plain stores"), (8, "sc1 stores"), (12, "sc0 sc1 stores"), (16, "sc1 nt stores"), (20, "sc0 sc1 nt stores")):
            g2, path2 = build_hsaco(vpl, args.w, os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm"), args.nb, dbg)
            m2 = Module(path2, g2.P.name())
            variant(0.0, 0.0, m2, note)
        for dbg, note in ((1, "no division"), (2, "no stores"), (3, "no division, no stores")):
            g2, path2 = build_hsaco(vpl, args.w, os.path.join(ROOT, "mc-cnn-python_amd", "build", "asm"), args.nb, dbg)
            m2 = Module(path2, g2.P.name())
            variant(0.0, 0.0, m2, note)
            variant(1.0, 0.0, m2, note)
            variant(1.0, 1.0, m2, note)
    ms = timeit(lambda: sd.cbca_hwd_pair(a, b, sl, c, d, sr, D, 1, 14), args.iters)
    print("cbca_iter_hwd_pair  %8.4f ms  %6.1f GB/s (%.1f%% of 8 TB/s)" % (ms, 4 * vb / ms / 1e6, 4 * vb / ms / 1e6 / 80), flush=True)
    sys.exit(0 if allok else 1)


if __name__ == "__main__":
    main()

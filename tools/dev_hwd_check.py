#!/usr/bin/env python3
"""Dev check of the pixel-major kernels against the plane-major ones (bit-exact) + timing at a config.
    python tools/dev_hwd_check.py [--config cfg2] [--iters 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src")); sys.path.insert(0, ROOT)
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, synthetic
from bench import CONFIGS

def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def check_shape(H, W, D, seed, L=14, tau=0.02):
    Li, Ri, _, _, _ = synthetic.make_pair(H, W, min(D, W - 2), seed=seed)
    img = torch.from_numpy(Li[:, :, 0]).cuda()
    sup = sd.cross_arms(img, tau, L)
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = (torch.rand((D, H, W), device="cuda", generator=g) * 3 - 2).contiguous()
    ref, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 1, L, hip.MCCNN_CBCA_REFERENCE_ORDER)
    hv = sd.dhw_to_hwd(v)
    out, _ = sd.cbca_hwd(hv, torch.full_like(hv, float("nan")), sup, D, 1, L)
    got = sd.hwd_to_dhw(out, D)
    ok = torch.equal(got, ref)
    cnt = sd.support_count(sup)
    w1 = sd.wta(v); w2 = sd.wta_hwd(hv, D)
    okw = torch.equal(w1, w2)
    d = (w1 + 0.5 * (torch.rand_like(w1) > 0.5)).contiguous()
    s1 = sd.subpixel(d, v); s2 = sd.subpixel_hwd(d, hv, D)
    oks = torch.equal(s1.nan_to_num(123.), s2.nan_to_num(123.))
    print("shape %dx%dx%d seed %d: cbca %s (maxdiff %.3g) wta %s subpixel %s  region mean %.1f max %d" % (
        H, W, D, seed, ok, float((got - ref).abs().max()), okw, oks, float(cnt.float().mean()), int(cnt.max())), flush=True)
    return ok and okw and oks

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--skip-check", action="store_true")
    args = ap.parse_args()
    hip.require_device()
    allok = True
    if not args.skip_check:
        for (H, W, D, seed) in [(24, 32, 8, 0), (40, 48, 16, 1), (37, 61, 24, 2), (64, 130, 64, 3), (50, 70, 2, 4),
                                (33, 45, 130, 5), (20, 300, 256, 6), (70, 97, 400, 7), (128, 256, 192, 8), (5, 9, 3, 9),
                                (256, 256, 64, 10)]:
            allok &= check_shape(H, W, D, seed)
        # flat image: every arm at its limit (27 x 27 regions)
        img = torch.zeros((60, 80), device="cuda"); sup = sd.cross_arms(img, 0.02, 14)
        v = torch.rand((20, 60, 80), device="cuda") - 0.5
        ref, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
        out, _ = sd.cbca_hwd(sd.dhw_to_hwd(v), torch.empty((60, 80, 20), device="cuda"), sup, 20, 1, 14)
        ok = torch.equal(sd.hwd_to_dhw(out, 20), ref); allok &= ok
        print("flat image (max arms): %s" % ok, flush=True)
        print("ALL OK" if allok else "MISMATCH", flush=True)
    H, W, D = CONFIGS[args.config]
    Li, Ri, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(Li[:, :, 0]).cuda(), torch.from_numpy(Ri[:, :, 0]).cuda()
    sl, sr = sd.cross_arms_pair(dl, dr, 0.02, 14)
    Dp = sd.hwd_pitch(D)
    g = torch.Generator(device="cuda").manual_seed(0)
    a = -torch.rand((H, W, Dp), device="cuda", generator=g); b = torch.empty_like(a)
    c = a.clone(); d = torch.empty_like(a)
    vb = 4.0 * H * W * D
    ms = timeit(lambda: sd.cbca_hwd(a, b, sl, D, 1, 14), args.iters)
    print("cbca_iter_hwd       %8.4f ms  %6.1f GB/s (%.1f%% of 8 TB/s)" % (ms, 2 * vb / ms / 1e6, 2 * vb / ms / 1e6 / 80), flush=True)
    ms = timeit(lambda: sd.cbca_hwd_pair(a, b, sl, c, d, sr, D, 1, 14), args.iters)
    print("cbca_iter_hwd_pair  %8.4f ms  %6.1f GB/s (%.1f%% of 8 TB/s)" % (ms, 4 * vb / ms / 1e6, 4 * vb / ms / 1e6 / 80), flush=True)
    ms = timeit(lambda: sd.wta_hwd(a, D), args.iters)
    print("wta_hwd             %8.4f ms  %6.1f GB/s" % (ms, vb / ms / 1e6), flush=True)
    if not args.skip_check and H * W * D < 2e8:
        va = sd.hwd_to_dhw(a, D)
        ms = timeit(lambda: sd.cbca(va, torch.empty_like(va), sl, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER), 3)
        print("cbca_iter ref4 (DHW) %8.4f ms" % ms, flush=True)
        ref, _ = sd.cbca(va.clone(), torch.empty_like(va), sl, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
        out, _ = sd.cbca_hwd(a, b, sl, D, 1, 14)
        print("full-size bit-exact vs ref4: %s" % torch.equal(sd.hwd_to_dhw(out, D), ref), flush=True)
    sys.exit(0 if allok else 1)

if __name__ == "__main__":
    main()

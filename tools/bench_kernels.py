#!/usr/bin/env python3
"""Per-kernel micro-benchmarks of libmccnn_hip.so at a BASELINE config (HIP events on the launch stream).

    python tools/bench_kernels.py [--config cfg2] [--iters 20] [--only cbca_iter,sgm_pass]

Prints one line per kernel: avg ms per launch, algorithmic GB/s, fraction of the 8 TB/s HBM peak.  Meant to be run
under rocprofv3 (--kernel-trace --stats, or --pmc ...) when a single kernel is being tuned.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _hipabi as hip  # noqa: E402
import stereo_device as sd  # noqa: E402
import synthetic  # noqa: E402
from bench import CONFIGS, HBM_PEAK_GBS  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--shape", default="", help="H,W,D instead of a named config (scaling experiments)")
    args = ap.parse_args()
    hip.require_device()
    H, W, D = CONFIGS[args.config]
    if args.shape:
        H, W, D = (int(x) for x in args.shape.split(","))
    only = set(x for x in args.only.split(",") if x)
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    va = -torch.rand((D, H, W), device="cuda", generator=g)
    vb = torch.empty_like(va)
    vol_bytes = 4.0 * D * H * W
    sup = sd.cross_arms(dl, 0.02, 14)
    rows = []

    def add(name, fn, nbytes):
        if only and name not in only:
            return
        ms = timeit(fn, args.iters)
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append((name, ms, gbs))
        print("%-22s %8.4f ms  %8.1f GB/s  %5.1f%% of HBM peak" % (name, ms, gbs, 100 * gbs / HBM_PEAK_GBS), flush=True)

    add("cbca_iter", lambda: sd.cbca(va, vb, sup, 1, 14, hip.MCCNN_CBCA_SEPARABLE), 2 * vol_bytes)
    sup2 = sd.cross_arms(dr, 0.02, 14)
    vc, vd = torch.empty_like(va), torch.empty_like(va)
    vc.copy_(va)
    add("cbca_iter_pair", lambda: sd.cbca_pair(va, vb, sup, vc, vd, sup2, 1, 14, hip.MCCNN_CBCA_SEPARABLE),
        4 * vol_bytes)
    del vc, vd
    add("cbca_iter_reforder", lambda: sd.cbca(va, vb, sup, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER), 2 * vol_bytes)
    hwd = sd.dhw_to_hwd(va)
    hwd2 = sd.dhw_to_hwd(vb)
    scratch = sd.sgm_scratch(H, W, D, va.device)
    for name, r in (("sgm_pass_h", (0, 1)), ("sgm_pass_v", (1, 0))):
        add(name, lambda r=r: sd.sgm_pass_hwd(dl, dr, [hwd, hwd2], [0, 1], D, r, 2.3, 55.9, 4.0, 8.0, 0.08, scratch),
            4 * vol_bytes)
    # one volume per launch: what the free-running chains of StereoMatcher issue (each chain on its own stream), alone here
    for name, r in (("sgm_pass_one_volume_h", (0, 1)), ("sgm_pass_one_volume_v", (1, 0))):
        add(name, lambda r=r: sd.sgm_pass_hwd(dl, dr, [hwd], [0], D, r, 2.3, 55.9, 4.0, 8.0, 0.08, scratch), 2 * vol_bytes)
    if D <= sd.SGM_FIRST_PASS_MAX_D:
        import ctypes

        def first_pass():
            src = (ctypes.c_void_p * 2)(va.data_ptr(), vb.data_ptr())
            dst = (ctypes.c_void_p * 2)(hwd.data_ptr(), hwd2.data_ptr())
            sides = (ctypes.c_int * 2)(0, 1)
            hip.check(hip.load().mccnn_sgm_first_pass(hip.ptr(dl), hip.ptr(dr), src, dst, sides, 2, D, H, W, 2.3, 55.9,
                                                      4.0, 8.0, 0.08, hip.ptr(scratch), scratch.numel(), hip.stream()),
                      "mccnn_sgm_first_pass")
        add("sgm_first_pass", first_pass, 4 * vol_bytes)
    # the pixel-major kernels of the bit-exact variant (two-volume launches like the pair runs them)
    if not only or only & {"cbca_iter_hwd", "cbca_iter_hwd_pair", "wta_hwd", "cbca_iter_prog_pair", "cbca_iter_prog_pair_skip",
                           "cbca_prog_build", "cbca_iter_prog", "cbca_iter_prog_skip"}:
        hb, hb2 = torch.empty_like(hwd), torch.empty_like(hwd2)
        add("cbca_iter_hwd", lambda: sd.cbca_hwd(hwd, hb, sup, D, 1, 14), 2 * vol_bytes)
        add("cbca_iter_hwd_pair", lambda: sd.cbca_hwd_pair(hwd, hb, sup, hwd2, hb2, sup2, D, 1, 14), 4 * vol_bytes)
        progs = sd.cbca_prog_buffers(D, H, W, hwd.device)
        if progs is not None:
            sd.cbca_prog_build_pair(sup, sup2, D, 14, progs)
            add("cbca_prog_build", lambda: sd.cbca_prog_build_pair(sup, sup2, D, 14, progs), 0.0)
            add("cbca_iter_prog_pair", lambda: sd.cbca_prog_pair(hwd, hb, sup, hwd2, hb2, sup2, progs, D, 1, 14), 4 * vol_bytes)

            def skip_iteration():    # a third-or-later iteration: pixels whose region is the pixel itself are left alone
                hip.check(hip.load().mccnn_cbca_iter_prog_pair_skip(
                    hip.ptr(hwd), hip.ptr(hb), hip.ptr(sup), hip.ptr(progs[0]), hip.ptr(hwd2), hip.ptr(hb2), hip.ptr(sup2),
                    hip.ptr(progs[1]), D, H, W, 14, hip.stream()), "mccnn_cbca_iter_prog_pair_skip")
            add("cbca_iter_prog_pair_skip", skip_iteration, 4 * vol_bytes)
            lib = hip.load()        # one volume per launch (what StereoMatcher issues as two chains on two streams), alone
            add("cbca_iter_prog", lambda: hip.check(lib.mccnn_cbca_iter_prog(
                hip.ptr(hwd), hip.ptr(hb), hip.ptr(sup), hip.ptr(progs[0]), D, H, W, 14, hip.stream()), "mccnn_cbca_iter_prog"),
                2 * vol_bytes)
            add("cbca_iter_prog_skip", lambda: hip.check(lib.mccnn_cbca_iter_prog_skip(
                hip.ptr(hwd), hip.ptr(hb), hip.ptr(sup), hip.ptr(progs[0]), D, H, W, 14, hip.stream()), "mccnn_cbca_iter_prog_skip"),
                2 * vol_bytes)
        add("wta_hwd", lambda: sd.wta_hwd(hwd, D), vol_bytes)
        del hb, hb2
    add("dhw_to_hwd", lambda: sd.dhw_to_hwd(va, hwd), 2 * vol_bytes)
    add("hwd_to_dhw", lambda: sd.hwd_to_dhw(hwd, D, vb), 2 * vol_bytes)
    add("wta", lambda: sd.wta(va), vol_bytes)
    fl = torch.nn.functional.normalize(torch.randn((H, W, 64), device="cuda", generator=g), dim=-1).contiguous()
    fr = torch.nn.functional.normalize(torch.randn((H, W, 64), device="cuda", generator=g), dim=-1).contiguous()
    add("cost_volume_exact", lambda: sd.cost_volume(fl, fr, D, hip.MCCNN_CV_EXACT, out=(va, vb)), 2 * vol_bytes)
    if D <= 512:
        hl, hr = (torch.zeros((H, W, sd.hwd_pitch(D)), device="cuda") for _ in range(2))
        add("cost_volume_hwd", lambda: sd.cost_volume_hwd(fl, fr, D, out=(hl, hr)), 2 * vol_bytes)
    add("cost_volume_mfma", lambda: sd.cost_volume(fl, fr, D, hip.MCCNN_CV_MFMA, out=(va, vb)), 2 * vol_bytes)

    # feature stack: both views; "bytes" = the activations each of the four 64 -> 64 layers reads and writes once
    from model import NET
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda", seed=0)
    act_bytes = 4 * 2 * 2.0 * H * W * 256
    add("features_miopen", lambda: net.features_pair_hwc(dl, dr), act_bytes)
    add("features_split", lambda: net.features_pair_hwc_split(dl, dr), act_bytes)
    if not only or "conv3x3_split" in only:
        x = sd.conv1_split(torch.stack((dl, dr)).contiguous(), net.weights[0].detach().contiguous(),
                           net.biases[0].detach(), 5)
        pair = torch.stack((dl, dr)).contiguous()
        ms = timeit(lambda: sd.conv1_split(pair, net.weights[0].detach().contiguous(), net.biases[0].detach(), 5),
                    args.iters)
        print("%-22s %8.4f ms" % ("conv1_split", ms), flush=True)
        pk, ws = sd.conv3x3_split_pack(net.weights[1])
        ms = timeit(lambda: sd.conv3x3_split(x, pk, ws, net.biases[1].detach(), last=False), args.iters)
        ms_last = timeit(lambda: sd.conv3x3_split(x, pk, ws, net.biases[1].detach(), last=True), args.iters)
        print("%-22s %8.4f ms" % ("conv3x3_split(last)", ms_last), flush=True)
        flop = 2.0 * 2 * (H + 6) * (W + 6) * 64 * 64 * 9
        print("%-22s %8.4f ms  %8.1f TFLOP/s float32-equivalent, %6.1f TFLOP/s f16 MFMA issued (3 products, of 2500)"
              % ("conv3x3_split", ms, flop / ms / 1e9, 3 * flop / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()

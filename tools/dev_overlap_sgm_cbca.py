#!/usr/bin/env python3
"""Dev harness (GPU box): what does running one volume's SGM passes beside the other volume's aggregation launches buy?
The SGM scanline kernel is bandwidth-bound with one wave per SIMD, the aggregation interpreter waits on latencies with
three: times N x (4 one-volume SGM passes) on stream A and N x (4 one-volume aggregation launches) on stream B, one
after the other and at the same time.   python tools/dev_overlap_sgm_cbca.py [--config cfg2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
from bench import CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    hip.require_device()
    H, W, D = CONFIGS[args.config]
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
    sl, sr = sd.cross_arms_pair(dl, dr, 0.02, 14)
    progs = sd.cbca_prog_buffers(D, H, W, dl.device)
    sd.cbca_prog_build_pair(sl, sr, D, 14, progs)
    g = torch.Generator(device="cuda").manual_seed(0)
    Dp = sd.hwd_pitch(D)
    a = -torch.rand((H, W, Dp), device="cuda", generator=g); b = torch.empty_like(a)
    c = -torch.rand((H, W, Dp), device="cuda", generator=g)
    scratch = sd.sgm_scratch(H, W, D, dl.device)
    lib = hip.load()
    dirs = [(0, 1), (0, -1), (-1, 0), (1, 0)]

    def sgm4():
        for r in dirs:
            sd.sgm_pass_hwd(dl, dr, [c], [hip.MCCNN_SIDE_LEFT], D, r, 2.3, 55.9, 4.0, 8.0, 0.08, scratch)

    def agg(fn, n):
        x, y = a, b
        for _ in range(n):
            hip.check(fn(hip.ptr(x), hip.ptr(y), hip.ptr(sl), hip.ptr(progs[0]), D, H, W, 14, hip.stream()), "agg")
            x, y = y, x

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(what):
        main = torch.cuda.current_stream()
        ts = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            what(main)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    for name, fn, n in (("4 full one-volume aggregation launches", lib.mccnn_cbca_iter_prog, 4),
                        ("4 skip one-volume aggregation launches", lib.mccnn_cbca_iter_prog_skip, 4)):
        t_sgm = timed(lambda m: sgm4())
        t_agg = timed(lambda m: agg(fn, n))

        def both(m):
            sA.wait_stream(m); sB.wait_stream(m)
            with torch.cuda.stream(sA):
                sgm4()
            with torch.cuda.stream(sB):
                agg(fn, n)
            m.wait_stream(sA); m.wait_stream(sB)
        t_both = timed(both)
        print("4 one-volume SGM passes %.4f ms; %s %.4f ms; one after the other %.4f ms; at the same time %.4f ms (%.1f %% of the sum)"
              % (t_sgm, name, t_agg, t_sgm + t_agg, t_both, 100.0 * t_both / (t_sgm + t_agg)), flush=True)


if __name__ == "__main__":
    main()

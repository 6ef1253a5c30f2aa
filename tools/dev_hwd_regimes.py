#!/usr/bin/env python3
"""Timing of the pixel-major reference-order CBCA on two extreme images at a config: white noise (every arm 0: one
element per region) and a constant image (every arm at its limit: 27 x 27 regions), plus the synthetic pair."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src")); sys.path.insert(0, ROOT)
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, synthetic
from bench import CONFIGS
from dev_hwd_check import timeit
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
hip.require_device()
H, W, D = CONFIGS[args.config]
Dp = sd.hwd_pitch(D)
a = -torch.rand((H, W, Dp), device="cuda"); b = torch.empty_like(a)
vb = 4.0 * H * W * D
Li = synthetic.make_pair(H, W, D, seed=100)[0]
for name, img in (("noise (arms 0)", torch.randn((H, W), device="cuda")), ("constant (arms 13)", torch.zeros((H, W), device="cuda")),
                  ("synthetic pair", torch.from_numpy(Li[:, :, 0]).cuda())):
    sup = sd.cross_arms(img, 0.02, 14)
    ms = timeit(lambda: sd.cbca_hwd(a, b, sup, D, 1, 14), args.iters)
    print("%-20s region mean %6.1f  %8.4f ms  %6.1f GB/s" % (name, float(sd.support_count(sup).float().mean()), ms, 2 * vb / ms / 1e6), flush=True)

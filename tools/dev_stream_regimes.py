#!/usr/bin/env python3
"""Timing of the streaming (separable) CBCA kernel on images whose arms are all equal (white noise: 0, constant: 13 -
its LDS gathers are conflict-free then) and on the synthetic pair (unequal arms in neighbouring lanes)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src")); sys.path.insert(0, ROOT)
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, synthetic
from bench import CONFIGS
from dev_hwd_check import timeit
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="cfg2"); ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
hip.require_device()
H, W, D = CONFIGS[args.config]
a = -torch.rand((D, H, W), device="cuda"); b = torch.empty_like(a); c = a.clone(); d = torch.empty_like(a)
vb = 4.0 * H * W * D
Li = synthetic.make_pair(H, W, D, seed=100)[0]
for name, img in (("noise (arms 0)", torch.randn((H, W), device="cuda")), ("constant (arms 13)", torch.zeros((H, W), device="cuda")),
                  ("synthetic pair", torch.from_numpy(Li[:, :, 0]).cuda())):
    sup = sd.cross_arms(img, 0.02, 14); sup2 = sd.cross_arms(img.flip(1).contiguous(), 0.02, 14)
    ms = timeit(lambda: sd.cbca_pair(a, b, sup, c, d, sup2, 1, 14, hip.MCCNN_CBCA_SEPARABLE), args.iters)
    print("%-20s cbca_iter_pair %8.4f ms  %6.1f GB/s" % (name, ms, 4 * vb / ms / 1e6), flush=True)

#!/usr/bin/env python3
"""Where does the pixel-major CBCA differ from the plane-major reference-order kernel?  (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src")); sys.path.insert(0, ROOT)
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, synthetic
H, W, D, seed = 20, 300, 256, 6
Li = synthetic.make_pair(H, W, min(D, W - 2), seed=seed)[0]
sup = sd.cross_arms(torch.from_numpy(Li[:, :, 0]).cuda(), 0.02, 14)
g = torch.Generator(device="cuda").manual_seed(seed)
v = (torch.rand((D, H, W), device="cuda", generator=g) * 3 - 2).contiguous()
ref, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
hv = sd.dhw_to_hwd(v)
out, _ = sd.cbca_hwd(hv, torch.full_like(hv, float("nan")), sup, D, 1, 14)
got = sd.hwd_to_dhw(out, D)
bad = (got != ref) | torch.isnan(got)
print("mismatches", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero().cpu().numpy()
print("d values:", np.unique(idx[:, 0])[:40], "n", len(np.unique(idx[:, 0])))
print("y values:", np.unique(idx[:, 1]))
print("x values:", np.unique(idx[:, 2])[:60], "n", len(np.unique(idx[:, 2])))
print("x mod 5:", np.bincount(idx[:, 2] % 5))
for i in idx[:10]:
    print(i, float(got[tuple(i)]), float(ref[tuple(i)]), float(got[tuple(i)]) / float(ref[tuple(i)]))

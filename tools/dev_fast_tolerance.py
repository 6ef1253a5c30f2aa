#!/usr/bin/env python3
"""Dev (GPU box): the distance of the fast variants (and of the bit-exact stages behind split features) from the
bit-exact variant at every BASELINE configuration - the numbers behind src/tolerances.py."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, synthetic, tf_checkpoint
from bench import CONFIGS
from model import NET

layers = tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz"))
net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(layers)
out = {}
for cfg in sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4"]:
    H, W, D = CONFIGS[cfg]
    for seed in (100, 101):
        L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=seed)
        l, r = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
        ke = {}
        e = sd.StereoMatcher(net).match(l, r, D, keep=ke)
        for name, kw in (("fast", dict(cv_mode=hip.MCCNN_CV_MFMA, cbca_order=hip.MCCNN_CBCA_SEPARABLE, features="split_f16")),
                         ("fast_libfeat", dict(cv_mode=hip.MCCNN_CV_MFMA, cbca_order=hip.MCCNN_CBCA_SEPARABLE)),
                         ("exact_splitfeat", dict(features="split_f16"))):
            kf = {}
            f = sd.StereoMatcher(net, **kw).match(l, r, D, keep=kf)
            both_nan = torch.isnan(f) & torch.isnan(e)
            d = torch.where(both_nan, torch.zeros_like(f), (f - e).abs())
            d = torch.nan_to_num(d, nan=float("inf"), posinf=float("inf")).flatten()
            n = d.numel()
            rec = {"pixels": n, "flips_l": int((kf["wta"][0] != ke["wta"][0]).sum()), "flips_r": int((kf["wta"][1] != ke["wta"][1]).sum()),
                   "nonfinite_exact": int((~torch.isfinite(e)).sum()), "nonfinite_this": int((~torch.isfinite(f)).sum()),
                   "abs_exact_gt_D": int((e.abs() > D).sum())}
            for t in (1e-3, 1e-2, 0.1, 0.5, 1.0):
                rec["frac_le_%g" % t] = round(float((d <= t).float().mean()), 6)
            for q in (0.99, 0.995, 0.999, 0.9999):
                rec["p%g" % (100 * q)] = float(d.kthvalue(max(1, int(round(q * n)))).values)
            out["%s_seed%d_%s" % (cfg, seed, name)] = rec
            print(cfg, seed, name, json.dumps(rec), flush=True)
            del kf, f
        del ke, e
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fast_tolerance.json"), "w"), indent=1)

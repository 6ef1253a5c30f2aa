#!/usr/bin/env python3
"""Dev (GPU box): final-map agreement with the reference's golden maps for both feature paths."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import stereo_device as sd, tf_checkpoint
from model import NET
layers = tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz"))
net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(layers)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz"))):
    g = dict(np.load(path)); D = g["cv_l"].shape[0]
    for feat in ("miopen", "split_f16"):
        m = sd.StereoMatcher(net, features=feat); keep = {}
        out = m.match(dev(g["left"]), dev(g["right"]), D, keep=keep).cpu().numpy()
        l, r = dev(g["left"][:, :, 0]), dev(g["right"][:, :, 0])
        f = net.features_pair_hwc(l, r) if feat == "miopen" else net.features_pair_hwc_split(l, r)
        fe = max(np.abs(f[0].cpu().numpy() - g["fl"]).max(), np.abs(f[1].cpu().numpy() - g["fr"]).max())
        d = np.abs(out - g["bilateral"]); d = d[np.isfinite(d)]
        cvd = np.abs(keep["cv"][0].cpu().numpy() - g["cv_l"]).max()
        print("%-22s %-9s feature err %.2e cv err %.2e | flips %d | within 1e-3: %.4f 1e-2: %.4f | max %.3g | sorted top %s" % (
            os.path.basename(path), feat, fe, cvd, int((keep["wta"][0].cpu().numpy() != g["wta_l"]).sum()),
            (d <= 1e-3).mean(), (d <= 1e-2).mean(), d.max(), np.round(np.sort(d)[-5:], 4)), flush=True)

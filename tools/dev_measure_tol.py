#!/usr/bin/env python3
"""Dev: measure the separable-CBCA and whole-pair differences on the golden cases (numbers behind the test bounds)."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mc-cnn-python_amd", "src"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import _hipabi as hip, stereo_device as sd, process_functional as pf, tf_checkpoint
from helpers import hp_of
from model import NET
layers = tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz"))
net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(layers)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz"))):
    g = dict(np.load(path)); hp = hp_of(g); name = os.path.basename(path)
    pf.CBCA_ORDER = "separable"
    l1, r1 = pf.cost_volume_aggregation(g["left"], g["right"], g["cv_l"], g["cv_r"], hp["cbca_intensity"], hp["cbca_distance"], 1)
    l16, r16 = pf.cost_volume_aggregation(g["left"], g["right"], g["sgm_l"], g["sgm_r"], hp["cbca_intensity"], hp["cbca_distance"], hp["it2"])
    sp1 = np.spacing(np.float32(np.abs(g["cv_l"]).max())); sp16 = np.spacing(np.float32(np.abs(g["sgm_l"]).max()))
    print(name, "1 it: max|d| %.3e = %.2f spacings(max|in|=%.3g)" % (np.abs(l1 - g["cbca1it_l"]).max(), np.abs(l1 - g["cbca1it_l"]).max() / sp1, np.abs(g["cv_l"]).max()),
          "| 16 it: max|d| %.3e = %.2f spacings(max|in|=%.4g)" % (max(np.abs(l16 - g["cbca2_l"]).max(), np.abs(r16 - g["cbca2_r"]).max()),
                                                                  max(np.abs(l16 - g["cbca2_l"]).max(), np.abs(r16 - g["cbca2_r"]).max()) / sp16, np.abs(g["sgm_l"]).max()))
    D = g["cv_l"].shape[0]
    for label, cv, order in (("exact", hip.MCCNN_CV_EXACT, hip.MCCNN_CBCA_REFERENCE_ORDER), ("fast", hip.MCCNN_CV_MFMA, hip.MCCNN_CBCA_SEPARABLE)):
        m = sd.StereoMatcher(net, cv_mode=cv, cbca_order=order); keep = {}
        out = m.match(dev(g["left"]), dev(g["right"]), D, keep=keep).cpu().numpy()
        flips = int((keep["wta"][0].cpu().numpy() != g["wta_l"]).sum())
        close = np.isclose(out, g["bilateral"], atol=1e-3, equal_nan=True).mean()
        print("   whole pair %-5s: WTA flips %d of %d, within 1e-3 px %.4f, max|d| %.3g" % (label, flips, out.size, close, np.nanmax(np.abs(out - g["bilateral"]))))

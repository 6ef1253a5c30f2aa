#!/usr/bin/env python3
"""Dev harness (GPU box): per-launch events and stage brackets of a cfg2 pair for several StereoMatcher option sets.
    python tools/dev_ab_stages.py refresh_first=False ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
import tf_checkpoint
from model import NET
from bench import CONFIGS


def main():
    hip.require_device()
    cfg = "cfg2"
    variants = [{}]
    for a in sys.argv[1:]:
        if a in CONFIGS:
            cfg = a
            continue
        k, v = a.split("=")
        variants.append({k: {"True": True, "False": False}.get(v, v)})
    H, W, D = CONFIGS[cfg]
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda", seed=0)
    net.set_layers(tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz")))
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
    ref = None
    for kw in variants:
        m = sd.StereoMatcher(net, on_saturation="ignore", **kw)
        out = m.match(dl, dr, D)
        m.match(dl, dr, D)
        torch.cuda.synchronize()
        ref = out if ref is None else ref
        same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        timer = sd.StageTimer(True)
        for _ in range(5):
            m.match(dl, dr, D, timer=timer)
        torch.cuda.synchronize()
        rec = {k: (float(np.mean(v)), len(v) // 5) for k, v in timer.summary_ms().items()}
        spans = {k: float(np.mean(v)) for k, v in timer.spans_ms().items()}
        print("== %s   same bits as the default: %s" % (kw or "default", same))
        print("   spans ms:", {k: round(v, 4) for k, v in spans.items()})
        print("   launches (mean ms, per pair):", {k: (round(a, 4), n) for k, (a, n) in sorted(rec.items())})
        # the first aggregation launch by launch
        agg = [v for v in timer.records if v[0].startswith("cbca_iter_prog")][:8]
        print("   first launches:", [(n, round(a.elapsed_time(b), 4)) for n, a, b in agg], flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

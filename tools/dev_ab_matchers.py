#!/usr/bin/env python3
"""Dev harness (GPU box): same-box A/B of StereoMatcher options on the benchmark pair, graph replays, alternating.
    python tools/dev_ab_matchers.py one_launch_builder=False two_chains=False ...   (each argument = one variant against the default)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
import tf_checkpoint
from bench import CONFIGS
from model import NET


def main():
    hip.require_device()
    cfg = os.environ.get("CFG", "cfg2")
    H, W, D = CONFIGS[cfg]
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(
        tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz")))
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
    variants = [("default", {})]
    for a in sys.argv[1:]:
        kw = {}
        for kv in a.split(","):
            k, v = kv.split("=")
            kw[k] = {"True": True, "False": False}.get(v, v)
        variants.append((a, kw))
    ms = {}
    ref = None
    shared = None
    for name, kw in variants:
        m = sd.StereoMatcher(net, on_saturation="ignore", **kw)
        # SHARE_WS=1: every variant runs on the FIRST matcher's workspace (the same buffers): identical matchers with
        # workspaces of their own differ by up to 4 % (where the allocations happen to lie: profiles/r06_alloc_placement.txt)
        if os.environ.get("SHARE_WS") == "1":
            if shared is None:
                shared = m.workspace(H, W, D)
            else:
                m._ws = {(H, W, D): shared}
        out = m.match_graph(dl, dr, D).clone()
        ref = out if ref is None else ref
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), name
        ms[name] = (m, [])
    for rep in range(5):
        for name, (m, ts) in ms.items():
            for _ in range(5):
                m.match_graph(dl, dr, D)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(40):
                m.match_graph(dl, dr, D)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) / 40 * 1e3)
    for name, (m, ts) in ms.items():
        print("%-40s %s  median %.3f ms" % (name, " ".join("%.3f" % t for t in ts), sorted(ts)[len(ts) // 2]), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Dev harness (GPU box): does the time of a pair depend on WHERE its four volume buffers lie relative to each other?
Matchers whose workspace volumes are carved out of one slab with a given skew between consecutive buffers (bytes added to
the 384 MB stride), against matchers with torch's own four allocations; graph replays, alternating.
    python tools/dev_alloc_skew.py [skew_bytes ...]"""
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
    H, W, D = CONFIGS[os.environ.get("CFG", "cfg2")]
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(
        tf_checkpoint.load_fast_net_weights(os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz")))
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
    dl, dr = torch.from_numpy(L[:, :, 0]).cuda(), torch.from_numpy(R[:, :, 0]).cuda()
    skews = [int(a) for a in sys.argv[1:]] or [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096 + 256]
    variants = [("torch x4 (a)", None), ("torch x4 (b)", None)] + [("slab skew %d" % k, k) for k in skews] + [("torch x4 (c)", None)]
    if os.environ.get("LOTTERY"):      # N matchers of either kind, created alternately: is one slab reliably better?
        variants = []
        for i in range(int(os.environ["LOTTERY"])):
            variants += [("torch x4 #%d" % i, None), ("slab #%d" % i, 0)]
    ms, ref = {}, None
    for name, skew in variants:
        m = sd.StereoMatcher(net, on_saturation="ignore")
        ws = m.workspace(H, W, D)
        if skew is not None:
            n = ws["vol"][0].numel()
            stride = n + skew // 4
            slab = torch.empty((4 * stride,), dtype=torch.float32, device="cuda")
            ws["vol"] = [slab[i * stride:i * stride + n] for i in range(4)]
            ws["_slab"] = slab
        addrs = [v.data_ptr() for v in ws["vol"]]
        out = m.match_graph(dl, dr, D).clone()
        ref = out if ref is None else ref
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), name
        ms[name] = (m, [], addrs)
    for rep in range(4):
        for name, (m, ts, _) in ms.items():
            for _ in range(5):
                m.match_graph(dl, dr, D)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(40):
                m.match_graph(dl, dr, D)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) / 40 * 1e3)
    for name, (m, ts, addrs) in ms.items():
        rel = [(a - addrs[0]) % (1 << 21) for a in addrs]
        print("%-28s %s  median %.3f ms   buffer offsets mod 2 MiB: %s  base mod 2 MiB: %d" % (
            name, " ".join("%.3f" % t for t in ts), sorted(ts)[len(ts) // 2], rel, addrs[0] % (1 << 21)), flush=True)


if __name__ == "__main__":
    main()

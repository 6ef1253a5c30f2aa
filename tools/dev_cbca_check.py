#!/usr/bin/env python3
"""Dev check: streaming CBCA kernel (order 0) vs the bit-exact reference-order kernel (order 1) on assorted shapes, one iteration and several.  Prints max |diff| per shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import _hipabi as hip  # noqa: E402
import stereo_device as sd  # noqa: E402
import synthetic  # noqa: E402

hip.require_device()
bad = 0
shapes = [(48, 64, 4), (40, 223, 3), (37, 224, 3), (33, 225, 2), (70, 449, 3), (9, 5, 2), (3, 1, 2), (130, 750, 8),
          (500, 750, 6), (375, 1242, 4), (100, 1501, 3), (256, 256, 8), (65, 231, 5), (700, 300, 3)]
for (H, W, D) in shapes:
    L, R, _, _, _ = synthetic.make_pair(H, max(W, 8), max(D, 2), seed=H + W)
    img = torch.from_numpy(np.ascontiguousarray(L[:, :W, 0])).cuda()
    sup = sd.cross_arms(img, 0.02, 14)
    g = torch.Generator(device="cuda").manual_seed(H * 7 + W)
    v = (torch.rand((D, H, W), device="cuda", generator=g) * 3 - 2).contiguous()
    outs = {}
    for order in (0, 1):
        guard = torch.full((D * H * W + 64,), 777.0, device="cuda")
        dst = guard[32:32 + D * H * W].view(D, H, W)
        hip.check(hip.load().mccnn_cbca_iter(hip.ptr(v), hip.ptr(dst), hip.ptr(sup), D, H, W, 14, order, hip.stream()),
                  "cbca")
        torch.cuda.synchronize()
        assert float(guard[:32].min()) == 777.0 and float(guard[-32:].min()) == 777.0, "out-of-bounds store"
        outs[order] = dst.clone()
    d01 = float((outs[0] - outs[1]).abs().max())
    # several iterations, ping-pong
    a, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 5, 14, 0)
    b, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 5, 14, 1)
    d5 = float((a - b).abs().max())
    ok = d01 <= 1e-6 and d5 <= 2e-6 and bool(torch.isfinite(outs[0]).all())
    bad += 0 if ok else 1
    print("%4dx%4dx%2d  streaming vs reference order: 1 iteration %.2e, 5 iterations %.2e  (largest region %d)  %s"
          % (H, W, D, d01, d5, int(sd.support_count(sup).max()), "ok" if ok else "FAIL"), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)

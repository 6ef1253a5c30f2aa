#!/usr/bin/env python3
"""Dev: barrier-wait profile of the streaming CBCA kernel (library built with -DCBCA_PROFILE)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))
import torch
import _hipabi as hip
import stereo_device as sd
import synthetic
H, W, D = 500, 750, 256
L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100)
img = torch.from_numpy(L[:, :, 0]).cuda()
sup = sd.cross_arms(img, 0.02, 14)
va = -torch.rand((D, H, W), device="cuda"); vb = torch.empty_like(va)
lib = hip.load()
lib.mccnn_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
sd.cbca(va, vb, sup, 1, 14, 0); torch.cuda.synchronize()
lib.mccnn_debug_prof(None, 1)
N = 5
for _ in range(N): sd.cbca(va, vb, sup, 1, 14, 0)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 32)()
lib.mccnn_debug_prof(out, 0)
for r, name in enumerate(("scan", "hsum", "emit")):
    w, t, n = out[r * 4], out[r * 4 + 1], out[r * 4 + 2]
    print("%s: waves %d, avg total %.0f cycles, parked at barrier %.1f%%" % (name, n / N, t / max(n, 1), 100.0 * w / max(t, 1)))

names = {"scan": ["load wait+cvt+local", "dpp scan", "lds stores", "issue loads"],
         "hsum": ["word wait+unpack+issue reads", "read wait+diff", "prefix+ring stores", "issue loads"],
         "emit": ["word wait+addr+issue reads", "read wait+math+stores", "issue loads", "-"]}
for r, name in enumerate(("scan", "hsum", "emit")):
    n = out[r * 4 + 2]
    print(name, " | ".join("%s %.0f" % (names[name][i], out[12 + r * 4 + i] / max(n, 1) / 138.0) for i in range(4)), "(cycles per iteration)")
tr = (ctypes.c_ulonglong * (8 * 8 * 160))()
lib.mccnn_debug_trace.argtypes = [ctypes.c_void_p]
lib.mccnn_debug_trace(tr)
import numpy as np
t = np.array(tr[:], dtype=np.int64).reshape(8, 8, 80, 2)
for blk in ():
    print("block", blk * 61, "(wave: per-iteration [work cycles before arriving | parked at barrier])")
    for wv in range(8):
        arr, dep = t[blk, wv, :, 0], t[blk, wv, :, 1]
        work = arr[1:] - dep[:-1]
        park = dep[1:] - arr[1:]
        print(" wave %d work: %s" % (wv, " ".join("%4d" % x for x in work[20:44])))
        print("        park: %s" % (" ".join("%4d" % x for x in park[20:44])))
    it = t[blk, 0, 1:, 1] - t[blk, 0, :-1, 1]
    print(" iteration period (wave 0): mean %.0f  min %d  max %d  p90 %.0f" % (it[5:75].mean(), it[5:75].min(), it[5:75].max(), np.percentile(it[5:75], 90)))

#!/usr/bin/env python3
"""Generates the golden vectors in this directory by RUNNING THE REFERENCE ITSELF (dev container only).

    python tests/golden/gen_golden.py            # writes tests/golden/ref_*.npz + mccnn_fast_weights.npz

/root/reference/src/process_functional.py is loaded in memory through ref_shim.py (Python-2 print / integer
division rewritten on the fly, tensorflow/cv2 stubbed) and every stage the reference times in match.py:129-179 is
executed on small seeded synthetic pairs.  Only numbers are written here: inputs, and the reference's outputs.

What is and is not pinned by the real reference:
  * a2..a11 (cost volume ... bilateral): outputs of the reference's own NumPy code  -> pinned.
  * a1 (features): TensorFlow is not installable here, so `fl`/`fr` come from the float64-accumulating
    restatement in oracle/mccnn_oracle.c fed with the reference's trained checkpoint (parsed by
    mc-cnn-python_amd/src/tf_checkpoint.py, CRC32C-verified).  Feature parity with TF is therefore UNPINNED.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_shim  # noqa: E402
import synthetic  # noqa: E402
import tf_checkpoint  # noqa: E402
import oracle  # noqa: E402

CKPT = "/root/reference/data/tensorboard_log/model_epoch2000.ckpt"
HP = dict(cbca_intensity=0.02, cbca_distance=14, it1=2, it2=16, sgm_P1=2.3, sgm_P2=55.9, sgm_Q1=4, sgm_Q2=8,
          sgm_D=0.08, sgm_V=1.5, blur_sigma=6, blur_threshold=2)
CASES = [(24, 32, 8, 0), (24, 32, 8, 1), (40, 48, 16, 0), (40, 48, 16, 1)]
DIRS = {"right": (0, 1), "left": (0, -1), "up": (-1, 0), "bottom": (1, 0)}


def arms_from_region(region, num):
    """Derives (up, down, left, right) arm lengths of each pixel from the reference's coordinate lists."""
    H, W = num.shape
    arms = np.zeros((H, W, 4), dtype=np.uint8)
    for h in range(H):
        for w in range(W):
            pts = region[h, w, :num[h, w]]
            col = pts[pts[:, 1] == w][:, 0]          # vertical arm = entries on the anchor column ...
            row = pts[pts[:, 0] == h][:, 1]          # horizontal arm of the anchor = entries on its row
            arms[h, w] = (h - col.min(), col.max() - h, w - row.min(), row.max() - w)
    return arms


def run_case(pf, H, W, D, seed, layers):
    t0 = time.time()
    L, R, l8, r8, dmap = synthetic.make_pair(H, W, D, seed)
    fl, fr = oracle.compute_features(L, R, 11, 11, layers)
    out = dict(left_u8=l8, right_u8=r8, left=L, right=R, fl=fl, fr=fr, true_disp=dmap,
               hp_names=np.array(sorted(HP)), hp_values=np.array([HP[k] for k in sorted(HP)], dtype=np.float64))
    with ref_shim.quiet():
        lcv, rcv = pf.compute_cost_volume(fl, fr, D)
        out["cv_l"], out["cv_r"] = lcv, rcv

        regl, numl = pf.compute_cross_region(L, HP["cbca_intensity"], HP["cbca_distance"])
        regr, numr = pf.compute_cross_region(R, HP["cbca_intensity"], HP["cbca_distance"])
        out["region_num_l"], out["region_num_r"] = numl, numr
        out["arms_l"], out["arms_r"] = arms_from_region(regl, numl), arms_from_region(regr, numr)
        if (H, W) == (24, 32) and seed == 0:
            out["region_l_crop"] = regl[:6, :8].copy()     # explicit coordinate lists for an 6x8 corner

        a1l, a1r = pf.cost_volume_aggregation(L, R, lcv, rcv, HP["cbca_intensity"], HP["cbca_distance"], 1)
        out["cbca1it_l"], out["cbca1it_r"] = a1l, a1r
        a2l, a2r = pf.cost_volume_aggregation(L, R, lcv, rcv, HP["cbca_intensity"], HP["cbca_distance"], HP["it1"])
        out["cbca1_l"], out["cbca1_r"] = a2l, a2r

        # one direction at a time from a common input (semi_global_matching mutates -> pass copies)
        for name, r in DIRS.items():
            p1 = HP["sgm_P1"] if r[0] == 0 else HP["sgm_P1"] / HP["sgm_V"]
            out["sgm_%s_l" % name] = pf.semi_global_matching(L, R, a2l.copy(), r, p1, HP["sgm_P2"], HP["sgm_Q1"],
                                                             HP["sgm_Q2"], HP["sgm_D"], "L")
            out["sgm_%s_r" % name] = pf.semi_global_matching(L, R, a2r.copy(), r, p1, HP["sgm_P2"], HP["sgm_Q1"],
                                                             HP["sgm_Q2"], HP["sgm_D"], "R")
        sl, sr = pf.SGM_average(a2l.copy(), a2r.copy(), L, R, HP["sgm_P1"], HP["sgm_P2"], HP["sgm_Q1"],
                                HP["sgm_Q2"], HP["sgm_D"], HP["sgm_V"])
        out["sgm_l"], out["sgm_r"] = sl, sr

        cl, cr = pf.cost_volume_aggregation(L, R, sl, sr, HP["cbca_intensity"], HP["cbca_distance"], HP["it2"])
        out["cbca2_l"], out["cbca2_r"] = cl, cr

        dl, dr = pf.disparity_prediction(cl, cr)
        out["wta_l"], out["wta_r"] = dl, dr
        di = pf.interpolation(dl, dr, D)
        out["interp"] = di
        ds = pf.subpixel_enhance(di, cl)
        out["subpixel"] = ds
        dm = pf.median_filter(ds, 5, 5)
        out["median"] = dm
        db = pf.bilateral_filter(L, dm, 5, 5, 0, HP["blur_sigma"], HP["blur_threshold"])
        out["bilateral"] = db
    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and k not in ("hp_values",):
            raise AssertionError("stage %s returned float64" % k)
    path = os.path.join(HERE, "ref_%dx%dx%d_s%d.npz" % (H, W, D, seed))
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f s, %d KB)" % (path, time.time() - t0, os.path.getsize(path) // 1024))


def main():
    assert ref_shim.available(), "the reference is only present in the dev container"
    pf, _util = ref_shim.load_reference()
    layers = tf_checkpoint.load_fast_net_weights(CKPT)
    wpath = os.path.join(HERE, "mccnn_fast_weights.npz")
    blob = {}
    for k, (w, b) in enumerate(layers, start=1):
        blob["conv%d/weights" % k] = w
        blob["conv%d/biases" % k] = b
    np.savez_compressed(wpath, **blob)
    print("wrote %s (%d KB)" % (wpath, os.path.getsize(wpath) // 1024))
    for (H, W, D, seed) in CASES:
        run_case(pf, H, W, D, seed, layers)


if __name__ == "__main__":
    main()

"""ctypes front-end of the CPU oracle (oracle/mccnn_oracle.c).

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product path in mc-cnn-python_amd/.  Function names, argument order and return values mirror
/root/reference/src/process_functional.py so parity tests read like calls into the reference.

Scalar handling follows NumPy 2 promotion in the reference expressions: Python-float hyper-parameters are rounded
to float32 where they meet a float32 array (pf:504-505, 535-541), `sgm_P1/sgm_V` is a double division rounded once
(pf:204), the bilateral kernel is evaluated in float64 by util.normal and stored as float32 (pf:433-436).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmccnn_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "mccnn_oracle.c")
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def _img2d(image):
    image = _f32(image)
    if image.ndim == 3:
        assert image.shape[2] == 1
        image = image[:, :, 0]
    return np.ascontiguousarray(image)


# ---------------------------------------------------------------------------------------------------------
def net_features(image, layers):
    """model.py:51-64 on the once-padded image (pf:20-34).  image [H,W,1] -> [H,W,64] float32."""
    img = _img2d(image)
    H, W = img.shape
    nl = len(layers)
    pad = nl  # (patch-1)/2 with patch = 2*nl+1 (11 for 5 layers)
    x = np.zeros((H + 2 * pad, W + 2 * pad, 1), dtype=np.float32)
    x[pad:pad + H, pad:pad + W, 0] = img
    L = lib()
    for k, (w, b) in enumerate(layers):
        w = _f32(w)
        b = _f32(b)
        hin, win, cin = x.shape
        cout = w.shape[3]
        assert w.shape == (3, 3, cin, cout)
        y = np.empty((hin - 2, win - 2, cout), dtype=np.float32)
        L.orc_conv3x3_valid(_p(x), hin, win, cin, _p(w), _p(b), cout, int(k < nl - 1), _p(y))
        x = y
    L.orc_l2_normalize(_p(x), ctypes.c_long(x.shape[0] * x.shape[1]), x.shape[2])
    return x


def compute_features(left_image, right_image, patch_height, patch_width, layers):
    """pf:15-73 with the network weights passed as a list of (w_hwio, bias)."""
    assert patch_height == patch_width == 2 * len(layers) + 1
    return net_features(left_image, layers), net_features(right_image, layers)


def compute_cost_volume(featuresl, featuresr, ndisp):
    fl, fr = _f32(featuresl), _f32(featuresr)
    H, W, C = fl.shape
    lcv = np.empty((ndisp, H, W), dtype=np.float32)
    rcv = np.empty((ndisp, H, W), dtype=np.float32)
    lib().orc_cost_volume(_p(fl), _p(fr), H, W, C, int(ndisp), _p(lcv), _p(rcv))
    return lcv, rcv


def cross_arms(image, intensity_threshold, distance_threshold):
    img = _img2d(image)
    H, W = img.shape
    arms = np.empty((H, W, 4), dtype=np.uint8)
    cnt = np.empty((H, W), dtype=np.int32)
    lib().orc_cross_arms(_p(img), H, W, ctypes.c_float(np.float32(intensity_threshold)), int(distance_threshold),
                         _p(arms, _u8p), _p(cnt, _i32p))
    return arms, cnt


def compute_cross_region(image, intensity_threshold, distance_threshold):
    img = _img2d(image)
    H, W = img.shape
    L = int(distance_threshold)
    region = np.empty((H, W, (2 * L) ** 2, 2), dtype=np.int32)
    num = np.empty((H, W), dtype=np.int32)
    lib().orc_cross_region(_p(img), H, W, ctypes.c_float(np.float32(intensity_threshold)), L,
                           _p(region, _i32p), _p(num, _i32p))
    return region, num


def cost_volume_aggregation(left_image, right_image, left_cost_volume, right_cost_volume,
                            intensity_threshold, distance_threshold, max_average_time):
    outs = []
    for img, vol in ((left_image, left_cost_volume), (right_image, right_cost_volume)):
        img = _img2d(img)
        v = _f32(vol).copy()
        D, H, W = v.shape
        lib().orc_cbca(_p(img), _p(v), D, H, W, ctypes.c_float(np.float32(intensity_threshold)),
                       int(distance_threshold), int(max_average_time))
        outs.append(v)
    return outs[0], outs[1]


def _sgm_scalars(sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D):
    c = ctypes.c_float
    return (c(np.float32(sgm_P1)), c(np.float32(sgm_P2)), c(np.float32(sgm_Q1)), c(np.float32(sgm_Q2)),
            c(np.float32(sgm_D)))


def semi_global_matching(left_image, right_image, cost_volume, r, sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D, choice):
    """In place on `cost_volume` (must be a C-contiguous float32 array) and returns it, like pf:544,568."""
    assert choice in ("L", "R")
    assert cost_volume.dtype == np.float32 and cost_volume.flags.c_contiguous
    il, ir = _img2d(left_image), _img2d(right_image)
    D, H, W = cost_volume.shape
    p1, p2, q1, q2, t = _sgm_scalars(sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D)
    lib().orc_sgm_pass(_p(il), _p(ir), _p(cost_volume), D, H, W, int(r[0]), int(r[1]), p1, p2, q1, q2, t,
                       0 if choice == "L" else 1)
    return cost_volume


def SGM_average(left_cost_volume, right_cost_volume, left_image, right_image,
                sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D, sgm_V):
    """pf:187-235.  Mutates both volume arguments (the reference does) and returns new arrays."""
    il, ir = _img2d(left_image), _img2d(right_image)
    p1, p2, q1, q2, t = _sgm_scalars(sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D)
    p1v = ctypes.c_float(np.float32(sgm_P1 / sgm_V))
    outs = []
    for side, vol in ((0, left_cost_volume), (1, right_cost_volume)):
        assert vol.dtype == np.float32 and vol.flags.c_contiguous
        D, H, W = vol.shape
        lib().orc_sgm_average(_p(il), _p(ir), _p(vol), D, H, W, p1, p1v, p2, q1, q2, t, side)
        outs.append(vol.copy())
    return outs[0], outs[1]


def disparity_prediction(left_cost_volume, right_cost_volume):
    outs = []
    for vol in (left_cost_volume, right_cost_volume):
        v = _f32(vol)
        D, H, W = v.shape
        d = np.empty((H, W), dtype=np.float32)
        lib().orc_wta(_p(v), D, H, W, _p(d))
        outs.append(d)
    return outs[0], outs[1]


def lr_status(left_disparity_map, right_disparity_map, ndisp):
    dl, dr = _f32(left_disparity_map), _f32(right_disparity_map)
    H, W = dl.shape
    st = np.empty((H, W), dtype=np.int32)
    lib().orc_lr_status(_p(dl), _p(dr), H, W, int(ndisp), _p(st, _i32p))
    return st


def interpolation(left_disparity_map, right_disparity_map, ndisp):
    dl, dr = _f32(left_disparity_map), _f32(right_disparity_map)
    H, W = dl.shape
    out = np.empty((H, W), dtype=np.float32)
    lib().orc_interpolation(_p(dl), _p(dr), H, W, int(ndisp), _p(out))
    return out


def subpixel_enhance(left_disparity_map, left_cost_volume):
    dl, v = _f32(left_disparity_map), _f32(left_cost_volume)
    D, H, W = v.shape
    out = np.empty((H, W), dtype=np.float32)
    lib().orc_subpixel(_p(dl), _p(v), D, H, W, _p(out))
    return out


def median_filter(left_disparity_map, filter_height, filter_width):
    dl = _f32(left_disparity_map)
    H, W = dl.shape
    out = np.empty((H, W), dtype=np.float32)
    lib().orc_median(_p(dl), H, W, int(filter_height), int(filter_width), _p(out))
    return out


def bilateral_table(filter_height, filter_width, mean, std_dev):
    """pf:428-436 with util.normal (util.py:45-48): float64 evaluation, float32 storage."""
    constant1 = 1. / (np.sqrt(2 * np.pi) * std_dev)
    constant2 = -1. / (2 * std_dev * std_dev)
    center_h = (filter_height - 1) // 2
    center_w = (filter_width - 1) // 2
    tab = np.zeros([filter_height, filter_width], dtype=np.float32)
    for h in range(filter_height):
        for w in range(filter_width):
            x = np.sqrt((h - center_h) ** 2 + (w - center_w) ** 2)
            tab[h, w] = constant1 * np.exp(constant2 * ((x - mean) ** 2))
    return tab


def bilateral_filter(left_image, left_disparity_map, filter_height, filter_width, mean, std_dev, blur_threshold):
    img = _img2d(left_image)
    dl = _f32(left_disparity_map)
    H, W = dl.shape
    tab = bilateral_table(filter_height, filter_width, mean, std_dev)
    out = np.empty((H, W), dtype=np.float32)
    lib().orc_bilateral(_p(img), _p(dl), H, W, int(filter_height), int(filter_width), _p(tab),
                        ctypes.c_float(np.float32(blur_threshold)), _p(out))
    return out


def match_pair(left_image, right_image, ndisp, layers, args=None, return_all=False):
    """The timed region of match.py:129-179 on standardised [H,W,1] images."""
    a = dict(cbca_intensity=0.02, cbca_distance=14, cbca_num_iterations1=2, cbca_num_iterations2=16,
             sgm_P1=2.3, sgm_P2=55.9, sgm_Q1=4, sgm_Q2=8, sgm_D=0.08, sgm_V=1.5, blur_sigma=6, blur_threshold=2,
             patch_size=2 * len(layers) + 1)
    if args:
        a.update(args)
    st = {}
    fl, fr = compute_features(left_image, right_image, a["patch_size"], a["patch_size"], layers)
    lcv, rcv = compute_cost_volume(fl, fr, ndisp)
    st["cost_volume"] = (lcv.copy(), rcv.copy()) if return_all else None
    lcv, rcv = cost_volume_aggregation(left_image, right_image, lcv, rcv, a["cbca_intensity"], a["cbca_distance"],
                                       a["cbca_num_iterations1"])
    st["cbca1"] = (lcv.copy(), rcv.copy()) if return_all else None
    lcv, rcv = SGM_average(lcv, rcv, left_image, right_image, a["sgm_P1"], a["sgm_P2"], a["sgm_Q1"], a["sgm_Q2"],
                           a["sgm_D"], a["sgm_V"])
    st["sgm"] = (lcv.copy(), rcv.copy()) if return_all else None
    lcv, rcv = cost_volume_aggregation(left_image, right_image, lcv, rcv, a["cbca_intensity"], a["cbca_distance"],
                                       a["cbca_num_iterations2"])
    st["cbca2"] = (lcv, rcv) if return_all else None
    dl, dr = disparity_prediction(lcv, rcv)
    st["wta"] = (dl, dr)
    di = interpolation(dl, dr, ndisp)
    st["interp"] = di
    ds = subpixel_enhance(di, lcv)
    st["subpixel"] = ds
    dm = median_filter(ds, 5, 5)
    st["median"] = dm
    db = bilateral_filter(left_image, dm, 5, 5, 0, a["blur_sigma"], a["blur_threshold"])
    st["bilateral"] = db
    return (db, st) if return_all else db

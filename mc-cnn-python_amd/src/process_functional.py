"""Drop-in for /root/reference/src/process_functional.py: the same eleven functions, same argument order, same
return values, computed on an MI355X through libmccnn_hip.so (include/mccnn.h).

    from process_functional import *          # exactly what the reference's match.py:13 does

Arrays may be NumPy (the reference's convention: float32, images [H,W,1], features [H,W,64], volumes [D,H,W],
maps [H,W]) or torch device tensors; the result comes back in the kind that went in.  NumPy inputs are copied to
HBM, processed, and copied back - use stereo_device.StereoMatcher to keep a whole pair resident instead.

There is no CPU fallback: without a HIP device or without the built library every function raises.

Options the reference does not have (module attributes; the DEFAULTS are the bit-exact variants, so that a drop-in
user gets the reference's numbers - the fast ones are opt-in and only tolerance-bounded):
    COST_VOLUME_MODE  "exact" (NumPy summation order, bit-exact, default) | "mfma" (matrix cores, <= 2e-6 abs)
    CBCA_ORDER        "reference" (flat list order, bit-exact, default: the pixel-major kernel, mccnn_cbca_iter_hwd)
                      | "reference_plane_major" (the same sums by the plane-major kernel of mccnn_cbca_iter - slower;
                        also what distances > 14 fall back to) | "separable" (fast, <= 1e-6 abs per iteration)
Opt-in departures from the reference's results (defaults reproduce it):
    CBCA_BOTH_VIEWS          False | True  - the paper's support regions intersected with the other view's (pf:122-144
                                             names it and skips it as impractical; pf:661-729 is its dead attempt)
    INTERPOLATION_DIRECTIONS 4 | 16        - mismatches filled from 16 rays as in the paper (pf:318)
    OCCLUSION_FROM_LEFT      False | True  - occlusions filled from the nearest match on the left, as in the paper (pf:361)
    NUMPY1_PROMOTION         False | True  - sub-pixel formula evaluated as NumPy < 2 promotes its scalars (float64,
                                             rounded once): what the reference's own Python 2.7 + NumPy 1.14 computes;
                                             the default is NumPy >= 2's float32 chain, which the golden vectors pin
"""
import numpy as np
import torch

import _hipabi as hip
import stereo_device as sd
from model import NET

FEATURES = "split_f16"       # "split_f16": hand-written matrix-core conv stack (float32-accurate); "library": MIOpen
COST_VOLUME_MODE = "exact"
CBCA_ORDER = "reference"
CBCA_BOTH_VIEWS = False
INTERPOLATION_DIRECTIONS = 4
OCCLUSION_FROM_LEFT = False
NUMPY1_PROMOTION = False

_CV_MODES = {"exact": hip.MCCNN_CV_EXACT, "mfma": hip.MCCNN_CV_MFMA}
_CBCA_ORDERS = {"separable": hip.MCCNN_CBCA_SEPARABLE, "reference": hip.MCCNN_CBCA_REFERENCE_ORDER,
                "reference_plane_major": hip.MCCNN_CBCA_REFERENCE_ORDER}

__all__ = ["compute_features", "compute_cost_volume", "cost_volume_aggregation", "SGM_average",
           "disparity_prediction", "interpolation", "subpixel_enhance", "median_filter", "bilateral_filter",
           "semi_global_matching", "compute_cross_region"]


def _dev(x, dtype=torch.float32):
    """-> (contiguous device tensor, came_from_numpy)"""
    device = hip.require_device()
    if torch.is_tensor(x):
        return x.to(device=device, dtype=dtype).contiguous(), False
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device), True


def _img(x):
    t, was_np = _dev(x)
    if t.dim() == 3:
        assert t.shape[2] == 1, "images are [H,W,1] (match.py:122-125)"
        t = t.reshape(t.shape[0], t.shape[1])
    return t.contiguous(), was_np


def _ret(t, was_np):
    return t.cpu().numpy() if was_np else t


_NET_CACHE = {}


def _net_for(checkpoint, patch_height):
    key = (checkpoint, patch_height)
    net = _NET_CACHE.get(key)
    if net is None:
        device = hip.require_device()
        net = NET(None, input_patch_size=patch_height, num_conv_layers=(patch_height - 1) // 2, batch_size=1,
                  device=device)
        net.restore(checkpoint)  # raises for None, like Saver.restore(sess, None) does (pf:43)
        _NET_CACHE.clear()
        _NET_CACHE[key] = net
    return net


def compute_features(left_image, right_image, patch_height, patch_width, checkpoint):
    """pf:15-73.  The weights stay resident between calls (the reference rebuilds graph + session per pair)."""
    assert patch_height == patch_width, "square patches only (match.py:61-62)"
    L, was_np = _img(left_image)
    R, _ = _img(right_image)
    net = checkpoint if isinstance(checkpoint, NET) else _net_for(checkpoint, patch_height)
    if FEATURES == "split_f16" and net.supports_split_features():
        # the hand-written matrix-core stack (float32-accurate; csrc/conv_mfma.hip); should an activation leave the
        # range of its stored records, the float32 library convolutions take the pair instead
        fl, fr = net.features_pair_hwc_split(L, R)
        if not net.split_saturated():
            return _ret(fl, was_np), _ret(fr, was_np)
    return _ret(net.features_hwc(L), was_np), _ret(net.features_hwc(R), was_np)


def compute_cost_volume(featuresl, featuresr, ndisp):
    """pf:78-113."""
    fl, was_np = _dev(featuresl)
    fr, _ = _dev(featuresr)
    lcv, rcv = sd.cost_volume(fl, fr, int(ndisp), _CV_MODES[COST_VOLUME_MODE])
    return _ret(lcv, was_np), _ret(rcv, was_np)


def compute_cross_region(image, intensity_threshold, distance_threshold):
    """pf:571-657: (union_region int32 [H,W,(2L)^2,2] padded with -1, union_region_num int32 [H,W]).
    Only for API compatibility / small images - the aggregation below never materialises the lists."""
    img, was_np = _img(image)
    support = sd.cross_arms(img, intensity_threshold, int(distance_threshold))
    region = sd.cross_region_list(support, int(distance_threshold))
    return _ret(region, was_np), _ret(sd.support_count(support).contiguous(), was_np)


def cost_volume_aggregation(left_image, right_image, left_cost_volume, right_cost_volume,
                            intensity_threshold, distance_threshold, max_average_time):
    """pf:117-183.  Inputs are not modified; fresh volumes are returned."""
    outs = []
    was_np = False
    images = [_img(left_image)[0], _img(right_image)[0]]
    supports = [sd.cross_arms(img, intensity_threshold, int(distance_threshold)) for img in images]
    if not CBCA_BOTH_VIEWS and CBCA_ORDER == "reference" and int(distance_threshold) <= 14:
        # What match.py's default runs: the reference's summation order on pixel-major copies, both views per launch,
        # through the program-driven assembly kernel (its programs are built once here and serve every iteration; from
        # the second iteration on the pixels whose support region is the pixel itself are left alone - same bits).
        vl, was_np = _dev(left_cost_volume)
        vr, _ = _dev(right_cost_volume)
        D, H, W = vl.shape
        progs = sd.cbca_prog_buffers(D, H, W, vl.device) if tuple(vr.shape) == (D, H, W) else None
        if progs is not None:
            sd.cbca_prog_build_pair(supports[0], supports[1], D, int(distance_threshold), progs,
                                    "both" if int(max_average_time) >= 2 else "full")
            hl, hr = sd.dhw_to_hwd(vl), sd.dhw_to_hwd(vr)       # copies: the caller's arrays stay as they are
            (rl, _), (rr, _) = sd.cbca_prog_pair(hl, torch.empty_like(hl), supports[0], hr, torch.empty_like(hr),
                                                 supports[1], progs, D, int(max_average_time), int(distance_threshold),
                                                 right_stream=sd.right_stream(hl.device))
            return _ret(sd.hwd_to_dhw(rl, D), was_np), _ret(sd.hwd_to_dhw(rr, D), was_np)
    for k, vol in enumerate((left_cost_volume, right_cost_volume)):
        v, was_np = _dev(vol)
        if torch.is_tensor(vol) and v.data_ptr() == vol.data_ptr():
            v = v.clone()  # the ping-pong clobbers its input; the reference leaves the caller's array alone
        if CBCA_BOTH_VIEWS:
            res, _spare = sd.cbca_both_views(v, torch.empty_like(v), supports[k], supports[1 - k], int(max_average_time),
                                             int(distance_threshold), hip.MCCNN_SIDE_LEFT if k == 0 else hip.MCCNN_SIDE_RIGHT)
        elif CBCA_ORDER == "reference" and int(distance_threshold) <= 14:
            # the reference's summation order on a pixel-major copy (disparities on lanes): bit-identical to the
            # plane-major reference-order kernel, several times faster
            D = v.shape[0]
            hv = sd.dhw_to_hwd(v)
            hres, _spare = sd.cbca_hwd(hv, torch.empty_like(hv), supports[k], D, int(max_average_time),
                                       int(distance_threshold))
            res = sd.hwd_to_dhw(hres, D)
        else:
            res, _spare = sd.cbca(v, torch.empty_like(v), supports[k], int(max_average_time), int(distance_threshold),
                                  _CBCA_ORDERS[CBCA_ORDER])
        outs.append(_ret(res, was_np))
    return outs[0], outs[1]


def semi_global_matching(left_image, right_image, cost_volume, r, sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D, choice):
    """pf:476-568.  Updates `cost_volume` IN PLACE and returns it, as the reference does (pf:544,568)."""
    assert choice == "R" or choice == "L"
    rh, rw = int(r[0]), int(r[1])
    assert rh * rw == 0
    L, _ = _img(left_image)
    R, _ = _img(right_image)
    v, was_np = _dev(cost_volume)
    D, H, W = v.shape
    hwd = sd.dhw_to_hwd(v)
    scratch = sd.sgm_scratch(H, W, D, v.device)
    f32 = sd._f32
    sd.sgm_pass_hwd(L, R, [hwd], [hip.MCCNN_SIDE_LEFT if choice == "L" else hip.MCCNN_SIDE_RIGHT], D, (rh, rw),
                    f32(sgm_P1), f32(sgm_P2), f32(sgm_Q1), f32(sgm_Q2), f32(sgm_D), scratch)
    sd.hwd_to_dhw(hwd, D, v)
    if was_np:
        cost_volume[...] = v.cpu().numpy()
    elif v.data_ptr() != cost_volume.data_ptr():
        cost_volume.copy_(v)
    return cost_volume


def SGM_average(left_cost_volume, right_cost_volume, left_image, right_image,
                sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D, sgm_V):
    """pf:187-235.  Like the reference, the two volume arguments end up holding the result as well (its four
    semi_global_matching calls run in place on them) and new arrays are returned."""
    L, _ = _img(left_image)
    R, _ = _img(right_image)
    vl, was_np = _dev(left_cost_volume)
    vr, _ = _dev(right_cost_volume)
    D, H, W = vl.shape
    hwd_shape = (H, W, sd.hwd_pitch(D))
    hl = torch.empty(hwd_shape, dtype=torch.float32, device=vl.device)
    hr = torch.empty(hwd_shape, dtype=torch.float32, device=vl.device)
    scratch = sd.sgm_scratch(H, W, D, vl.device)
    sd.sgm_average_from_dhw(L, R, [vl, vr], [hl, hr], [hip.MCCNN_SIDE_LEFT, hip.MCCNN_SIDE_RIGHT], D, sgm_P1, sgm_P2,
                            sgm_Q1, sgm_Q2, sgm_D, sgm_V, scratch)
    ol, orr = sd.hwd_to_dhw(hl, D), sd.hwd_to_dhw(hr, D)
    if was_np:
        ol_np, or_np = ol.cpu().numpy(), orr.cpu().numpy()
        left_cost_volume[...] = ol_np
        right_cost_volume[...] = or_np
        return ol_np.copy(), or_np.copy()
    left_cost_volume.copy_(ol)
    right_cost_volume.copy_(orr)
    return ol, orr


def disparity_prediction(left_cost_volume, right_cost_volume):
    """pf:239-272."""
    vl, was_np = _dev(left_cost_volume)
    vr, _ = _dev(right_cost_volume)
    return _ret(sd.wta(vl), was_np), _ret(sd.wta(vr), was_np)


def interpolation(left_disparity_map, right_disparity_map, ndisp):
    """pf:279-378."""
    dl, was_np = _dev(left_disparity_map)
    dr, _ = _dev(right_disparity_map)
    st = sd.lr_status(dl, dr, int(ndisp))
    return _ret(sd.interpolate(dl, st, directions=INTERPOLATION_DIRECTIONS, occlusion_from_left=OCCLUSION_FROM_LEFT),
                was_np)


def subpixel_enhance(left_disparity_map, left_cost_volume):
    """pf:381-400."""
    dl, was_np = _dev(left_disparity_map)
    v, _ = _dev(left_cost_volume)
    return _ret(sd.subpixel(dl, v, numpy1_promotion=NUMPY1_PROMOTION), was_np)


def median_filter(left_disparity_map, filter_height, filter_width):
    """pf:403-421."""
    dl, was_np = _dev(left_disparity_map)
    return _ret(sd.median(dl, int(filter_height), int(filter_width)), was_np)


def bilateral_filter(left_image, left_disparity_map, filter_height, filter_width, mean, std_dev, blur_threshold):
    """pf:424-470."""
    img, _ = _img(left_image)
    dl, was_np = _dev(left_disparity_map)
    return _ret(sd.bilateral(img, dl, int(filter_height), int(filter_width), mean, std_dev, blur_threshold), was_np)

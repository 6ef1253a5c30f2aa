"""Device-resident stages of the hot path: thin, allocation-aware wrappers over the C ABI (include/mccnn.h).

Everything here takes and returns torch device tensors; nothing touches host memory and nothing is computed by
PyTorch except the 5-layer conv stack (model.NET).  The NumPy-facing drop-in lives in process_functional.py and the
whole-pair driver (the timed region of the reference's match.py:129-179) is StereoMatcher.match below.
"""
import ctypes
import math

import numpy as np
import torch

import _hipabi as hip


def _f32(x):
    """The float32 rounding NumPy 2 applies when a Python scalar meets a float32 array."""
    return float(np.float32(x))


class StageTimer(object):
    """Optional per-stage HIP-event timing on the stream the kernels run on (bench.py turns it on)."""

    def __init__(self, enabled=False):
        self.enabled = enabled
        self.records = []  # (name, start_event, end_event)
        self._open = None

    def start(self, name):
        if not self.enabled:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._open = (name, ev)

    def stop(self):
        if not self.enabled or self._open is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((self._open[0], self._open[1], ev))
        self._open = None

    def span_start(self, name):
        """A bracket around several launches (a whole stage, possibly on several streams that fork from and join the
        current one): kept apart from the per-launch records."""
        if self.enabled:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.__dict__.setdefault("spans", []).append([name, ev, None])

    def span_stop(self, name):
        if self.enabled:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            for s in reversed(self.__dict__.get("spans", [])):
                if s[0] == name and s[2] is None:
                    s[2] = ev
                    break

    def spans_ms(self):
        out = {}
        for name, a, b in self.__dict__.get("spans", []):
            if b is not None:
                out.setdefault(name, []).append(a.elapsed_time(b))
        return out

    def summary_ms(self):
        """{name: [ms, ...]} - call after torch.cuda.synchronize()."""
        out = {}
        for name, a, b in self.records:
            out.setdefault(name, []).append(a.elapsed_time(b))
        return out


_NO_TIMER = StageTimer(False)


# ---- a1 ----------------------------------------------------------------------------------------------------------
def bias_act_(x, bias, relu):
    """In place on a contiguous NCHW tensor: x = relu(x + bias[c]) (or the bias alone) in one pass (model.py:118-123)."""
    N, C, H, W = x.shape
    assert x.is_contiguous() and bias.numel() == C
    hip.check(hip.load().mccnn_bias_act(hip.ptr(x), hip.ptr(bias), N, C, H * W, 1 if relu else 0, hip.stream()),
              "mccnn_bias_act")
    return x


def conv1_pad_bias_relu(images, weight, bias, pad):
    """images [N,H,W] -> [N,C,H+2pad-2,W+2pad-2]: zero padding + first 3x3 VALID conv (1 -> C maps) + bias + ReLU in one
    launch (pf:20-25, model.py:51-53).  weight: torch layout [C,1,3,3]."""
    N, H, W = images.shape
    C = weight.shape[0]
    assert tuple(weight.shape) == (C, 1, 3, 3) and images.is_contiguous() and weight.is_contiguous()
    out = torch.empty((N, C, H + 2 * pad - 2, W + 2 * pad - 2), dtype=torch.float32, device=images.device)
    hip.check(hip.load().mccnn_conv1_pad_bias_relu(hip.ptr(images), hip.ptr(weight), hip.ptr(bias), hip.ptr(out), N, H,
                                                   W, int(pad), C, hip.stream()), "mccnn_conv1_pad_bias_relu")
    return out


def l2norm_chw_to_hwc(chw, bias=None, out=None):
    """[C,H,W] conv output (+ the last layer's bias) -> [H,W,C] unit feature vectors (model.py:64)."""
    C, H, W = chw.shape
    if out is None:
        out = torch.empty((H, W, C), dtype=torch.float32, device=chw.device)
    assert tuple(out.shape) == (H, W, C) and out.is_contiguous()
    hip.check(hip.load().mccnn_l2norm_chw_to_hwc(hip.ptr(chw), hip.ptr(bias) if bias is not None else None,
                                                 hip.ptr(out), C, H, W, hip.stream()), "mccnn_l2norm_chw_to_hwc")
    return out


# ---- a1 on the matrix cores (opt-in): split-operand convolutions, csrc/conv_mfma.hip ---------------------------------
SPLIT_ACT_SCALE = 256.0   # stored activations carry this factor (power of two); they saturate at |x| = 65504 / 256


def conv3x3_split_pack(weight):
    """weight [64,64,3,3] (torch layout) -> (packed device buffer, weight_scale) for conv3x3_split.  Reads max |w|
    back to the host to choose the power-of-two scale: call once per weight set, not per image."""
    assert tuple(weight.shape) == (64, 64, 3, 3), "the split-operand kernel is built for 64 -> 64 maps, 3x3"
    w = weight.detach().contiguous().float()
    m = float(w.abs().max())
    scale = 2.0 ** math.floor(math.log2(1024.0 / m)) if m > 0.0 and math.isfinite(m) else 1.0
    lib = hip.load()
    packed = torch.empty((lib.mccnn_conv3x3_split_weights_bytes(),), dtype=torch.uint8, device=w.device)
    hip.check(lib.mccnn_conv3x3_split_pack(hip.ptr(w), scale, hip.ptr(packed), hip.stream()), "mccnn_conv3x3_split_pack")
    return packed, scale


def conv1_split(images, weight, bias, pad, act_scale=SPLIT_ACT_SCALE, sat_flag=None):
    """images [N,H,W] -> split records [N,H+2pad-2,W+2pad-2,256] (uint8 view): padding + layer 1 + bias + ReLU.
    sat_flag: int32 device tensor [1] that the kernel sets when an activation leaves the records' f16 range."""
    N, H, W = images.shape
    assert tuple(weight.shape) == (64, 1, 3, 3) and images.is_contiguous() and weight.is_contiguous()
    out = torch.empty((N, H + 2 * pad - 2, W + 2 * pad - 2, 256), dtype=torch.uint8, device=images.device)
    hip.check(hip.load().mccnn_conv1_split(hip.ptr(images), hip.ptr(weight), hip.ptr(bias), hip.ptr(out), N, H, W,
                                           int(pad), float(act_scale), hip.ptr(sat_flag) if sat_flag is not None else None,
                                           hip.stream()), "mccnn_conv1_split")
    return out


def conv3x3_split(x, packed, weight_scale, bias, last, act_scale=SPLIT_ACT_SCALE, sat_flag=None):
    """x: split records [N,Hi,Wi,256] -> VALID 3x3 conv + bias; last=False: ReLU, records [N,Hi-2,Wi-2,256];
    last=True: L2-normalised float32 features [N,Hi-2,Wi-2,64]."""
    N, Hi, Wi, rec = x.shape
    assert rec == 256 and x.dtype == torch.uint8 and x.is_contiguous()
    if last:
        out = torch.empty((N, Hi - 2, Wi - 2, 64), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((N, Hi - 2, Wi - 2, 256), dtype=torch.uint8, device=x.device)
    hip.check(hip.load().mccnn_conv3x3_split(hip.ptr(x), hip.ptr(packed), hip.ptr(bias), hip.ptr(out), N, Hi, Wi,
                                             float(weight_scale), float(act_scale), 1 if last else 0,
                                             hip.ptr(sat_flag) if sat_flag is not None else None, hip.stream()),
              "mccnn_conv3x3_split")
    return out


# ---- a2 ----------------------------------------------------------------------------------------------------------
def cost_volume(fl, fr, ndisp, mode=hip.MCCNN_CV_EXACT, out=None):
    H, W, C = fl.shape
    if out is None:
        lcv = torch.empty((ndisp, H, W), dtype=torch.float32, device=fl.device)
        rcv = torch.empty((ndisp, H, W), dtype=torch.float32, device=fl.device)
    else:
        lcv, rcv = out
    hip.check(hip.load().mccnn_cost_volume(hip.ptr(fl), hip.ptr(fr), H, W, C, int(ndisp), hip.ptr(lcv), hip.ptr(rcv),
                                           int(mode), hip.stream()), "mccnn_cost_volume")
    return lcv, rcv


def cost_volume_hwd(fl, fr, ndisp, out=None, mode=hip.MCCNN_CV_EXACT):
    """cost_volume(mode) written straight into pixel-major volumes [H,W,Dp] (mccnn_cost_volume_hwd)."""
    H, W, C = fl.shape
    dp = hwd_pitch(ndisp)
    if out is None:
        lcv = torch.zeros((H, W, dp), dtype=torch.float32, device=fl.device)
        rcv = torch.zeros((H, W, dp), dtype=torch.float32, device=fl.device)
    else:
        lcv, rcv = out
    hip.check(hip.load().mccnn_cost_volume_hwd(hip.ptr(fl), hip.ptr(fr), H, W, C, int(ndisp), hip.ptr(lcv), hip.ptr(rcv),
                                               int(mode), hip.stream()), "mccnn_cost_volume_hwd")
    return lcv, rcv


# ---- a3 ----------------------------------------------------------------------------------------------------------
def support_buffer(H, W, device):
    """An empty support plane (see cross_arms): the [H,W] view of a mccnn_support_bytes(H, W) allocation."""
    nbytes = int(hip.load().mccnn_support_bytes(H, W))
    buf = torch.empty(((nbytes + 3) // 4,), dtype=torch.int32, device=device)
    return buf[:H * W].view(H, W)


def cross_arms(image, intensity_threshold, distance_threshold, out=None):
    """image [H,W] -> support plane, int32 [H,W]: one packed word per pixel (mccnn_support_t: bits 0-4 up, 5-9 down,
    10-14 left, 15-19 right, 20-31 region size).  support_arms()/support_count() decode it.  The returned tensor is
    a view of the first plane of a mccnn_support_bytes(H, W) buffer; the derived planes (the words the streaming
    CBCA kernel reads) live behind it in the same storage and travel with the view.  `out`: a support_buffer()."""
    H, W = image.shape
    support = out if out is not None else support_buffer(H, W, image.device)
    hip.check(hip.load().mccnn_cross_arms(hip.ptr(image), H, W, _f32(intensity_threshold), int(distance_threshold),
                                          hip.ptr(support), hip.stream()), "mccnn_cross_arms")
    return support


def cross_arms_pair(image_l, image_r, intensity_threshold, distance_threshold, out_l=None, out_r=None):
    """cross_arms() on both views in one call (half the launches); returns (support_l, support_r)."""
    H, W = image_l.shape
    if tuple(image_r.shape) != (H, W):
        raise ValueError("cross_arms_pair: the two images must have the same shape")
    sl = out_l if out_l is not None else support_buffer(H, W, image_l.device)
    sr = out_r if out_r is not None else support_buffer(H, W, image_l.device)
    hip.check(hip.load().mccnn_cross_arms_pair(hip.ptr(image_l), hip.ptr(image_r), H, W, _f32(intensity_threshold),
                                               int(distance_threshold), hip.ptr(sl), hip.ptr(sr), hip.stream()),
              "mccnn_cross_arms_pair")
    return sl, sr


def support_arms(support):
    """uint8 [H,W,4]: up, down, left, right."""
    s = support.to(torch.int64) & 0xFFFFFFFF
    return torch.stack([(s >> sh) & 31 for sh in (0, 5, 10, 15)], dim=-1).to(torch.uint8)


def support_count(support):
    """int32 [H,W] region sizes (the reference's union_region_num)."""
    return (((support.to(torch.int64) & 0xFFFFFFFF) >> 20) & 0xFFF).to(torch.int32)


def cross_region_list(support, distance_threshold):
    H, W = support.shape
    L = int(distance_threshold)
    region = torch.empty((H, W, (2 * L) ** 2, 2), dtype=torch.int32, device=support.device)
    hip.check(hip.load().mccnn_cross_region_list(hip.ptr(support), H, W, L, hip.ptr(region), hip.stream()),
              "mccnn_cross_region_list")
    return region


# ---- a4 ----------------------------------------------------------------------------------------------------------
def cbca(vol, tmp, support, iterations, distance_threshold, order=hip.MCCNN_CBCA_SEPARABLE, timer=None):
    """`iterations` rounds of cross-based averaging.  Ping-pongs between `vol` and `tmp` (same shape);
    returns (result, spare) - the input buffer is clobbered when iterations >= 2, the reference's is not, so
    callers that need the input keep their own copy."""
    D, H, W = vol.shape
    lib = hip.load()
    have = support.untyped_storage().nbytes() - support.storage_offset() * support.element_size()
    if tuple(support.shape) != (H, W) or not support.is_contiguous() or have < lib.mccnn_support_bytes(H, W):
        raise ValueError("cbca: `support` must be the tensor cross_arms() returned (a copy drops its derived planes)")
    src, dst = vol, tmp
    timer = timer or _NO_TIMER
    for _ in range(int(iterations)):
        timer.start("cbca_iter")
        hip.check(lib.mccnn_cbca_iter(hip.ptr(src), hip.ptr(dst), hip.ptr(support), D, H, W,
                                      int(distance_threshold), int(order), hip.stream()), "mccnn_cbca_iter")
        timer.stop()
        src, dst = dst, src
    return src, dst


def cbca_pair(vol_l, tmp_l, support_l, vol_r, tmp_r, support_r, iterations, distance_threshold,
              order=hip.MCCNN_CBCA_SEPARABLE, timer=None):
    """cbca() on the left and the right volume together: every iteration is ONE launch that deals the work items of
    both volumes from one pool (mccnn_cbca_iter_pair; same results as two cbca() calls, fewer and fuller rounds of
    workgroups).  Returns ((result_l, spare_l), (result_r, spare_r))."""
    D, H, W = vol_l.shape
    if tuple(vol_r.shape) != (D, H, W):
        raise ValueError("cbca_pair: the two volumes must have the same shape")
    lib = hip.load()
    for sup in (support_l, support_r):
        have = sup.untyped_storage().nbytes() - sup.storage_offset() * sup.element_size()
        if tuple(sup.shape) != (H, W) or not sup.is_contiguous() or have < lib.mccnn_support_bytes(H, W):
            raise ValueError("cbca_pair: `support` must be the tensor cross_arms() returned")
    (sl, dl), (sr, dr) = (vol_l, tmp_l), (vol_r, tmp_r)
    timer = timer or _NO_TIMER
    for _ in range(int(iterations)):
        timer.start("cbca_iter_pair")
        hip.check(lib.mccnn_cbca_iter_pair(hip.ptr(sl), hip.ptr(dl), hip.ptr(support_l), hip.ptr(sr), hip.ptr(dr),
                                           hip.ptr(support_r), D, H, W, int(distance_threshold), int(order),
                                           hip.stream()), "mccnn_cbca_iter_pair")
        timer.stop()
        sl, dl, sr, dr = dl, sl, dr, sr
    return (sl, dl), (sr, dr)


def _check_support(support, H, W, who):
    have = support.untyped_storage().nbytes() - support.storage_offset() * support.element_size()
    if tuple(support.shape) != (H, W) or not support.is_contiguous() or have < hip.load().mccnn_support_bytes(H, W):
        raise ValueError("%s: `support` must be the tensor cross_arms() returned (a copy drops its derived planes)" % who)


def cbca_hwd(vol, tmp, support, D, iterations, distance_threshold, timer=None):
    """`iterations` rounds of cross-based averaging in the reference's summation order (bit-exact) on a pixel-major
    volume [H,W,Dp] (mccnn_cbca_iter_hwd).  Same ping-pong contract as cbca(): returns (result, spare)."""
    H, W, Dp = vol.shape
    assert Dp == hwd_pitch(D) and tuple(tmp.shape) == (H, W, Dp)
    _check_support(support, H, W, "cbca_hwd")
    lib = hip.load()
    src, dst = vol, tmp
    timer = timer or _NO_TIMER
    for _ in range(int(iterations)):
        timer.start("cbca_iter_hwd")
        hip.check(lib.mccnn_cbca_iter_hwd(hip.ptr(src), hip.ptr(dst), hip.ptr(support), int(D), H, W,
                                          int(distance_threshold), hip.stream()), "mccnn_cbca_iter_hwd")
        timer.stop()
        src, dst = dst, src
    return src, dst


def cbca_hwd_wta_max_d():
    """Largest D whose last aggregation iteration can carry the WTA (one chunk of disparities per wave)."""
    return 256


def cbca_hwd_pair(vol_l, tmp_l, support_l, vol_r, tmp_r, support_r, D, iterations, distance_threshold, timer=None,
                  wta_out=None, store_right=True):
    """cbca_hwd() on the left and the right volume, one launch per iteration (mccnn_cbca_iter_hwd_pair).
    Returns ((result_l, spare_l), (result_r, spare_r)).  wta_out = (disp_l, disp_r) [H,W] float32: the last iteration
    also writes the WTA disparities of both results (mccnn_cbca_iter_hwd_pair_wta; D <= cbca_hwd_wta_max_d());
    store_right=False then leaves the right result volume unwritten (its returned tensor holds stale data)."""
    H, W, Dp = vol_l.shape
    assert Dp == hwd_pitch(D)
    for t in (tmp_l, vol_r, tmp_r):
        if tuple(t.shape) != (H, W, Dp):
            raise ValueError("cbca_hwd_pair: the volumes must have the same shape")
    _check_support(support_l, H, W, "cbca_hwd_pair")
    _check_support(support_r, H, W, "cbca_hwd_pair")
    lib = hip.load()
    (sl, dl), (sr, dr) = (vol_l, tmp_l), (vol_r, tmp_r)
    timer = timer or _NO_TIMER
    n = int(iterations)
    if wta_out is not None and (n < 1 or D > cbca_hwd_wta_max_d()):
        raise ValueError("cbca_hwd_pair: the fused WTA needs at least one iteration and D <= %d" % cbca_hwd_wta_max_d())
    for it in range(n):
        timer.start("cbca_iter_hwd_pair")
        if wta_out is not None and it == n - 1:
            for t in wta_out:
                if tuple(t.shape) != (H, W) or t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("cbca_hwd_pair: wta_out must be two contiguous float32 [H,W] tensors")
            hip.check(lib.mccnn_cbca_iter_hwd_pair_wta(hip.ptr(sl), hip.ptr(dl), hip.ptr(support_l), hip.ptr(sr),
                                                       hip.ptr(dr), hip.ptr(support_r), int(D), H, W,
                                                       int(distance_threshold), hip.ptr(wta_out[0]), hip.ptr(wta_out[1]),
                                                       1 if store_right else 0, hip.stream()),
                      "mccnn_cbca_iter_hwd_pair_wta")
        else:
            hip.check(lib.mccnn_cbca_iter_hwd_pair(hip.ptr(sl), hip.ptr(dl), hip.ptr(support_l), hip.ptr(sr), hip.ptr(dr),
                                                   hip.ptr(support_r), int(D), H, W, int(distance_threshold),
                                                   hip.stream()), "mccnn_cbca_iter_hwd_pair")
        timer.stop()
        sl, dl, sr, dr = dl, sl, dr, sr
    return (sl, dl), (sr, dr)


def cbca_prog_buffers(D, H, W, device):
    """Two program buffers (left, right image) for cbca_prog_pair, or None when the shape is outside what the
    program-driven kernels encode (mccnn_cbca_prog_bytes == 0): the caller then stays with cbca_hwd_pair."""
    n = int(hip.load().mccnn_cbca_prog_bytes(int(D), int(H), int(W)))
    if n == 0:
        return None
    return (torch.empty((n // 4,), dtype=torch.int32, device=device), torch.empty((n // 4,), dtype=torch.int32, device=device))


def cbca_prog_build_pair(support_l, support_r, D, distance_threshold, progs, which="both"):
    """Compiles both images' support regions into the per-patch programs of the assembly aggregation kernel: once per
    pair, after cross_arms_pair, for all iterations.  which: "full" (mccnn_cbca_prog_build_pair: what every iteration can
    run), "skip" (mccnn_cbca_prog_build_skip_pair: the second set in the same buffers, which second and later iterations
    run, cbca_prog_pair), "both" (mccnn_cbca_prog_build_both_pair: the two sets from one pass over the support words, one
    launch of about the time of either of the others) or "both_two_launches" (the first two one after the other)."""
    H, W = support_l.shape
    _check_support(support_l, H, W, "cbca_prog_build_pair")
    _check_support(support_r, H, W, "cbca_prog_build_pair")
    if which not in ("full", "skip", "both", "both_two_launches"):
        raise ValueError("cbca_prog_build_pair: which must be 'full', 'skip', 'both' or 'both_two_launches'")
    lib = hip.load()
    if which == "both":       # one launch, one pass over the support words for the two sets
        hip.check(lib.mccnn_cbca_prog_build_both_pair(hip.ptr(support_l), hip.ptr(support_r), int(D), H, W,
                                                      int(distance_threshold), hip.ptr(progs[0]), hip.ptr(progs[1]),
                                                      hip.stream()), "mccnn_cbca_prog_build_both_pair")
        return progs
    for name, fn in (("full", lib.mccnn_cbca_prog_build_pair), ("skip", lib.mccnn_cbca_prog_build_skip_pair)):
        if which in (name, "both_two_launches"):
            hip.check(fn(hip.ptr(support_l), hip.ptr(support_r), int(D), H, W, int(distance_threshold), hip.ptr(progs[0]),
                         hip.ptr(progs[1]), hip.stream()), "mccnn_cbca_prog_build_%spair" % ("skip_" if name == "skip" else ""))
    return progs


_RIGHT_STREAMS = {}


def right_stream(device):
    """The stream the right volume's chain of aggregation launches runs on (cbca_prog_pair, right_stream=...)."""
    key = (device.type, device.index)
    if key not in _RIGHT_STREAMS:
        _RIGHT_STREAMS[key] = torch.cuda.Stream(device=device)
    return _RIGHT_STREAMS[key]


def skip_schedule(n, fused_last, skip_unit_regions=True, refresh_first=True):
    """Which kernel every iteration of an n-iteration aggregation runs: a list of "full", "refresh" (the full kernel that
    also writes unit-region pixels back into its input, mccnn_cbca_iter_prog_refresh), "skip" and "wta" (the full kernel
    fused with a7; fused_last).  cbca_prog_pair's docstring has the argument."""
    n = int(n)
    kinds = []
    for it in range(n):
        if fused_last and it == n - 1:
            kinds.append("wta")
        elif not skip_unit_regions or it == 0:
            kinds.append("full")
        elif refresh_first or not (n % 2 == 0 and it == n - 1):
            kinds.append("skip")
        else:
            kinds.append("full")
    # The first iteration's input buffer X keeps v0 at unit-region pixels.  As operands v0 and v1 are interchangeable, so
    # X matters only as the FINAL buffer: n even, and then only if the last iteration is a skip launch (a full / WTA
    # launch rewrites every pixel).  Only then does the first iteration have to refresh X.
    if n >= 2 and n % 2 == 0 and kinds[-1] == "skip":
        kinds[0] = "refresh"
    return kinds


def cbca_prog_pair(vol_l, tmp_l, support_l, vol_r, tmp_r, support_r, progs, D, iterations, distance_threshold, timer=None,
                   wta_out=None, store_right=True, skip_unit_regions=True, skip_ready=None, right_stream=None,
                   refresh_first=True):
    """cbca_hwd_pair's result (bit for bit) through the program-driven assembly kernel (mccnn_cbca_iter_prog_pair);
    `progs` from cbca_prog_build_pair on the same support buffers and D.  Same ping-pong contract; wta_out /
    store_right as in cbca_hwd_pair (mccnn_cbca_iter_prog_pair_wta for the last iteration).

    skip_unit_regions: from the SECOND iteration on, pixels whose support region is the pixel itself are left alone
    (mccnn_cbca_iter_prog_pair_skip).  Such a pixel gets v1 = (0 + v0) / 1 in the first iteration (pf:156-161 with
    aver_num = 1), which is v0 except that -0.0 becomes +0.0 and a signalling NaN is quieted, and (0 + v1) / 1 = v1 bit
    for bit ever after.  The first iteration (full programs) puts v1 into the partner buffer; with refresh_first
    (default, round 6) and an even number of iterations that would otherwise have to end with a full launch, it is a
    mccnn_cbca_iter_prog_refresh launch, which also writes v1 back into the input buffer, so that BOTH buffers
    hold the final value of these pixels and every later iteration may skip them, whichever buffer it writes: as a
    NEIGHBOUR in somebody else's region v0 and v1 give the same sum bit for bit (a running sum that started as 0 + x is
    never -0.0, so adding -0.0 or +0.0 cannot differ; a NaN operand gives the same quiet NaN either way), and nothing
    else reads them - except the caller, from the final buffer.  refresh_first=False is round 5's rule: the input buffer
    keeps v0, so with an even number of iterations - the result lands in the input buffer - the last iteration runs the
    full programs and rewrites every pixel.  The iteration that carries the WTA runs the full programs either way.  Same
    bits everywhere, fewer bytes moved.  NOTE: a refresh launch modifies the INPUT volume (v0 -> v1 at unit-region
    pixels); skip_schedule issues one only for an even number of iterations without a fused WTA - where the input buffer
    is also the result buffer, which holds v1 there in the end either way.
    skip_ready: an event after which the second program set is complete when it was built on another stream (the
    current stream waits for it in front of the first iteration that needs it).

    right_stream: when given, the two volumes run as two independent chains of ONE-volume launches
    (mccnn_cbca_iter_prog / _skip) - the left chain on the current stream, the right chain on `right_stream`, forked
    from the current stream here and joined to it before returning (or before the WTA-carrying last launch, which stays
    one two-volume launch).  The volumes never meet, so nothing orders the chains against each other, and one chain's
    launch fills the compute units the other's leaves idle while its heaviest patches finish: the same bits, 8.7 % less
    time for 16 iterations at 750x500x256 (profiles/r05_cbca_two_streams.txt)."""
    H, W, Dp = vol_l.shape
    assert Dp == hwd_pitch(D)
    for t in (tmp_l, vol_r, tmp_r):
        if tuple(t.shape) != (H, W, Dp):
            raise ValueError("cbca_prog_pair: the volumes must have the same shape")
    _check_support(support_l, H, W, "cbca_prog_pair")
    _check_support(support_r, H, W, "cbca_prog_pair")
    lib = hip.load()
    (sl, dl), (sr, dr) = (vol_l, tmp_l), (vol_r, tmp_r)
    timer = timer or _NO_TIMER
    n = int(iterations)
    if wta_out is not None and (n < 1 or D > cbca_hwd_wta_max_d()):
        raise ValueError("cbca_prog_pair: the fused WTA needs at least one iteration and D <= %d" % cbca_hwd_wta_max_d())
    kinds = skip_schedule(n, wta_out is not None, skip_unit_regions, refresh_first)

    def skips(it):
        return kinds[it] == "skip"

    n_chain = 0
    if right_stream is not None:
        # ---- two chains of one-volume launches (every iteration but a WTA-carrying last one) ----
        n_chain = n - 1 if wta_out is not None else n
        main = torch.cuda.current_stream()
        right_stream.wait_stream(main)
        chains = ((main, vol_l, tmp_l, support_l, progs[0]), (right_stream, vol_r, tmp_r, support_r, progs[1]))
        ends = []
        for st, src, dst, sup, prog in chains:
            with torch.cuda.stream(st):
                waited_ = False
                for it in range(n_chain):
                    skip = skips(it)
                    if skip and skip_ready is not None and not waited_:
                        st.wait_event(skip_ready)
                        waited_ = True
                    fn, who = ((lib.mccnn_cbca_iter_prog_skip, "mccnn_cbca_iter_prog_skip") if skip
                               else (lib.mccnn_cbca_iter_prog_refresh, "mccnn_cbca_iter_prog_refresh") if kinds[it] == "refresh"
                               else (lib.mccnn_cbca_iter_prog, "mccnn_cbca_iter_prog"))
                    timer.start("cbca_iter_prog_skip" if skip else "cbca_iter_prog")
                    hip.check(fn(hip.ptr(src), hip.ptr(dst), hip.ptr(sup), hip.ptr(prog), int(D), H, W,
                                 int(distance_threshold), hip.stream()), who)
                    timer.stop()
                    src, dst = dst, src
                ends.append((src, dst))
        main.wait_stream(right_stream)
        (sl, dl), (sr, dr) = ends
    waited = n_chain > 0 and any(skips(it) for it in range(n_chain))
    for it in range(n_chain, n):
        fused = wta_out is not None and it == n - 1
        skip = skips(it)
        timer.start("cbca_iter_prog_pair_skip" if skip else "cbca_iter_prog_pair")
        if fused:
            for t in wta_out:
                if tuple(t.shape) != (H, W) or t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("cbca_prog_pair: wta_out must be two contiguous float32 [H,W] tensors")
            hip.check(lib.mccnn_cbca_iter_prog_pair_wta(hip.ptr(sl), hip.ptr(dl), hip.ptr(support_l), hip.ptr(progs[0]),
                                                        hip.ptr(sr), hip.ptr(dr), hip.ptr(support_r), hip.ptr(progs[1]),
                                                        int(D), H, W, int(distance_threshold), hip.ptr(wta_out[0]),
                                                        hip.ptr(wta_out[1]), 1 if store_right else 0, hip.stream()),
                      "mccnn_cbca_iter_prog_pair_wta")
        else:
            fn, who = ((lib.mccnn_cbca_iter_prog_pair_skip, "mccnn_cbca_iter_prog_pair_skip") if skip
                       else (lib.mccnn_cbca_iter_prog_pair_refresh, "mccnn_cbca_iter_prog_pair_refresh")
                       if kinds[it] == "refresh" else (lib.mccnn_cbca_iter_prog_pair, "mccnn_cbca_iter_prog_pair"))
            if skip_ready is not None and skip and not waited:
                torch.cuda.current_stream().wait_event(skip_ready)
                waited = True
            hip.check(fn(hip.ptr(sl), hip.ptr(dl), hip.ptr(support_l), hip.ptr(progs[0]), hip.ptr(sr), hip.ptr(dr),
                         hip.ptr(support_r), hip.ptr(progs[1]), int(D), H, W, int(distance_threshold), hip.stream()), who)
        timer.stop()
        sl, dl, sr, dr = dl, sl, dr, sr
    return (sl, dl), (sr, dr)


def cbca_prog_chain(vol, tmp, support, prog, D, iterations, distance_threshold, first=0, total=None, fused_last=False,
                    skip_unit_regions=True, skip_ready=None, timer=None, refresh_first=True):
    """Iterations first .. iterations-1 of ONE volume's aggregation on the current stream (mccnn_cbca_iter_prog /
    _skip; cbca_prog_pair's rule for which iterations leave the unit-region pixels alone, with `total` = the length of
    the whole aggregation and fused_last = its last iteration carries the WTA and is not part of the chain)."""
    H, W, Dp = vol.shape
    lib = hip.load()
    n = int(total if total is not None else iterations)
    kinds = skip_schedule(n, fused_last, skip_unit_regions, refresh_first)
    src, dst = vol, tmp
    waited = False
    timer = timer or _NO_TIMER
    for it in range(int(first), int(iterations)):
        skip = kinds[it] == "skip"
        if skip and skip_ready is not None and not waited:
            torch.cuda.current_stream().wait_event(skip_ready)
            waited = True
        fn, who = ((lib.mccnn_cbca_iter_prog_skip, "mccnn_cbca_iter_prog_skip") if skip
                   else (lib.mccnn_cbca_iter_prog_refresh, "mccnn_cbca_iter_prog_refresh") if kinds[it] == "refresh"
                   else (lib.mccnn_cbca_iter_prog, "mccnn_cbca_iter_prog"))
        timer.start("cbca_iter_prog_skip" if skip else "cbca_iter_prog")
        hip.check(fn(hip.ptr(src), hip.ptr(dst), hip.ptr(support), hip.ptr(prog), int(D), H, W, int(distance_threshold),
                     hip.stream()), who)
        timer.stop()
        src, dst = dst, src
    return src, dst


def cbca_both_views(vol, tmp, support_self, support_other, iterations, distance_threshold, side, timer=None):
    """`iterations` rounds of cross-based averaging with the paper's two-view support regions (opt-in extra, see
    mccnn_cbca_iter_both): arms intersected with the other view's at the partner pixel x -/+ d.  Same ping-pong
    contract as cbca()."""
    D, H, W = vol.shape
    lib = hip.load()
    src, dst = vol, tmp
    timer = timer or _NO_TIMER
    for _ in range(int(iterations)):
        timer.start("cbca_iter_both")
        hip.check(lib.mccnn_cbca_iter_both(hip.ptr(src), hip.ptr(dst), hip.ptr(support_self), hip.ptr(support_other), D, H,
                                           W, int(distance_threshold), int(side), hip.stream()), "mccnn_cbca_iter_both")
        timer.stop()
        src, dst = dst, src
    return src, dst


# ---- a5 / a6 -------------------------------------------------------------------------------------------------------
def hwd_pitch(D):
    return hip.load().mccnn_hwd_pitch(int(D))


def dhw_to_hwd(dhw, hwd=None):
    D, H, W = dhw.shape
    if hwd is None:
        hwd = torch.empty((H, W, hwd_pitch(D)), dtype=torch.float32, device=dhw.device)
    hip.check(hip.load().mccnn_dhw_to_hwd(hip.ptr(dhw), hip.ptr(hwd), D, H, W, hip.stream()), "mccnn_dhw_to_hwd")
    return hwd


def hwd_to_dhw(hwd, D, dhw=None):
    H, W, _ = hwd.shape
    if dhw is None:
        dhw = torch.empty((D, H, W), dtype=torch.float32, device=hwd.device)
    hip.check(hip.load().mccnn_hwd_to_dhw(hip.ptr(hwd), hip.ptr(dhw), int(D), H, W, hip.stream()), "mccnn_hwd_to_dhw")
    return dhw


def sgm_scratch(H, W, D, device):
    n = hip.load().mccnn_sgm_scratch_bytes(H, W, int(D))
    return torch.empty((n,), dtype=torch.uint8, device=device)


def sgm_pass_hwd(image_left, image_right, vols_hwd, sides, D, r, p1, p2, q1, q2, thr, scratch):
    """One direction, in place on 1 or 2 HWD volumes.  p1..thr are already float32-rounded Python floats."""
    H, W = image_left.shape
    n = len(vols_hwd)
    vol_arr = (ctypes.c_void_p * 2)(*([v.data_ptr() for v in vols_hwd] + [None] * (2 - n)))
    side_arr = (ctypes.c_int * 2)(*(list(sides) + [0] * (2 - n)))
    hip.check(hip.load().mccnn_sgm_pass(hip.ptr(image_left), hip.ptr(image_right), vol_arr, side_arr, n, int(D), H, W,
                                        int(r[0]), int(r[1]), p1, p2, q1, q2, thr, hip.ptr(scratch),
                                        scratch.numel(), hip.stream()), "mccnn_sgm_pass")


SGM_DIRECTIONS = ((0, 1), (0, -1), (-1, 0), (1, 0))  # right, left, up, bottom (pf:194-208)


def sgm_flag_planes(image_left, image_right, D, sgm_D, out=None):
    """The flag planes of all four directions (mccnn_sgm_flags: which pixels' intensity step along r reaches sgm_D,
    pf:504-533), one buffer per direction: they depend on the images only, so a pair builds them once - off the critical
    path - and every pass of both volumes reads them (sgm_average_hwd(flags=...))."""
    H, W = image_left.shape
    thr = _f32(sgm_D)
    planes = out if out is not None else [sgm_scratch(H, W, D, image_left.device) for _ in SGM_DIRECTIONS]
    for r, buf in zip(SGM_DIRECTIONS, planes):
        hip.check(hip.load().mccnn_sgm_flags(hip.ptr(image_left), hip.ptr(image_right), int(D), H, W, int(r[0]), int(r[1]),
                                             thr, hip.ptr(buf), buf.numel(), hip.stream()), "mccnn_sgm_flags")
    return planes


def sgm_pass_flagged_hwd(vols_hwd, sides, D, r, p1, p2, q1, q2, flags):
    """One direction, in place on 1 or 2 HWD volumes, with the direction's flag planes already built (sgm_flag_planes)."""
    H, W, _ = vols_hwd[0].shape
    n = len(vols_hwd)
    vol_arr = (ctypes.c_void_p * 2)(*([v.data_ptr() for v in vols_hwd] + [None] * (2 - n)))
    side_arr = (ctypes.c_int * 2)(*(list(sides) + [0] * (2 - n)))
    hip.check(hip.load().mccnn_sgm_pass_flagged(vol_arr, side_arr, n, int(D), H, W, int(r[0]), int(r[1]), p1, p2, q1, q2,
                                                hip.ptr(flags), flags.numel(), hip.stream()), "mccnn_sgm_pass_flagged")


def sgm_average_hwd(image_left, image_right, vols_hwd, sides, D, sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D, sgm_V,
                    scratch, timer=_NO_TIMER, flags=None):
    """SGM_average (pf:187-235) on HWD volumes: the four passes compose in place (the reference aliases one array,
    pf:544,568) and its '(a+b+c+d)/4.' of four aliases of that array is the identity in binary floating point
    (the CPU checker used by the tests evaluates it literally; the parity tests pin the equality).
    flags: sgm_flag_planes() of the same images, D and sgm_D - the passes then launch no flag kernels of their own
    (`scratch` is not used)."""
    p1h = _f32(sgm_P1)
    p1v = _f32(sgm_P1 / sgm_V)  # Python double division, rounded once (pf:204)
    p2, q1, q2, thr = _f32(sgm_P2), _f32(sgm_Q1), _f32(sgm_Q2), _f32(sgm_D)
    for i, r in enumerate(SGM_DIRECTIONS):
        # (a launch that advances ONE volume - the free-running chains of StereoMatcher - is priced apart from the
        # two-volume launch: half the bytes, and it runs beside whatever the other volume's chain is doing)
        timer.start("sgm_pass" if len(vols_hwd) == 2 else "sgm_pass_one_volume")
        if flags is not None:
            sgm_pass_flagged_hwd(vols_hwd, sides, D, r, p1h if r[0] == 0 else p1v, p2, q1, q2, flags[i])
        else:
            sgm_pass_hwd(image_left, image_right, vols_hwd, sides, D, r, p1h if r[0] == 0 else p1v, p2, q1, q2, thr,
                         scratch)
        timer.stop()


SGM_FIRST_PASS_MAX_D = 256  # mccnn_sgm_first_pass gathers one [256 d][16 w] tile per wave


def sgm_average_from_dhw(image_left, image_right, vols_dhw, vols_hwd, sides, D, sgm_P1, sgm_P2, sgm_Q1, sgm_Q2, sgm_D,
                         sgm_V, scratch, timer=_NO_TIMER):
    """SGM_average starting from plane-major volumes: the first direction (0,1) reads `vols_dhw` and writes the
    pixel-major `vols_hwd` (mccnn_sgm_first_pass: layout change fused into the pass), the other three run in place on
    `vols_hwd`.  For D > 256 the layout change is a separate launch."""
    H, W = image_left.shape
    p1h = _f32(sgm_P1)
    p1v = _f32(sgm_P1 / sgm_V)
    p2, q1, q2, thr = _f32(sgm_P2), _f32(sgm_Q1), _f32(sgm_Q2), _f32(sgm_D)
    n = len(vols_dhw)
    rest = SGM_DIRECTIONS
    if D <= SGM_FIRST_PASS_MAX_D:
        src = (ctypes.c_void_p * 2)(*([v.data_ptr() for v in vols_dhw] + [None] * (2 - n)))
        dst = (ctypes.c_void_p * 2)(*([v.data_ptr() for v in vols_hwd] + [None] * (2 - n)))
        side_arr = (ctypes.c_int * 2)(*(list(sides) + [0] * (2 - n)))
        timer.start("sgm_first_pass")
        hip.check(hip.load().mccnn_sgm_first_pass(hip.ptr(image_left), hip.ptr(image_right), src, dst, side_arr, n,
                                                  int(D), H, W, p1h, p2, q1, q2, thr, hip.ptr(scratch),
                                                  scratch.numel(), hip.stream()), "mccnn_sgm_first_pass")
        timer.stop()
        rest = SGM_DIRECTIONS[1:]
    else:
        timer.start("dhw_to_hwd")
        for a, b in zip(vols_dhw, vols_hwd):
            dhw_to_hwd(a, b)
        timer.stop()
    for r in rest:
        timer.start("sgm_pass")
        sgm_pass_hwd(image_left, image_right, vols_hwd, sides, D, r, p1h if r[0] == 0 else p1v, p2, q1, q2, thr,
                     scratch)
        timer.stop()


# ---- a7 .. a11 -----------------------------------------------------------------------------------------------------
def wta(vol, out=None):
    D, H, W = vol.shape
    disp = out if out is not None else torch.empty((H, W), dtype=torch.float32, device=vol.device)
    hip.check(hip.load().mccnn_wta(hip.ptr(vol), D, H, W, hip.ptr(disp), hip.stream()), "mccnn_wta")
    return disp


def wta_hwd(vol_hwd, D, out=None):
    """wta() on a pixel-major volume [H,W,Dp]."""
    H, W, Dp = vol_hwd.shape
    assert Dp == hwd_pitch(D)
    disp = out if out is not None else torch.empty((H, W), dtype=torch.float32, device=vol_hwd.device)
    hip.check(hip.load().mccnn_wta_hwd(hip.ptr(vol_hwd), int(D), H, W, hip.ptr(disp), hip.stream()), "mccnn_wta_hwd")
    return disp


def subpixel_hwd(dl, vol_hwd, D, out=None, numpy1_promotion=False):
    """subpixel() on a pixel-major volume [H,W,Dp]."""
    H, W, Dp = vol_hwd.shape
    assert Dp == hwd_pitch(D)
    out = out if out is not None else torch.empty_like(dl)
    hip.check(hip.load().mccnn_subpixel_hwd(hip.ptr(dl), hip.ptr(vol_hwd), int(D), H, W, 1 if numpy1_promotion else 0,
                                            hip.ptr(out), hip.stream()), "mccnn_subpixel_hwd")
    return out


def lr_status(dl, dr, ndisp, out=None):
    H, W = dl.shape
    st = out if out is not None else torch.empty((H, W), dtype=torch.int32, device=dl.device)
    hip.check(hip.load().mccnn_lr_status(hip.ptr(dl), hip.ptr(dr), H, W, int(ndisp), hip.ptr(st), hip.stream()),
              "mccnn_lr_status")
    return st


def interpolate(dl, status, out=None, directions=4, occlusion_from_left=False):
    """directions / occlusion_from_left: the paper's rules the reference leaves out (pf:318, pf:361), opt-in."""
    H, W = dl.shape
    out = out if out is not None else torch.empty_like(dl)
    if int(directions) == 4 and not occlusion_from_left:
        hip.check(hip.load().mccnn_interpolate(hip.ptr(dl), hip.ptr(status), H, W, hip.ptr(out), hip.stream()),
                  "mccnn_interpolate")
    else:
        hip.check(hip.load().mccnn_interpolate_ex(hip.ptr(dl), hip.ptr(status), H, W, int(directions),
                                                  1 if occlusion_from_left else 0, hip.ptr(out), hip.stream()),
                  "mccnn_interpolate_ex")
    return out


def subpixel(dl, vol, out=None, numpy1_promotion=False):
    """numpy1_promotion: evaluate pf:396 as NumPy < 2 promotes it (float64, rounded once) instead of in float32."""
    D, H, W = vol.shape
    out = out if out is not None else torch.empty_like(dl)
    if numpy1_promotion:
        hip.check(hip.load().mccnn_subpixel_ex(hip.ptr(dl), hip.ptr(vol), D, H, W, 1, hip.ptr(out), hip.stream()),
                  "mccnn_subpixel_ex")
    else:
        hip.check(hip.load().mccnn_subpixel(hip.ptr(dl), hip.ptr(vol), D, H, W, hip.ptr(out), hip.stream()),
                  "mccnn_subpixel")
    return out


def median(dl, fh, fw, out=None):
    H, W = dl.shape
    out = out if out is not None else torch.empty_like(dl)
    hip.check(hip.load().mccnn_median(hip.ptr(dl), H, W, int(fh), int(fw), hip.ptr(out), hip.stream()),
              "mccnn_median")
    return out


def bilateral_table(filter_height, filter_width, mean, std_dev):
    """The reference's spatial kernel (pf:428-436, util.normal util.py:45-48): float64 evaluation, float32 storage."""
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


_TABLES = {}   # (fh, fw, mean, std_dev, device) -> device table: built once, not per pair


def bilateral_table_device(fh, fw, mean, std_dev, device):
    key = (int(fh), int(fw), float(mean), float(std_dev), str(device))
    tab = _TABLES.get(key)
    if tab is None:
        tab = torch.from_numpy(bilateral_table(int(fh), int(fw), mean, std_dev)).to(device)
        _TABLES[key] = tab
    return tab


def bilateral(image, dl, fh, fw, mean, std_dev, blur_threshold, out=None):
    H, W = dl.shape
    tab = bilateral_table_device(fh, fw, mean, std_dev, dl.device)
    out = out if out is not None else torch.empty_like(dl)
    hip.check(hip.load().mccnn_bilateral(hip.ptr(image), hip.ptr(dl), H, W, int(fh), int(fw), hip.ptr(tab),
                                         _f32(blur_threshold), hip.ptr(out), hip.stream()), "mccnn_bilateral")
    return out


# ---- whole pair ----------------------------------------------------------------------------------------------------
DEFAULT_HP = dict(cbca_intensity=0.02, cbca_distance=14, cbca_num_iterations1=2, cbca_num_iterations2=16,
                  sgm_P1=2.3, sgm_P2=55.9, sgm_Q1=4, sgm_Q2=8, sgm_D=0.08, sgm_V=1.5, blur_sigma=6,
                  blur_threshold=2)  # match.py:32-43


class StereoMatcher(object):
    """The timed region of match.py:129-179 for one stereo pair, resident on one GPU.

    net: model.NET with weights loaded (kept resident; the reference re-restores per pair, pf:43).
    cv_mode / cbca_order select the bit-exact (default, like process_functional and match.py) or the fast,
    tolerance-bounded variant of those two stages.  NOTE (round 3 on): the defaults are the bit-exact variants
    (MCCNN_CV_EXACT, MCCNN_CBCA_REFERENCE_ORDER: 13.7 ms per Middlebury-half pair); callers that relied on the
    earlier default (the fast variants, 9.3 ms) must ask for MCCNN_CV_MFMA / MCCNN_CBCA_SEPARABLE explicitly.
    on_saturation: what match() / match_graph() do when the hand-written feature kernels report that an activation
    left the range of their stored records (|x| >= 255.9: the data was clamped, the features of that pair are not
    float32-accurate).  "fallback" (default): the flag is read after every pair - ONE host synchronisation per pair -
    and such a pair is matched again with the float32 library convolutions; "raise": RuntimeError instead; "ignore":
    nothing is read and nothing blocks - for callers that keep several pairs in flight and poll
    features_saturated() themselves (match.py, bench.py).
    """

    def __init__(self, net, hp=None, cv_mode=hip.MCCNN_CV_EXACT, cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER,
                 feature_tile_rows=None, extras=None, features="auto", layout="auto", cbca_kernel="auto",
                 on_saturation="fallback", skip_unit_regions=True, two_chains=True, one_launch_builder=True,
                 side_early=False, free_chains=True, refresh_first=True, sgm_flags_once=True):
        self.device = hip.require_device()
        self.net = net
        self.hp = dict(DEFAULT_HP)
        if hp:
            self.hp.update(hp)
        self.cv_mode = cv_mode
        self.cbca_order = cbca_order
        self.feature_tile_rows = feature_tile_rows   # None: whole image; else NET.features_pair_hwc's band height
        # "split_f16": the split-operand matrix-core kernels (conv_mfma.hip; float32-accurate since round 4: as close to
        # a float64 evaluation as the library); "miopen": float32 library convolutions; "auto" (default): the
        # hand-written kernels where the network has their topology (3x3, 64 maps, >= 2 layers) and no row banding
        # is asked for, the library otherwise
        if features not in ("auto", "miopen", "split_f16"):
            raise ValueError("features must be 'auto', 'miopen' or 'split_f16'")
        if features == "split_f16" and feature_tile_rows is not None:
            raise ValueError("row banding is implemented for the library convolutions only")
        if features == "auto":
            features = "split_f16" if (feature_tile_rows is None and net.supports_split_features()) else "miopen"
        self.features = features
        # "auto": the bit-exact variant runs on pixel-major volumes (pixel_major()); "plane_major" keeps every stage
        # on the reference's [D,H,W] layout (the round-2 kernels: cross-checks, and what the fast variant always uses)
        if layout not in ("auto", "plane_major"):
            raise ValueError("layout must be 'auto' or 'plane_major'")
        self.layout = layout
        # reference-order aggregation on pixel-major volumes: "prog" = the program-driven assembly kernel
        # (mccnn_cbca_iter_prog_pair), "hwd" = cbca_hwd_kernel; "auto": the first wherever its programs encode the shape
        if cbca_kernel not in ("auto", "prog", "hwd"):
            raise ValueError("cbca_kernel must be 'auto', 'prog' or 'hwd'")
        self.cbca_kernel = cbca_kernel
        if on_saturation not in ("fallback", "raise", "ignore"):
            raise ValueError("on_saturation must be 'fallback', 'raise' or 'ignore'")
        self.on_saturation = on_saturation
        self._library_twin = None
        # cbca_prog_pair's rule (iterations that leave unit-region pixels alone: same bits, fewer bytes); False runs the
        # full programs in every iteration - the content-independent cost of the aggregation (bench.py reports both)
        self.skip_unit_regions = bool(skip_unit_regions)
        # the program-driven aggregation as two chains of one-volume launches on two streams (cbca_prog_pair,
        # right_stream): same bits; False = one two-volume launch per iteration
        self.two_chains = bool(two_chains)
        # both program sets from one launch beside the cost volume (mccnn_cbca_prog_build_both_pair); False = round 4's
        # two launches, the skip programs beside the first aggregation (A/B measurements)
        self.one_launch_builder = bool(one_launch_builder)
        # options measured with tools/dev_ab_matchers.py (profiles/r05_ab_matcher_options.txt): side_early - the side
        # stream's work beside the conv stack instead of beside the cost volume (+0.04 ms: worse); free_chains (default
        # since round 6) - each volume's aggregation -> SGM -> aggregation as ONE free-running chain of one-volume
        # launches on its own stream, no join between the stages (-0.10 .. -0.16 ms per cfg2 pair, same bits); False =
        # round 5's schedule: the chains join after every stage and the SGM passes are two-volume launches
        self.side_early = bool(side_early)
        self.free_chains = bool(free_chains)
        # skip_schedule's rule (round 6): an aggregation with an even number of iterations whose last launch carries no WTA
        # (match.py's first: 2 iterations) starts with a refresh launch and then skips to the end; False = round 5's rule
        # (its last launch is a full one)
        self.refresh_first = bool(refresh_first)
        # the SGM flag planes of the four directions built once per pair on the side stream (mccnn_sgm_flags) and read by
        # the passes of both chains; False = every one-volume pass launches its own flag kernel (8 per pair, on the chains)
        self.sgm_flags_once = bool(sgm_flags_once)
        # opt-in departures from the reference (they change the output): the paper's rules it leaves out, and the
        # scalar promotion of the NumPy it was written for
        self.extras = dict(both_view_support=False, interpolation_directions=4, occlusion_from_left=False,
                           numpy1_promotion=False)
        if extras:
            unknown = set(extras) - set(self.extras)
            if unknown:
                raise ValueError("unknown extras: %s" % sorted(unknown))
            self.extras.update(extras)
        self._ws = {}
        self._graphs = {}
        self._side = None
        self._right = None

    def _right_stream(self):
        """The stream of the right volume's chain of aggregation launches: this matcher's own (like its side stream), so
        that a stream never meets the captures of two different graphs - a module-wide stream that had been part of
        the capture of a graph destroyed since crashed the runtime in a later capture once in three test runs."""
        if self._right is None:
            self._right = torch.cuda.Stream()
        return self._right

    def features_saturated(self, reset=True):
        """True when the split-operand feature kernels clamped an activation since the last reset (blocks the host)."""
        return self.net.split_saturated(reset)

    def workspace(self, H, W, D):
        key = (H, W, D)
        ws = self._ws.get(key)
        if ws is None:
            dp = hwd_pitch(D)
            n = H * W * dp                        # >= D*H*W: every volume buffer can hold either layout
            dev = self.device
            ws = dict(
                vol=[torch.empty((n,), dtype=torch.float32, device=dev) for _ in range(4)],
                scratch=sgm_scratch(H, W, D, dev),
                sgm_flags=[sgm_scratch(H, W, D, dev) for _ in SGM_DIRECTIONS],     # flag planes of the four directions
                sup_l=support_buffer(H, W, dev), sup_r=support_buffer(H, W, dev),
                status=torch.empty((H, W), dtype=torch.int32, device=dev),
                maps=torch.empty((6, H, W), dtype=torch.float32, device=dev),   # dl, dr, interp, subpixel, median, out
            )
            ws["progs"] = None
            if self.pixel_major() and self.cbca_kernel != "hwd":
                ws["progs"] = cbca_prog_buffers(D, H, W, dev)
                if ws["progs"] is None and self.cbca_kernel == "prog":
                    raise ValueError("cbca_kernel='prog': %dx%dx%d is outside what the aggregation programs encode" % (W, H, D))
            bilateral_table_device(5, 5, 0, self.hp["blur_sigma"], dev)
            self._ws = {key: ws}  # one shape resident at a time
            self._graphs = {}
        return ws

    def pixel_major(self):
        """True when the pair runs on pixel-major volumes from the first aggregation on: the bit-exact variant
        (reference-order CBCA, arms <= 13, no two-view regions).  CBCA, SGM, WTA and sub-pixel then all work on
        [H,W,Dp] and no layout change surrounds the SGM stage."""
        return (self.layout == "auto" and self.cbca_order == hip.MCCNN_CBCA_REFERENCE_ORDER
                and not self.extras["both_view_support"]
                and int(self.hp["cbca_distance"]) <= 14)

    def _saturated_pair(self, left_image, right_image, ndisp, out):
        """on_saturation for the pair that has just been launched: None when its features were fine (or nobody is to
        look), else the map of the same pair behind the float32 library convolutions (written to `out` if given)."""
        if self.features != "split_f16" or self.on_saturation == "ignore" or torch.cuda.is_current_stream_capturing():
            return None
        if not self.features_saturated():
            return None
        if self.on_saturation == "raise":
            raise RuntimeError("StereoMatcher: an activation left the range of the split-operand feature kernels "
                               "(|x| >= 255.9); match this pair with features='miopen'")
        if self._library_twin is None:
            self._library_twin = StereoMatcher(self.net, hp=self.hp, cv_mode=self.cv_mode, cbca_order=self.cbca_order,
                                               extras=self.extras, features="miopen", layout=self.layout,
                                               cbca_kernel=self.cbca_kernel, on_saturation="ignore")
        return self._library_twin.match(left_image, right_image, ndisp, out=out)

    def match(self, left_image, right_image, ndisp, timer=_NO_TIMER, keep=None, _static_out=False, out=None):
        """_match() + the on_saturation policy (class docstring): a pair whose hand-written features were clamped is
        matched again with the library convolutions unless the caller asked to be left alone."""
        res = self._match(left_image, right_image, ndisp, timer=timer, keep=keep, _static_out=_static_out, out=out)
        if keep is None and not _static_out:
            redo = self._saturated_pair(left_image, right_image, ndisp, out)
            if redo is not None:
                return redo
        return res

    def _match(self, left_image, right_image, ndisp, timer=_NO_TIMER, keep=None, _static_out=False, out=None):
        """left/right: standardised float32 device tensors [H,W] (or [H,W,1]).  Returns the final left disparity
        map [H,W] on the device.  The matcher's workspace is reused by the next call, so the map is handed out as a
        copy - one [H,W] allocation + one copy per pair; pass `out` (a contiguous float32 [H,W] device tensor) and the
        last kernel writes there instead: nothing is allocated but the conv activations.  `keep`, if a dict, receives
        intermediate device tensors in the reference's [D,H,W] layout (tests)."""
        hp = self.hp
        L = left_image.reshape(left_image.shape[0], left_image.shape[1]).contiguous()
        R = right_image.reshape(right_image.shape[0], right_image.shape[1]).contiguous()
        H, W = L.shape
        D = int(ndisp)
        ws = self.workspace(H, W, D)
        dhw = (D, H, W)
        hwd = (H, W, hwd_pitch(D))
        nd, nh = D * H * W, H * W * hwd[2]
        as_dhw = lambda buf: buf[:nd].view(dhw)       # noqa: E731
        as_hwd = lambda buf: buf[:nh].view(hwd)       # noqa: E731
        b0, b1, b2, b3 = ws["vol"]
        sides = [hip.MCCNN_SIDE_LEFT, hip.MCCNN_SIDE_RIGHT]

        # The support arms and the aggregation programs depend on the images only: without per-stage timing they run
        # on a side stream beside the cost volume - the arms and, since round 5, both program sets in one launch (round
        # 4 built the skip programs in a launch of their own beside the first aggregation and SGM).  Round-4
        # measurements at cfg2 (one box, 100 pairs each, twice): everything on the main stream 9.22 / 9.22 ms,
        # everything beside the conv stack 9.19 / 9.12 (the builder's waves slow the matrix-core kernels down by what
        # they save), beside the cost volume 9.14 / 9.06.
        overlap = timer is _NO_TIMER
        skip_ready = full_ready = sup_l = sup_r = flag_planes = None
        want_flags = (self.sgm_flags_once and self.free_chains and keep is None and ws["progs"] is not None
                      and self.two_chains and self.pixel_major())

        def side_work(stage):
            """stage 0: support arms + both program sets (or, one_launch_builder=False, the full programs; stage 1: the
            skip programs)."""
            nonlocal skip_ready, full_ready, sup_l, sup_r, flag_planes
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                if stage == 0:
                    sup_l, sup_r = cross_arms_pair(L, R, hp["cbca_intensity"], hp["cbca_distance"], ws["sup_l"], ws["sup_r"])
                    if ws["progs"] is not None:
                        # both program sets from one pass over the support words (round 5: one launch of about the
                        # time either of the two earlier launches took)
                        cbca_prog_build_pair(sup_l, sup_r, D, hp["cbca_distance"], ws["progs"],
                                             "both" if (self.skip_unit_regions and self.one_launch_builder) else "full")
                    if want_flags:
                        flag_planes = sgm_flag_planes(L, R, D, hp["sgm_D"], out=ws["sgm_flags"])
                    full_ready = torch.cuda.Event()
                    full_ready.record(self._side)
                elif ws["progs"] is not None and self.skip_unit_regions and not self.one_launch_builder:
                    cbca_prog_build_pair(sup_l, sup_r, D, hp["cbca_distance"], ws["progs"], "skip")
                    skip_ready = torch.cuda.Event()
                    skip_ready.record(self._side)

        if overlap and self.side_early:
            side_work(0)
        timer.start("features")
        if self.features == "split_f16":
            fl, fr = self.net.features_pair_hwc_split(L, R)
        else:
            fl, fr = self.net.features_pair_hwc(L, R, tile_rows=self.feature_tile_rows)
        timer.stop()
        if overlap and not self.side_early:
            side_work(0)

        # the bit-exact variant writes its cost volume pixel-major right away (nothing converts layouts after that)
        direct = self.pixel_major() and D <= 512
        timer.start("cost_volume")
        if direct:
            lh, rh = cost_volume_hwd(fl, fr, D, out=(as_hwd(b2), as_hwd(b3)), mode=self.cv_mode)
        else:
            lcv, rcv = cost_volume(fl, fr, D, self.cv_mode, out=(as_dhw(b0), as_dhw(b1)))
        timer.stop()
        del fl, fr
        if keep is not None:
            keep["cv"] = (hwd_to_dhw(lh, D), hwd_to_dhw(rh, D)) if direct else (lcv.clone(), rcv.clone())

        if overlap:
            torch.cuda.current_stream().wait_event(full_ready)
            if ws["progs"] is not None and not self.one_launch_builder:
                side_work(1)
        else:
            timer.start("cross_arms")
            sup_l, sup_r = cross_arms_pair(L, R, hp["cbca_intensity"], hp["cbca_distance"], ws["sup_l"], ws["sup_r"])
            timer.stop()
            if ws["progs"] is not None:
                timer.start("cbca_prog_build")
                cbca_prog_build_pair(sup_l, sup_r, D, hp["cbca_distance"], ws["progs"])
                timer.stop()

        ex = self.extras
        m = ws["maps"] if keep is None else torch.empty_like(ws["maps"])
        if self.pixel_major():
            # ---- bit-exact variant: pixel-major from here on ----
            if not direct:
                timer.start("cv_to_pixel_major")
                lh, rh = dhw_to_hwd(lcv, as_hwd(b2)), dhw_to_hwd(rcv, as_hwd(b3))
                timer.stop()
            progs = ws["progs"]

            def aggregate_hwd(lh, lt, rh, rt, n, **kw):
                # the program-driven assembly kernel where its programs exist, cbca_hwd_kernel otherwise (same bits)
                if progs is None:
                    return cbca_hwd_pair(lh, lt, sup_l, rh, rt, sup_r, D, int(n), hp["cbca_distance"], timer, **kw)
                return cbca_prog_pair(lh, lt, sup_l, rh, rt, sup_r, progs, D, int(n), hp["cbca_distance"], timer,
                                      skip_ready=skip_ready if overlap else None,
                                      skip_unit_regions=self.skip_unit_regions, refresh_first=self.refresh_first,
                                      right_stream=self._right_stream() if self.two_chains else None, **kw)

            # free-running chains (default): each volume's aggregation -> SGM -> aggregation is ONE chain of one-volume
            # launches on its own stream; the two chains meet again only in front of the WTA-carrying last launch.  (With
            # `keep` the stages are joined, so that both volumes of a stage can be handed out.)
            free = (self.free_chains and keep is None and progs is not None and self.two_chains)
            if free:
                n1, n2 = int(hp["cbca_num_iterations1"]), int(hp["cbca_num_iterations2"])
                fuse = n2 >= 1 and D <= cbca_hwd_wta_max_d()
                if flag_planes is None and self.sgm_flags_once:      # (per-stage timing: nothing ran beside the cost volume)
                    timer.start("sgm_flags")
                    flag_planes = sgm_flag_planes(L, R, D, hp["sgm_D"], out=ws["sgm_flags"])
                    timer.stop()
                if flag_planes is None and "scratch2" not in ws:
                    ws["scratch2"] = sgm_scratch(H, W, D, self.device)
                main_s, right_s = torch.cuda.current_stream(), self._right_stream()
                right_s.wait_stream(main_s)
                ends = []
                for st, v, t, sup, prog, side, scr in ((main_s, lh, as_hwd(b0), sup_l, progs[0], sides[0], ws["scratch"]),
                                                       (right_s, rh, as_hwd(b1), sup_r, progs[1], sides[1], ws.get("scratch2"))):
                    with torch.cuda.stream(st):
                        # (the brackets are recorded on the chain's own stream: a stage's span beside the other chain)
                        timer.span_start("aggregation_1")
                        v, t = cbca_prog_chain(v, t, sup, prog, D, n1, hp["cbca_distance"],
                                               skip_unit_regions=self.skip_unit_regions, refresh_first=self.refresh_first,
                                               skip_ready=skip_ready if overlap else None, timer=timer)
                        timer.span_stop("aggregation_1")
                        timer.span_start("sgm")
                        # (the flag planes of the four directions were built once, on the side stream beside the cost
                        # volume: both chains read them and launch no flag kernels of their own - 8 launches of 11 us
                        # off the two critical chains)
                        sgm_average_hwd(L, R, [v], [side], D, hp["sgm_P1"], hp["sgm_P2"], hp["sgm_Q1"], hp["sgm_Q2"],
                                        hp["sgm_D"], hp["sgm_V"], scr, timer, flags=flag_planes)
                        timer.span_stop("sgm")
                        timer.span_start("aggregation_2")
                        v, t = cbca_prog_chain(v, t, sup, prog, D, n2 - 1 if fuse else n2, hp["cbca_distance"], total=n2,
                                               fused_last=fuse, skip_unit_regions=self.skip_unit_regions,
                                               refresh_first=self.refresh_first,
                                               skip_ready=skip_ready if overlap else None, timer=timer)
                        timer.span_stop("aggregation_2")
                        ends.append((v, t))
                main_s.wait_stream(right_s)
                (lh, lt), (rh, rt) = ends
                if fuse:
                    timer.start("cbca_iter_prog_pair")
                    hip.check(hip.load().mccnn_cbca_iter_prog_pair_wta(
                        hip.ptr(lh), hip.ptr(lt), hip.ptr(sup_l), hip.ptr(progs[0]), hip.ptr(rh), hip.ptr(rt), hip.ptr(sup_r),
                        hip.ptr(progs[1]), int(D), H, W, int(hp["cbca_distance"]), hip.ptr(m[0]), hip.ptr(m[1]), 0,
                        hip.stream()), "mccnn_cbca_iter_prog_pair_wta")
                    timer.stop()
                    lh, lt, rh, rt = lt, lh, rt, rh
            if not free:
                timer.span_start("aggregation_1")
                (lh, lt), (rh, rt) = aggregate_hwd(lh, as_hwd(b0), rh, as_hwd(b1), hp["cbca_num_iterations1"])
                timer.span_stop("aggregation_1")
            if keep is not None:
                keep["cbca1"] = (hwd_to_dhw(lh, D), hwd_to_dhw(rh, D))
            if not free:
                timer.span_start("sgm")
                sgm_average_hwd(L, R, [lh, rh], sides, D, hp["sgm_P1"], hp["sgm_P2"], hp["sgm_Q1"], hp["sgm_Q2"],
                                hp["sgm_D"], hp["sgm_V"], ws["scratch"], timer)
                timer.span_stop("sgm")
            if keep is not None:
                keep["sgm"] = (hwd_to_dhw(lh, D), hwd_to_dhw(rh, D))
            # the last iteration carries the WTA of both results (and leaves the right volume, which nothing else
            # reads, unwritten) when a wave holds all disparities of a pixel
            fuse = int(hp["cbca_num_iterations2"]) >= 1 and D <= cbca_hwd_wta_max_d()
            if not free:
                timer.span_start("aggregation_2")
                (lh, lt), (rh, rt) = aggregate_hwd(lh, lt, rh, rt, hp["cbca_num_iterations2"],
                                                   wta_out=(m[0], m[1]) if fuse else None, store_right=keep is not None)
                timer.span_stop("aggregation_2")
            if overlap and skip_ready is not None:
                torch.cuda.current_stream().wait_event(skip_ready)      # joins the side stream whatever the iteration count
            if keep is not None:
                keep["cbca2"] = (hwd_to_dhw(lh, D), hwd_to_dhw(rh, D))
            if fuse:
                dl, dr = m[0], m[1]
            else:
                timer.start("wta")
                dl = wta_hwd(lh, D, out=m[0])
                dr = wta_hwd(rh, D, out=m[1])
                timer.stop()
            sub = lambda di: subpixel_hwd(di, lh, D, out=m[3], numpy1_promotion=ex["numpy1_promotion"])   # noqa: E731
        else:
            def aggregate(vol, tmp, own, other, n, side):
                if ex["both_view_support"]:
                    return cbca_both_views(vol, tmp, own, other, n, hp["cbca_distance"], side, timer)
                return cbca(vol, tmp, own, n, hp["cbca_distance"], self.cbca_order, timer)

            def aggregate_both(lcv, t1d, rcv, t2d, n):
                # the separable kernel takes both views in one launch; the other variants run view by view
                if not ex["both_view_support"] and self.cbca_order == hip.MCCNN_CBCA_SEPARABLE:
                    return cbca_pair(lcv, t1d, sup_l, rcv, t2d, sup_r, n, hp["cbca_distance"], self.cbca_order, timer)
                return (aggregate(lcv, t1d, sup_l, sup_r, n, hip.MCCNN_SIDE_LEFT),
                        aggregate(rcv, t2d, sup_r, sup_l, n, hip.MCCNN_SIDE_RIGHT))

            (lcv, t1d), (rcv, t2d) = aggregate_both(lcv, as_dhw(b2), rcv, as_dhw(b3), hp["cbca_num_iterations1"])
            if keep is not None:
                keep["cbca1"] = (lcv.clone(), rcv.clone())

            # SGM on pixel-major copies in the spare ping-pong buffers (every buffer holds either layout)
            lh, rh = t1d.reshape(-1), t2d.reshape(-1)
            lh = next(as_hwd(b) for b in ws["vol"] if b.data_ptr() == lh.data_ptr())
            rh = next(as_hwd(b) for b in ws["vol"] if b.data_ptr() == rh.data_ptr())
            sgm_average_from_dhw(L, R, [lcv, rcv], [lh, rh], sides, D, hp["sgm_P1"], hp["sgm_P2"], hp["sgm_Q1"],
                                 hp["sgm_Q2"], hp["sgm_D"], hp["sgm_V"], ws["scratch"], timer)
            timer.start("hwd_to_dhw")
            hwd_to_dhw(lh, D, lcv)
            hwd_to_dhw(rh, D, rcv)
            timer.stop()
            if keep is not None:
                keep["sgm"] = (lcv.clone(), rcv.clone())

            (lcv, t1d), (rcv, t2d) = aggregate_both(lcv, t1d, rcv, t2d, hp["cbca_num_iterations2"])
            if keep is not None:
                keep["cbca2"] = (lcv.clone(), rcv.clone())
            timer.start("wta")
            dl = wta(lcv, out=m[0])
            dr = wta(rcv, out=m[1])
            timer.stop()
            sub = lambda di: subpixel(di, lcv, out=m[3], numpy1_promotion=ex["numpy1_promotion"])   # noqa: E731

        # per-pair maps live in the workspace unless the caller keeps intermediates (tests): apart from the returned
        # map (see `out`) a pair then allocates nothing but the conv activations and never blocks the host
        timer.start("post")
        st = lr_status(dl, dr, D, out=ws["status"] if keep is None else None)
        di = interpolate(dl, st, out=m[2], directions=ex["interpolation_directions"],
                         occlusion_from_left=ex["occlusion_from_left"])
        ds = sub(di)
        dm = median(ds, 5, 5, out=m[4])
        if out is not None and (tuple(out.shape) != (H, W) or out.dtype != torch.float32 or not out.is_contiguous()
                                or out.device != L.device):
            raise ValueError("match: `out` must be a contiguous float32 [H,W] tensor on the images' device")
        db = bilateral(L, dm, 5, 5, 0, hp["blur_sigma"], hp["blur_threshold"], out=out if out is not None else m[5])
        timer.stop()
        if keep is not None:
            keep.update(wta=(dl, dr), status=st, interp=di, subpixel=ds, median=dm, bilateral=db)
            return db
        # the workspace map is overwritten by the next pair: hand out a copy unless the caller passed its own tensor
        # or (match_graph) wants the static buffer
        return db if (_static_out or out is not None) else db.clone()

    def match_graph(self, left_image, right_image, ndisp):
        """match() replayed as ONE hipGraph launch: the ~75 kernel launches of a pair are captured once per image
        shape (after two eager warm-up pairs, so MIOpen has chosen its kernels and the allocator its blocks) and
        replayed on static input/output buffers.  Returns the static output map [H,W] (overwritten by the next
        call).  The images are copied into the static inputs in stream order; nothing synchronises."""
        L = left_image.reshape(left_image.shape[0], left_image.shape[1])
        R = right_image.reshape(right_image.shape[0], right_image.shape[1])
        key = (L.shape[0], L.shape[1], int(ndisp))
        g = self._graphs.get(key)
        if g is None:
            self.workspace(*key)                 # may reset self._graphs: one shape resident at a time
            # fresh side / right-chain streams for this capture: a stream never takes part in the captures of two graphs
            # (see _right_stream)
            self._side = None
            self._right = None
            sl, sr = torch.empty_like(L, dtype=torch.float32), torch.empty_like(R, dtype=torch.float32)
            sl.copy_(L)
            sr.copy_(R)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._match(sl, sr, ndisp)
            torch.cuda.current_stream().wait_stream(side)
            # every stream the pair touches (the side stream of the builder, the right volume's chain) is idle before the
            # capture begins: streams that enter a capture with eager work still queued have crashed the runtime once
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._match(sl, sr, ndisp, _static_out=True)
            g = (graph, sl, sr, out)
            self._graphs[key] = g
        graph, sl, sr, out = g
        sl.copy_(L)
        sr.copy_(R)
        graph.replay()
        self._saturated_pair(sl, sr, ndisp, out)      # on_saturation (one host synchronisation unless "ignore")
        return out

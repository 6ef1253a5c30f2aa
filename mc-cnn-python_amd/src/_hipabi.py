"""ctypes binding of libmccnn_hip.so (the C ABI declared in include/mccnn.h).

The product path has NO CPU fallback: if the shared library is missing, or a call fails, this module raises.
PyTorch-ROCm is used only as the owner of device memory and of the HIP stream the kernels are enqueued on.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCCNN_HIP_LIB selects another build of the same library (kernel A/B measurements); there is still no CPU path.
LIB_PATH = os.environ.get("MCCNN_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmccnn_hip.so")

MCCNN_ABI_VERSION = 7      # include/mccnn.h; load() refuses a library built from another header
MCCNN_CV_EXACT = 0
MCCNN_CV_MFMA = 1
MCCNN_CBCA_SEPARABLE = 0
MCCNN_CBCA_REFERENCE_ORDER = 1
MCCNN_SIDE_LEFT = 0
MCCNN_SIDE_RIGHT = 1
MCCNN_E_INVALID = -1      # include/mccnn.h: bad argument
MCCNN_E_UNSUPPORTED = -2  # shape / parameter the kernels are not built for
MCCNN_E_SCRATCH = -3      # scratch buffer too small

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/mccnn.h (tests/test_abi.py checks both ways)
SIGNATURES = {
    "mccnn_version": (_i, []),
    "mccnn_last_error_string": (ctypes.c_char_p, []),
    "mccnn_cost_volume": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mccnn_cost_volume_hwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mccnn_support_bytes": (_sz, [_i, _i]),
    "mccnn_cross_arms": (_i, [_vp, _i, _i, _f, _i, _vp, _vp]),
    "mccnn_cross_arms_pair": (_i, [_vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "mccnn_cross_region_list": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mccnn_cbca_iter": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_both": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_hwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_hwd_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_hwd_pair_wta": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mccnn_cbca_prog_bytes": (_sz, [_i, _i, _i]),
    "mccnn_cbca_prog_build_pair": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mccnn_cbca_iter_prog_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_prog_build_skip_pair": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mccnn_cbca_prog_build_both_pair": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mccnn_cbca_iter_prog_pair_skip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_prog": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_prog_skip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_prog_refresh": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_prog_pair_refresh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mccnn_cbca_iter_prog_pair_wta": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mccnn_wta_hwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mccnn_subpixel_hwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mccnn_hwd_pitch": (_i, [_i]),
    "mccnn_dhw_to_hwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mccnn_hwd_to_dhw": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mccnn_sgm_scratch_bytes": (_sz, [_i, _i, _i]),
    "mccnn_sgm_pass": (_i, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i,
                            _f, _f, _f, _f, _f, _vp, _sz, _vp]),
    "mccnn_sgm_flags": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "mccnn_sgm_pass_flagged": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _sz,
                                    _vp]),
    "mccnn_sgm_first_pass": (_i, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i,
                                  _i, _f, _f, _f, _f, _f, _vp, _sz, _vp]),
    "mccnn_wta": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mccnn_lr_status": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mccnn_interpolate": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mccnn_interpolate_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mccnn_subpixel": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mccnn_subpixel_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mccnn_median": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mccnn_bilateral": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp]),
    "mccnn_bias_act": (_i, [_vp, _vp, _i, _i, ctypes.c_long, _i, _vp]),
    "mccnn_conv1_pad_bias_relu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mccnn_conv3x3_split_weights_bytes": (_sz, []),
    "mccnn_conv3x3_split_pack": (_i, [_vp, _f, _vp, _vp]),
    "mccnn_conv1_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "mccnn_conv3x3_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp]),
    "mccnn_l2norm_chw_to_hwc": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


class MccnnHipError(RuntimeError):
    pass


def load():
    """Loads libmccnn_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MccnnHipError(
            "libmccnn_hip.so not found at %s - build it with `make -C mc-cnn-python_amd` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table ever diverge
        fn.restype = res
        fn.argtypes = args
    have = lib.mccnn_version()
    if have != MCCNN_ABI_VERSION:
        raise MccnnHipError("%s implements ABI version %d, this binding expects %d: rebuild the library "
                            "(`make -C mc-cnn-python_amd`)" % (LIB_PATH, have, MCCNN_ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mccnn_last_error_string()
        raise MccnnHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream():
    """The HIP stream of the current torch device context, as the void* the ABI expects."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor."""
    if not t.is_cuda:
        raise MccnnHipError("expected a device tensor; the HIP path never runs on host memory")
    if not t.is_contiguous():
        raise MccnnHipError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def require_device():
    if not torch.cuda.is_available():
        raise MccnnHipError("no HIP device visible: the MI355X path cannot run (and there is no CPU fallback)")
    load()
    return torch.device("cuda", torch.cuda.current_device())

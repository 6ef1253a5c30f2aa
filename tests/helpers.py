import numpy as np


def bits_equal(a, b):
    """Bit-for-bit equality of two arrays (float32 compared through their uint32 patterns; NaN == NaN)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        same = a.view(np.uint32) == b.view(np.uint32)
        # +0/-0 and NaN payloads are not distinguished by any consumer of these maps
        same |= (a == b) | (np.isnan(a) & np.isnan(b))
        return bool(same.all())
    return bool(np.array_equal(a, b))


def hp_of(g):
    return dict(zip([str(k) for k in g["hp_names"]], [float(v) for v in g["hp_values"]]))


def assert_bits(a, b, what):
    assert bits_equal(a, b), "%s: not bit-identical (max abs diff %g, %d of %d differ)" % (
        what, float(np.nanmax(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))),
        int((np.asarray(a) != np.asarray(b)).sum()), np.asarray(a).size)

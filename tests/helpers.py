import contextlib

import numpy as np

_QNAN = np.uint32(0x7fc00000)


def _canonical_bits(a):
    """uint32 patterns of a float32 array with every NaN mapped to one pattern (sign and payload dropped): NumPy on the
    host and the GPU both return A quiet NaN for an invalid operation, but which payload / sign is not part of IEEE 754
    and differs between x86 (default NaN = negative quiet NaN) and gfx950 (positive)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = _QNAN
    return u


def bits_strict(a, b):
    """Bit-for-bit equality of two arrays.  float32: compared through their uint32 patterns, so +0.0 != -0.0; only NaN
    payloads are canonicalised (_canonical_bits)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        return bool(np.array_equal(_canonical_bits(a), _canonical_bits(b)))
    return bool(np.array_equal(a, b))


def bits_equal(a, b):
    """The strict comparison (kept under the name the tests have always used; until round 6 it also let +0 == -0 pass)."""
    return bits_strict(a, b)


def bits_equal_up_to_zero_sign(a, b):
    """bits_strict() that lets +0.0 == -0.0 pass: only for the places listed in DESIGN.md section 2 where a sign of zero
    is known to differ from the reference and no consumer can see it."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        same = _canonical_bits(a) == _canonical_bits(b)
        same |= (a == b)
        return bool(same.all())
    return bool(np.array_equal(a, b))


def hp_of(g):
    return dict(zip([str(k) for k in g["hp_names"]], [float(v) for v in g["hp_values"]]))


def _describe(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return "shape / dtype %s %s against %s %s" % (a.shape, a.dtype, b.shape, b.dtype)
    if a.dtype == np.float32:
        ua, ub = _canonical_bits(a), _canonical_bits(b)
        differ = ua != ub
        zero_sign = differ & (a == b)
        with np.errstate(invalid="ignore"):
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        mx = float(np.nanmax(d)) if np.isfinite(d).any() else float("nan")
        return "max abs diff %g, %d of %d patterns differ (%d of them only in the sign of a zero)" % (
            mx, int(differ.sum()), a.size, int(zero_sign.sum()))
    return "%d of %d differ" % (int((a != b).sum()), a.size)


def assert_bits_strict(a, b, what):
    assert bits_strict(a, b), "%s: not bit-identical (%s)" % (what, _describe(a, b))


def assert_bits(a, b, what):
    """Strict since round 6: uint32 patterns, signs of zeros included, NaN payloads canonicalised only."""
    assert_bits_strict(a, b, what)


@contextlib.contextmanager
def module_setting(module, name, value):
    """Sets a module-level switch (process_functional.CBCA_ORDER, ...) for a block and puts back whatever stood there,
    so that no test decides which kernel a later test's un-set call reaches."""
    previous = getattr(module, name)
    setattr(module, name, value)
    try:
        yield
    finally:
        setattr(module, name, previous)

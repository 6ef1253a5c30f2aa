"""Seeded synthetic stereo pairs (there are no Middlebury images in the reference repo or on the GPU box).

Scene recipe: an [H, W+D] 8-bit scene = piecewise-flat blobs (Gaussian-filtered noise,
sigma 6 px, quantised to 16 grey levels) + fine texture (sigma 1 px, amplitude 64) on a random half of the
picture; the left view is a W-wide crop and the right view is the same scene resampled with a piecewise-constant
disparity <= 0.8*D.  Both are standardised exactly like match.py:118-121 (float32, population std).  8-bit
quantisation matters: it gives the cross-based aggregation real (non-trivial) support regions.
"""
import numpy as np


def _gauss_kernel(sigma):
    r = max(1, int(3.0 * sigma + 0.5))
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def _blur(img, sigma):
    k = _gauss_kernel(sigma)
    r = len(k) // 2
    p = np.pad(img, ((0, 0), (r, r)), mode="reflect")
    out = np.zeros_like(img)
    for i, kv in enumerate(k):
        out += kv * p[:, i:i + img.shape[1]]
    p = np.pad(out, ((r, r), (0, 0)), mode="reflect")
    out2 = np.zeros_like(img)
    for i, kv in enumerate(k):
        out2 += kv * p[i:i + img.shape[0], :]
    return out2


def standardize(gray_u8):
    """match.py:118-121: uint8 grayscale -> float32, (x - mean) / std (population std), shape [H, W, 1]."""
    img = np.asarray(gray_u8).astype(np.float32)
    img = (img - np.mean(img, axis=(0, 1))) / np.std(img, axis=(0, 1))
    return np.expand_dims(img.astype(np.float32), axis=2)


def make_scene_u8(height, width, ndisp, seed=0, texture=True):
    """Returns (left_u8 [H,W], right_u8 [H,W], true_disparity_of_right_pixels [H,W] int32).  texture=False: the second
    scene class - the flat blobs alone, no fine texture anywhere: almost every pixel has long support arms (the
    aggregation's most expensive kind of image; a few per cent of unit-region pixels instead of almost half)."""
    rng = np.random.default_rng(seed)
    sw = width + ndisp
    blobs = _blur(rng.standard_normal((height, sw)), 6.0)
    blobs = (blobs - blobs.min()) / max(blobs.max() - blobs.min(), 1e-12)
    blobs = np.floor(blobs * 15.999) * 16.0                       # 16 flat grey levels
    tex = _blur(rng.standard_normal((height, sw)), 1.0)
    tex = tex / max(np.abs(tex).max(), 1e-12) * 64.0
    mask = _blur(rng.standard_normal((height, sw)), 10.0) > 0.0   # texture on a random half
    scene = np.clip(blobs + (np.where(mask, tex, 0.0) if texture else 0.0), 0, 255).astype(np.uint8)

    off = max(1, ndisp // 10)
    # piecewise-constant disparity over a coarse 3x3 block grid, in right-image coordinates
    dmax = max(1, int(0.8 * ndisp) - 1)
    gh, gw = 3, 3
    dgrid = rng.integers(1, dmax + 1, size=(gh, gw))
    hh = np.minimum((np.arange(height) * gh) // max(height, 1), gh - 1)
    ww = np.minimum((np.arange(width) * gw) // max(width, 1), gw - 1)
    dmap = dgrid[hh][:, ww].astype(np.int32)
    cols = off + np.arange(width)[None, :] + dmap
    cols = np.minimum(cols, sw - 1)
    left = scene[:, off:off + width]
    right = np.take_along_axis(scene, cols, axis=1)
    return np.ascontiguousarray(left), np.ascontiguousarray(right), dmap


def make_pair(height, width, ndisp, seed=0, texture=True):
    """Standardised float32 pair ([H,W,1], [H,W,1]) ready for compute_features, plus the u8 sources."""
    left_u8, right_u8, dmap = make_scene_u8(height, width, ndisp, seed, texture)
    return standardize(left_u8), standardize(right_u8), left_u8, right_u8, dmap

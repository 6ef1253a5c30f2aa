"""Seeded synthetic stereo pairs (there are no Middlebury images in the reference repo or on the GPU box).

Three scene classes (make_scene_u8's `kind`): the default recipe below, its texture-free variant, and a 1/f-spectrum
"natural" picture (natural_scene_u8).

Default scene recipe: an [H, W+D] 8-bit scene = piecewise-flat blobs (Gaussian-filtered noise,
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


def natural_scene_u8(height, width, seed=0, contrast=48.0, exponent=1.0):
    """An 8-bit picture with the second-order statistics of photographs: Gaussian noise shaped to a 1/f**exponent
    AMPLITUDE spectrum (power ~ 1/f^2 for exponent 1: the classic natural-image law), mean 128, standard deviation
    `contrast` grey levels (a typical photograph: 40-60), clipped and rounded to uint8.  After match.py's
    standardisation the arm threshold 0.02 (pf:588) is about one grey level of such a picture, so an arm only runs over
    neighbours of exactly the same grey value: almost every support region is the pixel itself."""
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal((height, width))
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    spec = np.fft.rfft2(noise) / f ** float(exponent)
    spec[0, 0] = 0.0
    img = np.fft.irfft2(spec, s=(height, width))
    img = img / max(img.std(), 1e-12) * float(contrast) + 128.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


SCENE_KINDS = ("blobs+texture", "flat", "natural")


def make_scene_u8(height, width, ndisp, seed=0, texture=True, kind=None):
    """Returns (left_u8 [H,W], right_u8 [H,W], true_disparity_of_right_pixels [H,W] int32).  Three seeded scene classes
    (`kind`): "blobs+texture" (default; SURVEY 8d's recipe: about half of the pixels are unit regions), "flat"
    (= texture=False: the flat blobs alone - almost every pixel has long support arms, the aggregation's most expensive
    kind of image) and "natural" (natural_scene_u8: 1/f amplitude spectrum, 8 bit - almost every support region is the
    pixel itself, like a photograph's)."""
    if kind is None:
        kind = "blobs+texture" if texture else "flat"
    if kind not in SCENE_KINDS:
        raise ValueError("kind must be one of %s" % (SCENE_KINDS,))
    rng = np.random.default_rng(seed)
    sw = width + ndisp
    if kind == "natural":
        scene = natural_scene_u8(height, sw, seed=seed)
        rng.standard_normal((4, 4))                               # (keeps the disparity grid independent of the scene kind)
    else:
        blobs = _blur(rng.standard_normal((height, sw)), 6.0)
        blobs = (blobs - blobs.min()) / max(blobs.max() - blobs.min(), 1e-12)
        blobs = np.floor(blobs * 15.999) * 16.0                       # 16 flat grey levels
        tex = _blur(rng.standard_normal((height, sw)), 1.0)
        tex = tex / max(np.abs(tex).max(), 1e-12) * 64.0
        mask = _blur(rng.standard_normal((height, sw)), 10.0) > 0.0   # texture on a random half
        scene = np.clip(blobs + (np.where(mask, tex, 0.0) if kind == "blobs+texture" else 0.0), 0, 255).astype(np.uint8)

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


def make_pair(height, width, ndisp, seed=0, texture=True, kind=None):
    """Standardised float32 pair ([H,W,1], [H,W,1]) ready for compute_features, plus the u8 sources."""
    left_u8, right_u8, dmap = make_scene_u8(height, width, ndisp, seed, texture, kind)
    return standardize(left_u8), standardize(right_u8), left_u8, right_u8, dmap

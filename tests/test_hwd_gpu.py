"""GPU: the pixel-major ("HWD", [H][W][Dp]) kernels of the bit-exact variant - mccnn_cbca_iter_hwd(_pair),
mccnn_wta_hwd, mccnn_subpixel_hwd - called through the C ABI, against the reference's golden vectors, the CPU oracle
on ragged shapes, and their plane-major counterparts.  Everything here is bit-exact (integer/index work and float32
sums in the reference's own order)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import assert_bits, hp_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    import _hipabi
    _hipabi.require_device()
    import stereo_device
    return stereo_device


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _agg(sd, vol_dhw, image, tau, dist, n):
    """n reference-order iterations through the pixel-major kernel; numpy [D,H,W] in and out."""
    v = dev(vol_dhw)
    D = v.shape[0]
    sup = sd.cross_arms(dev(image.reshape(image.shape[0], image.shape[1])), tau, dist)
    hv = sd.dhw_to_hwd(v)
    res, _ = sd.cbca_hwd(hv, torch.full_like(hv, float("nan")), sup, D, n, dist)
    return sd.hwd_to_dhw(res, D).cpu().numpy()


def test_golden_cbca_pixel_major_bit_exact(sd, golden_cases):
    """pf:149-163 as the reference itself computed it: 1, it1 (=2) and it2 (=16) iterations, both views."""
    for name, g in golden_cases:
        hp = hp_of(g)
        tau, dist = hp["cbca_intensity"], int(hp["cbca_distance"])
        for side, img in (("l", g["left"]), ("r", g["right"])):
            assert_bits(_agg(sd, g["cv_" + side], img, tau, dist, 1), g["cbca1it_" + side], name + " 1 iteration " + side)
            assert_bits(_agg(sd, g["cv_" + side], img, tau, dist, int(hp["it1"])), g["cbca1_" + side], name + " it1 " + side)
            assert_bits(_agg(sd, g["sgm_" + side], img, tau, dist, int(hp["it2"])), g["cbca2_" + side], name + " it2 " + side)


@pytest.mark.parametrize("H,W,D", [(45, 100, 20), (64, 127, 33), (100, 70, 8), (7, 5, 3), (30, 13, 2), (21, 64, 130),
                                   (16, 75, 256), (12, 50, 400), (3, 200, 64), (24, 70, 192), (10, 40, 178),
                                   (10, 40, 150)])
def test_oracle_cbca_pixel_major_ragged_shapes(sd, H, W, D):
    """Widths that are not a multiple of the pixel group (or smaller than one), heights below the arm limit, one and two
    256-disparity chunks, D not a multiple of 4, the three-disparities-per-lane path (padded D a multiple of 3 in
    129 .. 192: 130, 178, 192) and its neighbour that is not (150): against the CPU checker, 2 iterations."""
    import oracle as o
    import synthetic
    rng = np.random.default_rng(H * 1000 + W)
    L, R, _, _, _ = synthetic.make_pair(H, W, min(16, W - 2), seed=W)
    vl = (rng.random((D, H, W), dtype=np.float32) * 3 - 2).astype(np.float32)
    vr = (rng.random((D, H, W), dtype=np.float32) * 3 - 2).astype(np.float32)
    ol, orr = o.cost_volume_aggregation(L, R, vl, vr, 0.02, 14, 2)
    assert_bits(_agg(sd, vl, L, 0.02, 14, 2), ol, "left")
    assert_bits(_agg(sd, vr, R, 0.02, 14, 2), orr, "right")


def test_oracle_cbca_pixel_major_long_arms_and_other_distances(sd):
    """Flat image (every arm at the limit: 27 x 27 regions, every window slot in use), a striped one, and distance
    thresholds below the default."""
    import oracle as o
    rng = np.random.default_rng(1)
    H, W, D = 50, 90, 5
    L = np.zeros((H, W, 1), np.float32)
    R = np.zeros((H, W, 1), np.float32)
    R[:, ::7] = 1.0
    vl = rng.standard_normal((D, H, W)).astype(np.float32)
    vr = rng.standard_normal((D, H, W)).astype(np.float32)
    for dist in (14, 6, 2, 1):
        ol, orr = o.cost_volume_aggregation(L, R, vl, vr, 0.02, dist, 2)
        assert_bits(_agg(sd, vl, L, 0.02, dist, 2), ol, "flat image, distance %d" % dist)
        assert_bits(_agg(sd, vr, R, 0.02, dist, 2), orr, "striped image, distance %d" % dist)


def test_cbca_pixel_major_special_values(sd):
    """inf / nan / signed zeros travel through the running sum exactly as float32 addition defines (the plane-major
    reference-order kernel is the witness: same additions in the same order)."""
    import _hipabi as hip
    rng = np.random.default_rng(3)
    H, W, D = 40, 66, 9
    img = dev(np.floor(rng.random((H, W), dtype=np.float32) * 3) * np.float32(0.01))
    sup = sd.cross_arms(img, 0.02, 14)
    v = rng.standard_normal((D, H, W)).astype(np.float32)
    v[0, 5, 7] = np.inf
    v[1, 9, 30] = -np.inf
    v[2, 20, 40] = np.nan
    v[3] = -0.0
    v[4, ::2] = 0.0
    vd = dev(v)
    ref, _ = sd.cbca(vd.clone(), torch.empty_like(vd), sup, 2, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
    hv = sd.dhw_to_hwd(vd)
    got, _ = sd.cbca_hwd(hv, torch.empty_like(hv), sup, D, 2, 14)
    a, b = sd.hwd_to_dhw(got, D).cpu().numpy(), ref.cpu().numpy()
    assert np.array_equal(a.view(np.uint32) | (np.isnan(a) * 0xFFFFFFFF).astype(np.uint32),
                          b.view(np.uint32) | (np.isnan(b) * 0xFFFFFFFF).astype(np.uint32))


@pytest.mark.parametrize("H,W,D,iters", [(70, 300, 5, 1), (40, 33, 1, 2), (64, 100, 300, 3)])
def test_cbca_pixel_major_pair_launch_equals_single_launches(sd, H, W, D, iters):
    import synthetic
    L, R, _, _, _ = synthetic.make_pair(H, W, min(16, W - 2), seed=11)
    sl, sr = sd.cross_arms_pair(dev(L[:, :, 0]), dev(R[:, :, 0]), 0.02, 14)
    g = torch.Generator(device="cuda").manual_seed(5)
    Dp = sd.hwd_pitch(D)
    a = torch.rand((H, W, Dp), device="cuda", generator=g) - 0.5
    b = torch.rand((H, W, Dp), device="cuda", generator=g) - 0.5
    (pl, _), (pr, _) = sd.cbca_hwd_pair(a.clone(), torch.empty_like(a), sl, b.clone(), torch.empty_like(b), sr, D, iters, 14)
    ql, _ = sd.cbca_hwd(a.clone(), torch.empty_like(a), sl, D, iters, 14)
    qr, _ = sd.cbca_hwd(b.clone(), torch.empty_like(b), sr, D, iters, 14)
    assert torch.equal(pl[:, :, :D], ql[:, :, :D]) and torch.equal(pr[:, :, :D], qr[:, :, :D])


@pytest.mark.parametrize("H,W,D,iters", [(70, 300, 5, 1), (40, 33, 64, 2), (33, 47, 192, 1), (21, 64, 130, 2),
                                         (64, 100, 256, 3), (9, 12, 3, 1), (16, 40, 250, 1)])
def test_last_iteration_with_fused_wta_equals_iteration_then_wta(sd, H, W, D, iters):
    """mccnn_cbca_iter_hwd_pair_wta: same volumes and the same disparities as the unfused launches - on ties across
    lanes, NaN / inf costs and all-NaN pixels too; with store_right = 0 the right output volume is not touched."""
    import synthetic
    L, R, _, _, _ = synthetic.make_pair(H, W, min(16, W - 2), seed=17)
    sl, sr = sd.cross_arms_pair(dev(L[:, :, 0]), dev(R[:, :, 0]), 0.02, 14)
    g = torch.Generator(device="cuda").manual_seed(9)
    Dp = sd.hwd_pitch(D)
    a = torch.round((torch.rand((H, W, Dp), device="cuda", generator=g) - 0.5) * 8) / 8      # coarse values: many ties
    b = torch.rand((H, W, Dp), device="cuda", generator=g) - 0.5
    a[3, 5, :] = float("nan"); a[4, 6, D // 2] = float("nan"); b[2, 1, 0] = float("-inf"); b[5, 7, :] = float("inf")
    (pl, _), (pr, _) = sd.cbca_hwd_pair(a.clone(), torch.empty_like(a), sl, b.clone(), torch.empty_like(b), sr, D, iters, 14)
    wl, wr = sd.wta_hwd(pl, D), sd.wta_hwd(pr, D)
    fl, fr = torch.full((H, W), 123.0, device="cuda"), torch.full((H, W), 123.0, device="cuda")
    (ql, _), (qr, _) = sd.cbca_hwd_pair(a.clone(), torch.empty_like(a), sl, b.clone(), torch.empty_like(b), sr, D, iters,
                                        14, wta_out=(fl, fr))
    def same(x, y):
        return torch.equal(torch.nan_to_num(x, nan=777.0), torch.nan_to_num(y, nan=777.0))
    assert same(ql[:, :, :D], pl[:, :, :D]) and same(qr[:, :, :D], pr[:, :, :D])
    assert torch.equal(fl, wl) and torch.equal(fr, wr)
    # right volume left unwritten
    fl2, fr2 = torch.empty_like(fl), torch.empty_like(fr)
    ping, pong = b.clone(), torch.full_like(b, -7.0)
    (_, _), (rr, spare) = sd.cbca_hwd_pair(a.clone(), torch.empty_like(a), sl, ping, pong, sr, D, iters, 14,
                                           wta_out=(fl2, fr2), store_right=False)
    assert torch.equal(fl2, wl) and torch.equal(fr2, wr)
    if iters == 1:
        assert bool((rr == -7.0).all()), "store_right = 0 must leave the right output volume untouched"


def test_fused_wta_refuses_more_than_one_chunk(sd):
    import _hipabi as hip
    H, W, D = 8, 16, 300
    img = torch.zeros((H, W), device="cuda")
    sup = sd.cross_arms(img, 0.02, 14)
    Dp = sd.hwd_pitch(D)
    a, b = torch.zeros((H, W, Dp), device="cuda"), torch.zeros((H, W, Dp), device="cuda")
    c, d = torch.zeros((H, W, Dp), device="cuda"), torch.zeros((H, W, Dp), device="cuda")
    dl, dr = torch.zeros((H, W), device="cuda"), torch.zeros((H, W), device="cuda")
    rc = hip.load().mccnn_cbca_iter_hwd_pair_wta(hip.ptr(a), hip.ptr(b), hip.ptr(sup), hip.ptr(c), hip.ptr(d), hip.ptr(sup),
                                                 D, H, W, 14, hip.ptr(dl), hip.ptr(dr), 1, hip.stream())
    assert rc == hip.MCCNN_E_UNSUPPORTED
    with pytest.raises(ValueError):
        sd.cbca_hwd_pair(a, b, sup, c, d, sup, D, 1, 14, wta_out=(dl, dr))


def test_golden_wta_and_subpixel_pixel_major(sd, golden_cases):
    for name, g in golden_cases:
        for side in ("l", "r"):
            v = dev(g["cbca2_" + side])
            D = v.shape[0]
            assert_bits(sd.wta_hwd(sd.dhw_to_hwd(v), D).cpu().numpy(), g["wta_" + side], name + " wta " + side)
        v = dev(g["cbca2_l"])
        D = v.shape[0]
        assert_bits(sd.subpixel_hwd(dev(g["interp"]), sd.dhw_to_hwd(v), D).cpu().numpy(), g["subpixel"], name + " subpixel")


@pytest.mark.parametrize("H,W,D", [(30, 50, 12), (17, 33, 70), (9, 40, 256), (8, 21, 401), (5, 7, 2)])
def test_wta_subpixel_pixel_major_ties_nan_inf(sd, H, W, D):
    """Ties (lowest index wins, also across lanes and across 256-disparity groups), all-NaN / all-inf pixels (-1),
    NaN and inf among finite costs, zero denominators in the parabola: identical to the plane-major kernels, which are
    pinned against the reference's golden vectors and the oracle."""
    rng = np.random.default_rng(D)
    v = np.round(rng.standard_normal((D, H, W)).astype(np.float32) * 2) / 2      # many exact ties
    v[:, 0, 0] = np.nan
    v[:, 0, 1] = np.inf
    v[:, 1, 2] = 1.0                          # all equal: index 0
    v[D // 2, 2, 3] = np.nan
    v[:, 2, 4] = -np.inf                      # -inf everywhere: index 0
    v[D - 1, 3, 5] = -100.0                   # minimum in the last disparity
    if D > 256:
        v[:, 4, 6] = 0.0
        v[[3, 300], 4, 6] = -7.0              # tie across the two disparity groups: 3 wins
    vd = dev(v)
    hv = sd.dhw_to_hwd(vd)
    w1, w2 = sd.wta(vd), sd.wta_hwd(hv, D)
    assert torch.equal(w1, w2)
    assert float(w2[0, 0]) == -1.0 and float(w2[0, 1]) == -1.0 and float(w2[1, 2]) == 0.0 and float(w2[3, 5]) == D - 1
    d = (w1.clamp(min=0) + 0.5 * (torch.rand_like(w1) > 0.5)).contiguous()
    for promo in (False, True):
        s1 = sd.subpixel(d, vd, numpy1_promotion=promo)
        s2 = sd.subpixel_hwd(d, hv, D, numpy1_promotion=promo)
        assert torch.equal(torch.nan_to_num(s1, nan=12345.0), torch.nan_to_num(s2, nan=12345.0))


def test_pixel_major_abi_error_behaviour(sd):
    import _hipabi as hip
    lib = hip.load()
    H, W, D = 8, 16, 4
    sup = sd.cross_arms(torch.zeros((H, W), device="cuda"), 0.02, 14)
    a = torch.zeros((H, W, 4), device="cuda")
    b = torch.zeros_like(a)
    st = hip.stream()
    assert lib.mccnn_cbca_iter_hwd(hip.ptr(a), hip.ptr(a), hip.ptr(sup), D, H, W, 14, st) == hip.MCCNN_E_INVALID
    assert lib.mccnn_cbca_iter_hwd(hip.ptr(a), hip.ptr(b), hip.ptr(sup), D, H, W, 15, st) == hip.MCCNN_E_UNSUPPORTED
    assert b"L=15" in lib.mccnn_last_error_string()
    assert lib.mccnn_cbca_iter_hwd(None, hip.ptr(b), hip.ptr(sup), D, H, W, 14, st) == hip.MCCNN_E_INVALID
    assert lib.mccnn_cbca_iter_hwd(hip.ptr(a), hip.ptr(b), hip.ptr(sup), D, H + 1, W, 14, st) == hip.MCCNN_E_INVALID  # other image
    assert lib.mccnn_cbca_iter_hwd_pair(hip.ptr(a), hip.ptr(b), hip.ptr(sup), hip.ptr(a), hip.ptr(b), hip.ptr(sup), D, H,
                                        W, 14, st) == hip.MCCNN_E_INVALID                                            # aliasing
    assert lib.mccnn_wta_hwd(hip.ptr(a), 0, H, W, hip.ptr(b), st) == hip.MCCNN_E_INVALID
    assert lib.mccnn_subpixel_hwd(None, hip.ptr(a), D, H, W, 0, hip.ptr(b), st) == hip.MCCNN_E_INVALID
    assert lib.mccnn_cbca_iter_hwd(hip.ptr(a), hip.ptr(b), hip.ptr(sup), D, H, W, 14, st) == 0


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_size_pixel_major_equals_plane_major(sd, cfg):
    """One reference-order iteration at BASELINE's full sizes (750x500x256: one 256-disparity chunk with every lane in
    use; 1242x375x192: a partial chunk, 207 pixel groups per row) on both kernels: bit-identical volumes, hence the
    same WTA map."""
    import _hipabi as hip
    import synthetic
    from bench import CONFIGS
    H, W, D = CONFIGS[cfg]
    L = synthetic.make_pair(H, W, D, seed=100)[0]
    sup = sd.cross_arms(dev(L[:, :, 0]), 0.02, 14)
    g = torch.Generator(device="cuda").manual_seed(0)
    v = -torch.rand((D, H, W), device="cuda", generator=g) * 50
    ref, _ = sd.cbca(v.clone(), torch.empty_like(v), sup, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
    hv = sd.dhw_to_hwd(v)
    got, _ = sd.cbca_hwd(hv, torch.empty_like(hv), sup, D, 1, 14)
    assert torch.equal(sd.hwd_to_dhw(got, D), ref)
    assert torch.equal(sd.wta_hwd(got, D), sd.wta(ref))


def test_golden_cost_volume_pixel_major(sd, golden_cases):
    """pf:78-113 written straight into pixel-major volumes: the reference's own cost volumes, bit for bit."""
    for name, g in golden_cases:
        D = g["cv_l"].shape[0]
        lh, rh = sd.cost_volume_hwd(dev(g["fl"]), dev(g["fr"]), D)
        assert_bits(sd.hwd_to_dhw(lh, D).cpu().numpy(), g["cv_l"], name + " cv_l")
        assert_bits(sd.hwd_to_dhw(rh, D).cpu().numpy(), g["cv_r"], name + " cv_r")


@pytest.mark.parametrize("H,W,D", [(21, 70, 40), (40, 200, 64), (12, 300, 256), (5, 66, 64), (7, 130, 128), (3, 450, 400),
                                   (9, 35, 33), (4, 64, 1), (6, 131, 2), (2, 700, 130)])
def test_cost_volume_pixel_major_equals_plane_major(sd, H, W, D):
    """Tile edges (W and D either side of multiples of 64), D = W - 2 (the longest border the reference defines),
    D = 1 (no border), two 256-disparity groups in the border sweep, features that are not unit vectors: the same bits
    as the plane-major NumPy-order kernel, which is pinned against the golden vectors and the oracle."""
    import _hipabi as hip
    g = torch.Generator(device="cuda").manual_seed(H * W + D)
    fl = torch.randn((H, W, 64), device="cuda", generator=g)
    fr = torch.randn((H, W, 64), device="cuda", generator=g)
    l0, r0 = sd.cost_volume(fl, fr, D, hip.MCCNN_CV_EXACT)
    lh, rh = sd.cost_volume_hwd(fl, fr, D)
    assert torch.equal(sd.hwd_to_dhw(lh, D), l0), "left volume"
    assert torch.equal(sd.hwd_to_dhw(rh, D), r0), "right volume"
    # the matrix-core cost volume (fast variants) written pixel-major: the same bits as its plane-major form
    fl, fr = torch.nn.functional.normalize(fl, dim=-1), torch.nn.functional.normalize(fr, dim=-1)
    l1, r1 = sd.cost_volume(fl, fr, D, hip.MCCNN_CV_MFMA)
    lm, rm = sd.cost_volume_hwd(fl, fr, D, mode=hip.MCCNN_CV_MFMA)
    assert torch.equal(sd.hwd_to_dhw(lm, D), l1), "left volume, matrix cores"
    assert torch.equal(sd.hwd_to_dhw(rm, D), r1), "right volume, matrix cores"


def test_cost_volume_pixel_major_abi_errors(sd):
    import _hipabi as hip
    lib = hip.load()
    f = torch.zeros((4, 40, 64), device="cuda")
    o = torch.zeros((4, 40, 8), device="cuda")
    st = hip.stream()
    assert lib.mccnn_cost_volume_hwd(hip.ptr(f), hip.ptr(f), 4, 40, 64, 8, hip.ptr(o), hip.ptr(o), 7, st) \
        == hip.MCCNN_E_INVALID
    assert lib.mccnn_cost_volume_hwd(hip.ptr(f), hip.ptr(f), 4, 40, 64, 39, hip.ptr(o), hip.ptr(o), hip.MCCNN_CV_EXACT, st) \
        == hip.MCCNN_E_UNSUPPORTED
    assert lib.mccnn_cost_volume_hwd(None, hip.ptr(f), 4, 40, 64, 8, hip.ptr(o), hip.ptr(o), hip.MCCNN_CV_EXACT, st) \
        == hip.MCCNN_E_INVALID


@pytest.mark.parametrize("H,W,D", [(12, 90, 64), (9, 300, 256), (6, 77, 40)])
def test_cost_volume_border_fill_special_values_against_the_oracle(sd, H, W, D):
    """pf:94-95 / 105-106 (the 3-tap border recurrences) on scores that are subnormal, +-0, +-inf, NaN, next to FLT_MIN and
    next to FLT_MAX: the fill's three-operation replacement of `s / 3.f` (cost_volume.hip, `third`; exhaustively
    checked on x86 by tools/probe/div3_exhaustive.c) runs here on the device - float32 denormals on, no contraction
    (-ffp-contract=off in the Makefile) - and must give the checker's bits, signs of zeros included.  The features
    have one non-zero channel, so a score is minus ONE exact product and every special value is placed at will."""
    import oracle as o
    from helpers import assert_bits_strict
    rng = np.random.default_rng(H * W + D)
    mags = np.array([0.0, -0.0, 1e-22, 1e-20, 3e-20, 1e-19, 1.0, -1.0, 0.75, 1e18, 1.8e19, -1.8e19, 2.5e19,
                     np.inf, -np.inf, np.nan], np.float32)

    def feats(special_rate):
        f = np.zeros((H, W, 64), np.float32)
        m = rng.standard_normal((H, W)).astype(np.float32)
        hit = rng.random((H, W)) < special_rate
        m[hit] = mags[rng.integers(0, mags.size, int(hit.sum()))]
        f[:, :, 0] = m
        return f
    # rows 0..: mostly ordinary values with specials sprinkled in; last rows: specials everywhere
    fl, fr = feats(0.15), feats(0.15)
    fl[-2:], fr[-2:] = feats(1.0)[-2:], feats(1.0)[-2:]
    with np.errstate(all="ignore"):
        ol, orr = o.compute_cost_volume(fl, fr, D)
    assert np.isnan(ol).any() and np.isinf(ol).any()
    tiny = np.abs(ol[np.isfinite(ol) & (ol != 0)])
    assert (tiny < 1.2e-38).any() and (tiny > 1e38).any()                  # subnormal and near-FLT_MAX entries exist
    gl, gr = sd.cost_volume_hwd(dev(fl), dev(fr), D)
    assert_bits_strict(sd.hwd_to_dhw(gl, D).cpu().numpy(), ol, "cost volume + border fill, special values (left)")
    assert_bits_strict(sd.hwd_to_dhw(gr, D).cpu().numpy(), orr, "cost volume + border fill, special values (right)")
    # the plane-major form of the same stage (process_functional's boundary)
    pl, pr = sd.cost_volume(dev(fl), dev(fr), D)
    assert_bits_strict(pl.cpu().numpy(), ol, "plane-major cost volume, special values (left)")
    assert_bits_strict(pr.cpu().numpy(), orr, "plane-major cost volume, special values (right)")


@pytest.mark.parametrize("H,W,D", [(37, 61, 256), (20, 90, 192), (16, 70, 40), (12, 300, 400)])
def test_sgm_flag_planes_built_once_equal_the_pass_that_builds_its_own(sd, H, W, D):
    """mccnn_sgm_flags + mccnn_sgm_pass_flagged (the flag planes of a direction built once per pair and read by the passes
    of both volumes, also when these run as separate one-volume launches) against mccnn_sgm_pass and against the CPU
    checker's SGM_average (pf:187-235): every bit."""
    import oracle as o
    import synthetic
    rng = np.random.default_rng(H + W + D)
    L, R, _, _, _ = synthetic.make_pair(H, W, min(D, W - 2), seed=11)
    l, r = dev(L[:, :, 0]), dev(R[:, :, 0])
    vl = (-rng.random((D, H, W), dtype=np.float32)).astype(np.float32)
    vr = (-rng.random((D, H, W), dtype=np.float32)).astype(np.float32)
    hp = (2.3, 55.9, 4.0, 8.0, 0.08, 1.5)
    want = o.SGM_average(vl.copy(), vr.copy(), L, R, *[x if i != 2 and i != 3 else int(x) for i, x in enumerate(hp)])
    scratch = sd.sgm_scratch(H, W, D, l.device)
    flags = sd.sgm_flag_planes(l, r, D, hp[4])
    # two-volume launches on planes built once
    a, b = sd.dhw_to_hwd(dev(vl)), sd.dhw_to_hwd(dev(vr))
    sd.sgm_average_hwd(l, r, [a, b], [0, 1], D, *hp, None, flags=flags)
    # one-volume launches, the two volumes on two streams, the same planes
    c, d = sd.dhw_to_hwd(dev(vl)), sd.dhw_to_hwd(dev(vr))
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    sd.sgm_average_hwd(l, r, [c], [0], D, *hp, None, flags=flags)
    with torch.cuda.stream(s2):
        sd.sgm_average_hwd(l, r, [d], [1], D, *hp, None, flags=flags)
    torch.cuda.current_stream().wait_stream(s2)
    # the pass that builds its own planes
    e, f = sd.dhw_to_hwd(dev(vl)), sd.dhw_to_hwd(dev(vr))
    sd.sgm_average_hwd(l, r, [e, f], [0, 1], D, *hp, scratch)
    for name, (x, y) in (("planes built once", (a, b)), ("one-volume launches on two streams", (c, d)), ("mccnn_sgm_pass", (e, f))):
        assert_bits(sd.hwd_to_dhw(x, D).cpu().numpy(), want[0], "SGM_average, %s (left)" % name)
        assert_bits(sd.hwd_to_dhw(y, D).cpu().numpy(), want[1], "SGM_average, %s (right)" % name)

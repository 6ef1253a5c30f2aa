"""GPU: the opt-in extras - rules of the MC-CNN paper that the reference names and leaves out, and the
scalar promotion of the NumPy it was written for.  None of this is reference behaviour, so the checkers are plain
Python restatements of the definitions in include/mccnn.h, kept here with the tests; the defaults (extras off) are
covered by the parity tests."""
import math

import numpy as np
import pytest
import torch

from helpers import assert_bits

pytestmark = pytest.mark.gpu

RAYS = [(1, 0), (1, 0.5), (1, 1), (0.5, 1), (0, 1), (-0.5, 1), (-1, 1), (-1, 0.5), (-1, 0), (-1, -0.5), (-1, -1),
        (-0.5, -1), (0, -1), (0.5, -1), (1, -1), (1, -0.5)]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def interpolate_check(dl, st, directions, occ_left):
    H, W = dl.shape
    out = dl.copy()
    for h in range(H):
        for w in range(W):
            if st[h, w] == 1:
                nb = []
                if directions == 16:
                    for dx, dy in RAYS:
                        xx, yy = float(w), float(h)
                        while True:
                            xx += dx
                            yy += dy
                            xi, yi = int(math.floor(xx + 0.5)), int(math.floor(yy + 0.5))
                            if xi < 0 or xi >= W or yi < 0 or yi >= H:
                                break
                            if st[yi, xi] == 0:
                                nb.append(dl[yi, xi])
                                break
                else:
                    for rng_ in (range(w + 1, W), range(w - 1, -1, -1)):
                        for x in rng_:
                            if st[h, x] == 0:
                                nb.append(dl[h, x])
                                break
                    for rng_ in (range(h + 1, H), range(h - 1, -1, -1)):
                        for y in rng_:
                            if st[y, w] == 0:
                                nb.append(dl[y, w])
                                break
                if nb:
                    out[h, w] = np.median(np.array(nb, dtype=np.float32))
            elif st[h, w] == 2:
                for x in (range(w - 1, -1, -1) if occ_left else range(w + 1, W)):
                    if st[h, x] == 0:
                        out[h, w] = dl[h, x]
                        break
    return out


def test_paper_interpolation_rules():
    import oracle as o
    import stereo_device as sd
    rng = np.random.default_rng(4)
    H, W, D = 37, 53, 20
    dl = rng.integers(0, D, size=(H, W)).astype(np.float32)
    st = rng.choice([0, 1, 2], size=(H, W), p=[0.3, 0.45, 0.25]).astype(np.int32)
    st[:5, :9] = 1                                    # a block with no match nearby in some directions
    for directions in (4, 16):
        for occ_left in (False, True):
            got = sd.interpolate(dev(dl), dev(st), directions=directions, occlusion_from_left=occ_left).cpu().numpy()
            assert_bits(got, interpolate_check(dl, st, directions, occ_left), "interp %d %s" % (directions, occ_left))
    # (4, right) is the reference's rule: equal to the oracle on a consistent status map
    dr = rng.integers(0, D, size=(H, W)).astype(np.float32)
    st2 = o.lr_status(dl, dr, D)
    got = sd.interpolate(dev(dl), dev(st2)).cpu().numpy()
    assert_bits(got, o.interpolation(dl, dr, D), "default rule")
    assert not np.array_equal(sd.interpolate(dev(dl), dev(st2), directions=16).cpu().numpy(), got)


def test_numpy1_promotion_subpixel():
    import stereo_device as sd
    rng = np.random.default_rng(1)
    H, W, D = 30, 44, 12
    vol = rng.random((D, H, W), dtype=np.float32)
    d = rng.integers(0, D, size=(H, W)).astype(np.float32)
    want = d.copy()
    for h in range(H):
        for w in range(W):
            di = d[h, w]
            if int(di - 1) < 0 or int(di + 1) >= D:
                continue
            cm, cp, c = vol[int(di - 1), h, w], vol[int(di + 1), h, w], vol[int(di), h, w]
            num = np.float32(cp - cm)                               # float32 - float32 stays float32 under NumPy 1
            den = 2.0 * (np.float64(cp) - 2.0 * np.float64(c) + np.float64(cm))
            want[h, w] = np.float32(np.float64(di) - np.float64(num) / den)
    got = sd.subpixel(dev(d), dev(vol), numpy1_promotion=True).cpu().numpy()
    assert_bits(got, want, "NumPy-1 promotion")
    plain = sd.subpixel(dev(d), dev(vol)).cpu().numpy()
    assert np.abs(plain - got).max() <= 1e-3 and not np.array_equal(plain, got)


def cbca_both_check(vol, arms_self, arms_other, side):
    """arms: uint8 [H,W,4] = up, down, left, right."""
    D, H, W = vol.shape
    out = np.empty_like(vol)
    for d in range(D):
        sh = -d if side == 0 else d
        for y in range(H):
            for x in range(W):
                def arms(qy):
                    a = arms_self[qy, x].astype(int)
                    xo = x + sh
                    if 0 <= xo < W:
                        a = np.minimum(a, arms_other[qy, xo].astype(int))
                    return a
                u, dn, _, _ = arms(y)
                s, n = np.float32(0), 0
                for qy in [y] + [y - k for k in range(1, u + 1)] + [y + k for k in range(1, dn + 1)]:
                    _, _, l, r = arms(qy)
                    for xx in [x] + [x - k for k in range(1, l + 1)] + [x + k for k in range(1, r + 1)]:
                        s = np.float32(s + vol[d, qy, xx])
                    n += l + r + 1
                out[d, y, x] = np.float32(s / np.float32(n))
    return out


def test_both_view_support_regions():
    import _hipabi as hip
    import stereo_device as sd
    import synthetic
    H, W, D = 26, 40, 5
    L, R, _, _, _ = synthetic.make_pair(H, W, 8, seed=6)
    il, ir = dev(L[:, :, 0]), dev(R[:, :, 0])
    sl, sr = sd.cross_arms(il, 0.02, 14), sd.cross_arms(ir, 0.02, 14)
    al, ar = sd.support_arms(sl).cpu().numpy(), sd.support_arms(sr).cpu().numpy()
    rng = np.random.default_rng(2)
    vol = (-rng.random((D, H, W), dtype=np.float32)).astype(np.float32)
    for side, own, other, a_own, a_other in ((hip.MCCNN_SIDE_LEFT, sl, sr, al, ar), (hip.MCCNN_SIDE_RIGHT, sr, sl, ar, al)):
        got, _ = sd.cbca_both_views(dev(vol), torch.empty((D, H, W), device="cuda"), own, other, 1, 14, side)
        assert_bits(got.cpu().numpy(), cbca_both_check(vol, a_own, a_other, side), "both views, side %d" % side)
    # plane 0 of the left volume with identical views: the intersection changes nothing -> the reference-order result
    same, _ = sd.cbca_both_views(dev(vol[:1]), torch.empty((1, H, W), device="cuda"), sl, sl, 1, 14, hip.MCCNN_SIDE_LEFT)
    ref, _ = sd.cbca(dev(vol[:1]), torch.empty((1, H, W), device="cuda"), sl, 1, 14, hip.MCCNN_CBCA_REFERENCE_ORDER)
    assert torch.equal(same, ref)


def test_extras_through_the_matcher_and_cli_flags(net_layers):
    import _hipabi as hip
    import match
    import stereo_device as sd
    import synthetic
    from model import NET
    H, W, D = 48, 64, 12
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=8)
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(net_layers)
    base = sd.StereoMatcher(net, cv_mode=hip.MCCNN_CV_EXACT, cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER)
    ref = base.match(dev(L), dev(R), D).clone()
    for extras in (dict(both_view_support=True), dict(interpolation_directions=16, occlusion_from_left=True),
                   dict(numpy1_promotion=True)):
        m = sd.StereoMatcher(net, cv_mode=hip.MCCNN_CV_EXACT, cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER, extras=extras)
        out = m.match(dev(L), dev(R), D)
        assert out.shape == ref.shape and bool(torch.isfinite(out).all())
        assert not torch.equal(out, ref), extras                 # they are departures, and they are off by default
        assert float((out - ref).abs().median()) <= 1.0
    with pytest.raises(ValueError):
        sd.StereoMatcher(net, extras=dict(no_such_option=True))
    a = match.parser.parse_args(["--list_file", "l", "--data_dir", "d", "--save_dir", "s", "-t", "x", "-s", "0", "-e", "1",
                                 "--paper_support_regions", "--paper_interpolation", "--numpy1_promotion"])
    assert a.paper_support_regions and a.paper_interpolation and a.numpy1_promotion

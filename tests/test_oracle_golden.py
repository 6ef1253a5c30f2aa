"""CPU: the C oracle (oracle/mccnn_oracle.c) against the golden vectors that were produced by RUNNING the reference
(tests/golden/gen_golden.py).  Every stage a2..a11 must be bit-identical; this is what pins the oracle."""
import numpy as np
import pytest

import oracle as o
from helpers import assert_bits, hp_of

DIRS = dict(right=(0, 1), left=(0, -1), up=(-1, 0), bottom=(1, 0))


def test_cost_volume(golden_cases):
    for name, g in golden_cases:
        l, r = o.compute_cost_volume(g["fl"], g["fr"], g["cv_l"].shape[0])
        assert_bits(l, g["cv_l"], name + " cv_l")
        assert_bits(r, g["cv_r"], name + " cv_r")


def test_cross_region(golden_cases):
    for name, g in golden_cases:
        hp = hp_of(g)
        for side in "lr":
            arms, cnt = o.cross_arms(g["left" if side == "l" else "right"], hp["cbca_intensity"], int(hp["cbca_distance"]))
            assert np.array_equal(arms, g["arms_" + side]), name
            assert np.array_equal(cnt, g["region_num_" + side]), name
        if "region_l_crop" in g:
            reg, num = o.compute_cross_region(g["left"], hp["cbca_intensity"], int(hp["cbca_distance"]))
            assert np.array_equal(reg[:6, :8], g["region_l_crop"])
            assert np.array_equal(num, g["region_num_l"])


def test_cbca(golden_cases):
    for name, g in golden_cases:
        hp = hp_of(g)
        L, R = g["left"], g["right"]
        tau, dist = hp["cbca_intensity"], int(hp["cbca_distance"])
        for its, key in ((1, "cbca1it"), (int(hp["it1"]), "cbca1")):
            l, r = o.cost_volume_aggregation(L, R, g["cv_l"], g["cv_r"], tau, dist, its)
            assert_bits(l, g[key + "_l"], name + key)
            assert_bits(r, g[key + "_r"], name + key)
        l, r = o.cost_volume_aggregation(L, R, g["sgm_l"], g["sgm_r"], tau, dist, int(hp["it2"]))
        assert_bits(l, g["cbca2_l"], name + " cbca2_l")
        assert_bits(r, g["cbca2_r"], name + " cbca2_r")


def test_sgm_single_directions(golden_cases):
    for name, g in golden_cases:
        hp = hp_of(g)
        for dname, r in DIRS.items():
            p1 = hp["sgm_P1"] if r[0] == 0 else hp["sgm_P1"] / hp["sgm_V"]
            for side, ch in (("l", "L"), ("r", "R")):
                v = g["cbca1_" + side].copy()
                out = o.semi_global_matching(g["left"], g["right"], v, r, p1, hp["sgm_P2"], hp["sgm_Q1"], hp["sgm_Q2"],
                                             hp["sgm_D"], ch)
                assert out is v
                assert_bits(v, g["sgm_%s_%s" % (dname, side)], "%s sgm %s %s" % (name, dname, ch))


def test_sgm_average_is_sequential_composition(golden_cases):
    for name, g in golden_cases:
        hp = hp_of(g)
        a, b = g["cbca1_l"].copy(), g["cbca1_r"].copy()
        l, r = o.SGM_average(a, b, g["left"], g["right"], hp["sgm_P1"], hp["sgm_P2"], hp["sgm_Q1"], hp["sgm_Q2"],
                             hp["sgm_D"], hp["sgm_V"])
        assert_bits(l, g["sgm_l"], name + " sgm_l")
        assert_bits(r, g["sgm_r"], name + " sgm_r")
        assert_bits(a, g["sgm_l"], name + " (input mutated like the reference)")


def test_wta_to_bilateral(golden_cases):
    for name, g in golden_cases:
        hp = hp_of(g)
        D = g["cv_l"].shape[0]
        dl, dr = o.disparity_prediction(g["cbca2_l"], g["cbca2_r"])
        assert_bits(dl, g["wta_l"], name + " wta_l")
        assert_bits(dr, g["wta_r"], name + " wta_r")
        assert_bits(o.interpolation(g["wta_l"], g["wta_r"], D), g["interp"], name + " interp")
        assert_bits(o.subpixel_enhance(g["interp"], g["cbca2_l"]), g["subpixel"], name + " subpixel")
        assert_bits(o.median_filter(g["subpixel"], 5, 5), g["median"], name + " median")
        assert_bits(o.bilateral_filter(g["left"], g["median"], 5, 5, 0, hp["blur_sigma"], hp["blur_threshold"]),
                    g["bilateral"], name + " bilateral")


def test_end_to_end_chain(golden_cases, net_layers):
    """oracle.match_pair (the whole timed region) lands on the reference's final map."""
    name, g = golden_cases[0]
    D = g["cv_l"].shape[0]
    out, st = o.match_pair(g["left"], g["right"], D, net_layers, return_all=True)
    assert_bits(st["cost_volume"][0], g["cv_l"], "chain cv")   # features are deterministic -> same volume
    assert_bits(out, g["bilateral"], "chain final")


def test_features_are_unit_vectors(golden_cases, net_layers):
    name, g = golden_cases[0]
    f = o.net_features(g["left"], net_layers)
    assert f.shape == g["fl"].shape
    assert_bits(f, g["fl"], "features (oracle is deterministic)")
    assert np.allclose(np.linalg.norm(f.astype(np.float64), axis=-1), 1.0, atol=1e-6)

"""GPU: the split-operand (f16 hi/lo, three MFMA products) feature stack of csrc/conv_mfma.hip - SURVEY 8 a1, the default.

Checker: the same network evaluated in float64 by torch on the CPU (conv2d VALID on the once-padded image, ReLU,
tf.nn.l2_normalize - model.py:51-64 of the reference as model.NET restates it).  The float32 library path (MIOpen) is
measured against the same float64 values; the split path has to be as close, and both inside the 1e-5 bound the golden
feature test uses.  The golden features themselves (float64-accumulating restatement of the reference) are checked too.
Measured numbers go to gpurun_out/parity_features_split.json."""
import json
import os

import numpy as np
import pytest
import torch

import tolerances as tol
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def features_float64(net, img):
    """[H,W] float32 image -> [H,W,64] float64 unit features on the CPU."""
    pad = (net.input_patch_size - 1) // 2
    x = F.pad(img.double().cpu()[None, None], (pad, pad, pad, pad))
    nl = net.num_conv_layers
    for k in range(nl):
        x = F.conv2d(x, net.weights[k].detach().double().cpu(), net.biases[k].detach().double().cpu())
        if k < nl - 1:
            x = F.relu(x)
    x = x[0].permute(1, 2, 0)
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))


def smooth_pair(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    a = F.avg_pool2d(torch.randn((1, 1, H + 8, W + 8), generator=g), 5, 1, 2)[0, 0, 4:-4, 4:-4]
    b = torch.roll(a, 3, 1) + 0.05 * torch.randn((H, W), generator=g)
    out = []
    for t in (a, b):
        out.append(((t - t.mean()) / t.std()).float().contiguous())
    return out


@pytest.fixture(scope="module")
def pf():
    import _hipabi
    _hipabi.require_device()
    import process_functional
    return process_functional


@pytest.fixture(scope="module")
def sd():
    import stereo_device
    return stereo_device


@pytest.fixture(scope="module")
def nets(net_layers):
    from model import NET
    trained = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(net_layers)
    random_init = NET(None, input_patch_size=11, batch_size=1, device="cuda", seed=3)
    return {"converted reference checkpoint": trained, "random init": random_init}


@pytest.mark.parametrize("H,W", [(2, 3), (3, 200), (16, 32), (37, 61), (120, 200), (75, 333)])
def test_split_features_as_close_to_float64_as_the_library_path(nets, H, W):
    record = {}
    for name, net in nets.items():
        L, R = smooth_pair(H, W, 7)
        ref = [features_float64(net, L), features_float64(net, R)]
        lib = net.features_pair_hwc(L.cuda(), R.cuda())
        spl = net.features_pair_hwc_split(L.cuda(), R.cuda())
        assert spl[0].shape == (H, W, 64) and spl[0].dtype == torch.float32 and spl[0].is_contiguous()
        e_lib = max(float((lib[i].double().cpu() - ref[i]).abs().max()) for i in range(2))
        e_spl = max(float((spl[i].double().cpu() - ref[i]).abs().max()) for i in range(2))
        record["%s %dx%d" % (name, W, H)] = {"library_fp32_max_abs_err": e_lib, "split_f16_max_abs_err": e_spl}
        # unit vectors: absolute error = error relative to the vector norm.  The SAME bound for both paths (round 4: the
        # cross terms have an accumulator of their own; measured library 1.1e-7 .. 2.8e-7, split 1.3e-7 .. 3.1e-7;
        # round 2's single accumulator: up to 5.3e-7)
        assert e_spl <= tol.FEATURES_F32_CLASS_ABS, "%s: split path %g from float64" % (name, e_spl)
        assert e_lib <= tol.FEATURES_F32_CLASS_ABS, "%s: library path %g from float64" % (name, e_lib)
        assert e_spl <= 1.5 * e_lib + 5e-8, "%s: split %g vs library %g" % (name, e_spl, e_lib)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_features_split.json")
    old = {}
    if os.path.isfile(path):
        with open(path) as f:
            old = json.load(f)
    old.update(record)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


def test_split_features_golden(pf, golden_cases, net_layers):
    """Same bound as test_golden_features (1e-5 against the float64-accumulating restatement of the reference)."""
    from model import NET
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(net_layers)
    for name, g in golden_cases:
        left = dev(g["left"]).reshape(g["left"].shape[0], g["left"].shape[1])
        right = dev(g["right"]).reshape(g["right"].shape[0], g["right"].shape[1])
        fl, fr = net.features_pair_hwc_split(left, right)
        assert np.abs(fl.cpu().numpy() - g["fl"]).max() <= 1e-5, name
        assert np.abs(fr.cpu().numpy() - g["fr"]).max() <= 1e-5, name


def test_saturation_is_reported_by_the_kernels(nets, pf):
    """An activation beyond the f16 range of the stored records (|x| >= 255.9) sets the device flag the kernels carry
    (no host-side guess): ordinary standardised images never do, an image scaled by 1e4 does, the flag resets on read,
    and process_functional.compute_features then answers with the float32 library path."""
    net = nets["converted reference checkpoint"]
    L, R = smooth_pair(40, 56, 5)
    net.features_pair_hwc_split(L.cuda(), R.cuda())
    assert net.split_saturated() is False
    big = (L * 1e4).cuda()
    net.features_pair_hwc_split(big, R.cuda())
    assert net.split_saturated() is True and net.split_saturated() is False
    fl, fr = pf.compute_features(big[:, :, None].cpu().numpy(), R[:, :, None].numpy(), 11, 11, net)
    lib = net.features_pair_hwc(big, R.cuda())
    assert np.array_equal(fl, lib[0].cpu().numpy()) or np.abs(fl - lib[0].cpu().numpy()).max() <= 1e-6
    assert net.split_saturated() is False                       # compute_features consumed the flag


def test_matcher_saturation_policy_and_nan_of_either_sign(sd, nets):
    """StereoMatcher(on_saturation=...): "fallback" (default) answers a pair whose activations left the records' range
    with the library-feature result - through match() and through the replayed graph -, "raise" raises, "ignore" hands
    out the clamped pair and leaves the flag to the caller.  A NaN pixel (either sign bit) sets the flag too."""
    import synthetic
    net = nets["converted reference checkpoint"]
    H, W, D = 40, 56, 12
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=2)
    l, r = dev(L[:, :, 0]), dev(R[:, :, 0])
    big = l * 1e4
    want = sd.StereoMatcher(net, features="miopen").match(big, r, D)
    fine = sd.StereoMatcher(net, features="split_f16")
    a = fine.match(l, r, D)
    assert fine._library_twin is None and not net.split_saturated()          # ordinary pair: nothing repeated
    got = fine.match(big, r, D)
    assert fine._library_twin is not None
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    got_graph = fine.match_graph(big, r, D).clone()
    assert torch.equal(got_graph.view(torch.int32), want.view(torch.int32))
    assert torch.equal(fine.match_graph(l, r, D).view(torch.int32), a.view(torch.int32))
    with pytest.raises(RuntimeError):
        sd.StereoMatcher(net, features="split_f16", on_saturation="raise").match(big, r, D)
    loose = sd.StereoMatcher(net, features="split_f16", on_saturation="ignore")
    loose.match(big, r, D)
    assert loose._library_twin is None and loose.features_saturated() and not loose.features_saturated()
    for bits in (0x7fc00000, 0xffc00000):
        bad = l.clone()
        bad.view(torch.int32)[5, 7] = bits - (1 << 32) if bits >> 31 else bits
        net.features_pair_hwc_split(bad, r)
        assert net.split_saturated() is True, hex(bits)


def test_extreme_contrast_eight_bit_images_and_the_saturation_fallback(sd, nets):
    """How often does the fallback fire on 8-bit images?  match.py standardises an image ((x - mean) / std, match.py:
    120-121), so a pixel's value is bounded by sqrt(H W) for ANY 8-bit content (a single white pixel on black), and the
    activations follow the weights from there.  Ten kinds of extreme 8-bit pairs at 96 x 128 (|x| <= 111) and at
    cfg2's 500 x 750 pixel count for the single-pixel image (|x| = 612): which of them trip the flag is recorded in
    gpurun_out/parity_features_split.json; the ones that do must come back as the library-feature result."""
    net = nets["converted reference checkpoint"]
    rng = np.random.default_rng(0)
    H, W, D = 96, 128, 16

    def std8(img):
        x = img.astype(np.float32)
        return ((x - x.mean()) / max(float(x.std()), 1e-12)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    one = np.zeros((H, W), np.uint8); one[H // 2, W // 2] = 255
    few = np.zeros((H, W), np.uint8); few[rng.integers(0, H, 12), rng.integers(0, W, 12)] = 255
    kinds = {
        "single white pixel on black": one,
        "twelve white pixels on black": few,
        "checkerboard 0/255": (((yy + xx) & 1) * 255).astype(np.uint8),
        "vertical stripes 0/255": ((xx & 1) * 255).astype(np.uint8),
        "half black half white": ((xx >= W // 2) * 255).astype(np.uint8),
        "salt 0.1 % on mid grey": np.where(rng.random((H, W)) < 1e-3, 255, 128).astype(np.uint8),
        "white noise": rng.integers(0, 256, (H, W)).astype(np.uint8),
        "one grey level apart": np.where(rng.random((H, W)) < 0.5, 127, 128).astype(np.uint8),
        "horizontal ramp": (xx * 255 // (W - 1)).astype(np.uint8),
        "synthetic scene": None,
    }
    import synthetic
    record = {}
    for name, img in kinds.items():
        if img is None:
            Ls, Rs = synthetic.make_pair(H, W, D, seed=4)[:2]
            l, r = dev(Ls[:, :, 0]), dev(Rs[:, :, 0])
        else:
            l = dev(std8(img))
            r = dev(std8(np.roll(img, -3, axis=1)))
        net.split_saturated()
        net.features_pair_hwc_split(l, r)
        fired = bool(net.split_saturated())
        record[name] = {"largest_standardised_pixel": float(l.abs().max()), "fallback_fired": fired}
        m = sd.StereoMatcher(net, features="split_f16")
        got = m.match(l, r, D)
        assert (m._library_twin is not None) == fired, name
        if fired:
            want = sd.StereoMatcher(net, features="miopen").match(l, r, D)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), name
    # the single pixel at cfg2's pixel count: the largest value ANY 8-bit Middlebury-half image can standardise to
    big = np.zeros((500, 750), np.uint8); big[250, 375] = 255
    l = dev(std8(big))
    net.features_pair_hwc_split(l, l)
    record["single white pixel on black, 750 x 500"] = {"largest_standardised_pixel": float(l.abs().max()),
                                                        "fallback_fired": bool(net.split_saturated())}
    assert not record["synthetic scene"]["fallback_fired"] and not record["white noise"]["fallback_fired"]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_features_split.json")
    old = {}
    if os.path.isfile(path):
        with open(path) as f:
            old = json.load(f)
    old["saturation_fallback_on_extreme_8bit_images"] = record
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


def test_split_weights_follow_weight_updates(nets):
    """The packed f16 weights are rebuilt when a weight tensor changes in place (training, set_layers)."""
    net = nets["random init"]
    L, R = smooth_pair(24, 40, 1)
    a, _ = net.features_pair_hwc_split(L.cuda(), R.cuda())
    with torch.no_grad():
        net.weights[2].mul_(1.5)
    b, _ = net.features_pair_hwc_split(L.cuda(), R.cuda())
    ref = features_float64(net, L)
    assert float((b.double().cpu() - ref).abs().max()) <= 2e-6
    assert not torch.equal(a, b)
    with torch.no_grad():
        net.weights[2].div_(1.5)


def test_whole_pair_with_split_features_matches_library_features(sd, nets):
    """cfg1-sized pair through the whole timed region, fast stage variants, the two feature paths side by side:
    feature differences of ~1e-6 may flip a WTA tie here and there; nothing else may move."""
    import _hipabi as hip
    import synthetic
    H, W, D = 128, 192, 48
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=5)
    net = nets["converted reference checkpoint"]
    outs = {}
    for feat in ("miopen", "split_f16"):
        m = sd.StereoMatcher(net, cv_mode=hip.MCCNN_CV_MFMA, cbca_order=hip.MCCNN_CBCA_SEPARABLE, features=feat)
        keep = {}
        out = m.match(dev(L[:, :, 0]), dev(R[:, :, 0]), D, keep=keep)
        outs[feat] = (out.cpu().numpy(), keep["wta"][0].cpu().numpy())
    flips = float((outs["miopen"][1] != outs["split_f16"][1]).mean())
    close = float(np.isclose(outs["miopen"][0], outs["split_f16"][0], atol=1e-3, equal_nan=True).mean())
    assert flips <= 0.002, "%.4f of the WTA picks differ" % flips
    assert close >= 0.99, "only %.4f of the final map within 1e-3 px" % close

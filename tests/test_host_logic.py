"""CPU: host-side pieces of the drop-in that need no GPU - checkpoint reader, file formats, the NET head on torch-CPU,
the bilateral kernel table, the CLI flags, the synthetic generator."""
import os
import struct

import numpy as np

from conftest import GOLDEN_DIR
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CKPT = "/root/reference/data/tensorboard_log/model_epoch2000.ckpt"


def test_checkpoint_npz_layers(net_layers):
    assert len(net_layers) == 5
    assert net_layers[0][0].shape == (3, 3, 1, 64) and net_layers[0][1].shape == (64,)
    for w, b in net_layers[1:]:
        assert w.shape == (3, 3, 64, 64) and b.shape == (64,) and w.dtype == np.float32


@pytest.mark.skipif(not os.path.isfile(REF_CKPT + ".index"), reason="reference checkpoint only in the dev container")
def test_tf_bundle_reader_matches_fixture(net_layers):
    import tf_checkpoint
    idx = tf_checkpoint.read_index(REF_CKPT)
    assert len(idx) == 20 and idx["conv2/weights"]["shape"] == [3, 3, 64, 64]
    assert idx["conv1/weights"]["offset"] == 512 and idx["conv5/weights"]["size"] == 147456
    layers = tf_checkpoint.load_fast_net_weights(REF_CKPT)     # CRC32C-verified
    for (w, b), (w2, b2) in zip(layers, net_layers):
        assert np.array_equal(w, w2) and np.array_equal(b, b2)


def test_tf_bundle_reader_detects_corruption(tmp_path):
    import tf_checkpoint
    if not os.path.isfile(REF_CKPT + ".index"):
        pytest.skip("reference checkpoint only in the dev container")
    for ext in (".index", ".data-00000-of-00001"):
        data = bytearray(open(REF_CKPT + ext, "rb").read())
        if ext.startswith(".data"):
            data[600] ^= 0xFF
        open(str(tmp_path / ("bad.ckpt" + ext)), "wb").write(bytes(data))
    with pytest.raises(ValueError, match="crc32c"):
        tf_checkpoint.load_fast_net_weights(str(tmp_path / "bad.ckpt"))
    with pytest.raises(ValueError):
        tf_checkpoint.load_fast_net_weights(None)


def test_net_head_matches_oracle_on_cpu(net_layers):
    """model.NET on torch-CPU (patch path, [B,11,11,1] -> [B,1,1,64]) vs the float64-accumulating restatement."""
    import oracle as o
    from model import NET
    rng = np.random.default_rng(0)
    patches = rng.standard_normal((6, 11, 11, 1)).astype(np.float32)
    net = NET(torch.from_numpy(patches), batch_size=6, device="cpu").set_layers(net_layers)
    feats = net(torch.from_numpy(patches)).numpy()
    assert feats.shape == (6, 1, 1, 64) and net.conv1.shape == (6, 9, 9, 64) and net.conv5.shape == (6, 1, 1, 64)
    assert np.allclose(np.linalg.norm(feats, axis=-1), 1.0, atol=1e-5)
    for i in range(6):
        x = patches[i]
        ref = x
        cur = np.ascontiguousarray(x)
        for k, (w, b) in enumerate(net_layers):
            out = np.empty((cur.shape[0] - 2, cur.shape[1] - 2, 64), np.float32)
            o.lib().orc_conv3x3_valid(o._p(cur), cur.shape[0], cur.shape[1], cur.shape[2], o._p(np.ascontiguousarray(w)),
                                      o._p(np.ascontiguousarray(b)), 64, int(k < 4), o._p(out))
            cur = out
        o.lib().orc_l2_normalize(o._p(cur), 1, 64)
        assert np.abs(cur.reshape(64) - feats[i].reshape(64)).max() <= 2e-6
    assert ref is not None


def test_net_weight_roundtrip(tmp_path, net_layers):
    from model import NET
    import tf_checkpoint
    net = NET(None, device="cpu").set_layers(net_layers)
    for (w, b), (w2, b2) in zip(net.get_layers(), net_layers):
        assert np.array_equal(w, w2) and np.array_equal(b, b2)
    p = str(tmp_path / "w.npz")
    tf_checkpoint.save_npz(p, net.get_layers())
    net2 = NET(None, device="cpu").restore(p)
    assert all(torch.equal(a, b) for a, b in zip(net.weights, net2.weights))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        net.save_weights(file_name="pretrain.npy")
        net3 = NET(None, weights_path=str(tmp_path / "pretrain.npy"), device="cpu")
        net3.load_initial_weights()
        assert all(torch.equal(a, b) for a, b in zip(net.weights, net3.weights))
    finally:
        os.chdir(cwd)


def test_bilateral_table_equals_oracle_and_formula():
    import oracle as o
    import stereo_device as sd
    import util
    t = sd.bilateral_table(5, 5, 0, 6)
    assert np.array_equal(t, o.bilateral_table(5, 5, 0, 6))
    g = util.normal(0, 6)
    assert t[2, 2] == np.float32(g(0.0)) and t[0, 0] == np.float32(g(np.sqrt(8.0)))


def test_pfm_roundtrip_and_layout(tmp_path):
    import util
    rng = np.random.default_rng(1)
    d = rng.random((7, 11), dtype=np.float32) * 100
    p = str(tmp_path / "d.pfm")
    util.writePfm(d, p)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n11 7\n-1.0\n")                       # util.py:60-62
    body = raw[len(b"Pf\n11 7\n-1.0\n"):]
    assert struct.unpack("<f", body[:4])[0] == d[6, 0]               # bottom row first, little endian
    assert np.array_equal(util.readPfm(p), d)


def test_calib_time_pgm(tmp_path):
    import util
    calib = tmp_path / "calib.txt"
    calib.write_text("cam0=[1 0 0; 0 1 0; 0 0 1]\ncam1=[1 0 0; 0 1 0; 0 0 1]\ndoffs=0\nbaseline=100\n"
                     "width=750\nheight=500\nndisp=256\nisint=0\n")
    assert util.parseCalib(str(calib)) == (500, 750, 256)            # (height, width, ndisp), util.py:27-43
    util.saveTimeFile(1.25, str(tmp_path / "t.txt"))
    assert (tmp_path / "t.txt").read_text() == "1.25"
    d = np.array([[0.4, 0.5, 1.5, 2.5], [254.6, 300.0, -3.0, np.nan]], np.float32)
    util.saveDisparity(d, str(tmp_path / "d.pgm"))
    raw = (tmp_path / "d.pgm").read_bytes()
    assert raw.startswith(b"P5\n4 2\n255\n")
    assert list(raw[-8:]) == [0, 0, 2, 2, 255, 255, 0, 0]           # round-half-even, saturate
    util.recurMk(str(tmp_path / "a" / "b" / "c"))
    assert (tmp_path / "a" / "b" / "c").is_dir()


def test_read_gray(tmp_path):
    import util
    from PIL import Image
    rng = np.random.default_rng(2)
    g = rng.integers(0, 256, size=(5, 6), dtype=np.uint8)
    Image.fromarray(g).save(str(tmp_path / "g.png"))
    assert np.array_equal(util.read_gray(str(tmp_path / "g.png")), g)
    rgb = rng.integers(0, 256, size=(5, 6, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(str(tmp_path / "c.png"))
    got = util.read_gray(str(tmp_path / "c.png")).astype(int)
    lum = 0.299 * rgb[:, :, 0] + 0.587 * rgb[:, :, 1] + 0.114 * rgb[:, :, 2]
    assert np.abs(got - lum).max() <= 1.0


def test_read_gray_matches_libpng_fixture():
    """Colour PNGs decoded to grey by the REAL libpng 1.6.37 set up as OpenCV's reader sets it up for IMREAD_GRAYSCALE
    (tests/golden/gen_png_gray.c, run in the build container): util.read_gray must give the same bytes."""
    import util
    for name in ("rgb", "rgba"):
        want = np.load(os.path.join(GOLDEN_DIR, "png_gray_%s.npy" % name))
        got = util.read_gray(os.path.join(GOLDEN_DIR, "png_color_%s.png" % name))
        assert got.dtype == np.uint8 and np.array_equal(got, want), name
    # and the arithmetic spelled out once more, independently of util: truncating 15-bit weights 9797 / 19234 / 3737
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(GOLDEN_DIR, "png_color_rgb.png")).convert("RGB")).astype(np.int64)
    want = np.load(os.path.join(GOLDEN_DIR, "png_gray_rgb.npy"))
    for y, x in ((0, 0), (9, 17), (23, 39), (12, 5)):
        r, g, b = (int(v) for v in rgb[y, x])
        assert (r * 9797 + g * 19234 + b * 3737) >> 15 == int(want[y, x])


def test_drop_in_defaults_are_the_bit_exact_variants():
    import match
    import process_functional as pf
    assert pf.COST_VOLUME_MODE == "exact" and pf.CBCA_ORDER == "reference" and pf.FEATURES == "split_f16"
    assert match.parser.parse_args(["--list_file", "l", "--data_dir", "d", "--save_dir", "s", "-t", "x", "-s", "0", "-e",
                                    "3"]).features is None          # None = the hand-written kernels where they apply
    base = ["--list_file", "l", "--data_dir", "d", "--save_dir", "s", "-t", "x", "-s", "0", "-e", "3"]
    assert match.parser.parse_args(base).fast is False
    assert match.parser.parse_args(base + ["--fast"]).fast is True


def test_recurmk_survives_concurrent_creation(tmp_path):
    """Several ranks create the same output tree at once (match.py under torchrun): no check-then-create race."""
    import threading
    import util
    target = str(tmp_path / "submit_t" / "a" / "b")
    errors = []

    def work():
        try:
            for _ in range(50):
                util.recurMk(target)
                util.testMk(target)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work) for _ in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors and os.path.isdir(target)


def test_cli_flags_match_reference():
    import match
    a = match.parser.parse_args(["--list_file", "l", "--data_dir", "d", "--save_dir", "s", "-t", "x", "-s", "0",
                                 "-e", "3"])
    assert (a.gpu, a.patch_size, a.resume) == ("0", 11, None)
    hp = match.hyper_parameters(a)
    assert hp == dict(cbca_intensity=0.02, cbca_distance=14, cbca_num_iterations1=2, cbca_num_iterations2=16,
                      sgm_P1=2.3, sgm_P2=55.9, sgm_Q1=4, sgm_Q2=8, sgm_D=0.08, sgm_V=1.5, blur_sigma=6,
                      blur_threshold=2)
    assert isinstance(hp["cbca_distance"], int) and isinstance(hp["cbca_num_iterations2"], int)
    assert (match.out_file, match.out_img_file, match.out_time_file) == ("disp0MCCNN.pfm", "disp0MCCNN.pgm",
                                                                         "timeMCCNN.txt")


def test_gpu_flag_pins_the_card_only_when_given(monkeypatch):
    """-g in every spelling argparse accepts (match.py:17, train.py:17) is explicit and overrides HIP_VISIBLE_DEVICES; the
    default "0" only fills an empty environment; under torchrun (world > 1) the flag is ignored.  One helper for
    match.py and train.py."""
    import match
    import train
    import util
    need = {match: ["--list_file", "l", "--data_dir", "d", "--save_dir", "s", "-t", "x", "-s", "0", "-e", "3"], train: ["--list_dir", "l", "--tensorboard_dir", "t", "--checkpoint_dir", "c"]}
    for mod in (match, train):
        for words, explicit, value in ((["-g", "3"], True, "3"), (["-g3"], True, "3"), (["--gpu=2"], True, "2"),
                                       (["--gp", "5"], True, "5"), ([], False, "0")):
            a = mod.parser.parse_known_args(need[mod] + words)[0]
            assert (a.gpu, getattr(a, "gpu_explicit", False)) == (value, explicit), (mod.__name__, words)
            monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6")
            util.pin_gpu(a, 1)
            assert os.environ["HIP_VISIBLE_DEVICES"] == (value if explicit else "6")
            monkeypatch.delenv("HIP_VISIBLE_DEVICES")
            util.pin_gpu(a, 1)
            assert os.environ["HIP_VISIBLE_DEVICES"] == value and os.environ["CUDA_VISIBLE_DEVICES"] == value
            monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6")
            util.pin_gpu(a, 8)
            assert os.environ["HIP_VISIBLE_DEVICES"] == "6"


def test_stated_tolerance_helper():
    import tolerances as tol
    assert tol.fast_violations(750000, 0, 0, 0.99, 0.005) == []
    assert len(tol.fast_violations(750000, 76, 0, 0.97, 0.03)) == 3
    assert tol.fast_violations(1500000, 98, 0, 0.9875, 0.0015) == []   # round 4's cfg4 numbers are inside the statement


def test_synthetic_pairs_are_deterministic_and_standardised():
    import synthetic
    a = synthetic.make_pair(40, 60, 16, seed=5)
    b = synthetic.make_pair(40, 60, 16, seed=5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    L = a[0]
    assert L.shape == (40, 60, 1) and L.dtype == np.float32
    assert abs(float(L.mean())) < 1e-5 and abs(float(L.std()) - 1.0) < 1e-4
    assert a[2].dtype == np.uint8 and len(np.unique(a[2])) > 8


def test_synthetic_scene_classes():
    """The three seeded scene classes bench.py times (synthetic.SCENE_KINDS): the default recipe, its texture-free variant
    (= texture=False) and the 1/f-spectrum "natural" picture; what distinguishes them for the aggregation is the share of
    pixels whose support region is the pixel itself (pf:580-629 with match.py's threshold 0.02 on the standardised image)."""
    import oracle as o
    import synthetic
    assert synthetic.SCENE_KINDS == ("blobs+texture", "flat", "natural")
    H, W, D = 96, 128, 16
    unit = {}
    for kind in synthetic.SCENE_KINDS:
        L, R, lu, ru, dmap = synthetic.make_pair(H, W, D, seed=7, kind=kind)
        L2 = synthetic.make_pair(H, W, D, seed=7, kind=kind)[0]
        assert np.array_equal(L, L2) and L.shape == (H, W, 1) and lu.dtype == np.uint8 and dmap.shape == (H, W)
        assert abs(float(L.mean())) < 1e-5 and abs(float(L.std()) - 1.0) < 1e-4
        _, cnt = o.cross_arms(L, 0.02, 14)
        unit[kind] = float((cnt == 1).mean())
    assert np.array_equal(synthetic.make_pair(H, W, D, seed=7, texture=False)[0], synthetic.make_pair(H, W, D, seed=7, kind="flat")[0])
    assert np.array_equal(synthetic.make_pair(H, W, D, seed=7)[0], synthetic.make_pair(H, W, D, seed=7, kind="blobs+texture")[0])
    assert unit["flat"] < 0.05 < unit["blobs+texture"] < 0.8 < unit["natural"], unit
    with pytest.raises(ValueError):
        synthetic.make_pair(H, W, D, kind="photograph")
    # 8-bit picture with a photograph's contrast (std 40-60 grey levels before standardisation)
    nat = synthetic.natural_scene_u8(H, W, seed=1)
    assert nat.dtype == np.uint8 and 35.0 < float(nat.std()) < 60.0


def test_shard_indices_cover_window_once():
    from distributed import shard_indices
    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in shard_indices(2, 12, 10, r, world))
        assert got == list(range(2, 10))                              # inclusive window, clipped to the list
        sizes = [len(shard_indices(2, 12, 10, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1

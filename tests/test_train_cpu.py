"""CPU: the training path - datagenerator.ImageDataGenerator and train.py as drop-ins for the reference's
train.py / datagenerator.py: sampling rules (datagenerator.py:137-216), hinge loss and momentum update
(train.py:85-106), checkpoints that match.py / NET.restore read back, and synchronous data parallelism over gloo."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

SRC = os.path.join(ROOT, "mc-cnn-python_amd", "src")


def _write_dataset(root, n_pairs=3, H=40, W=72, seed=0):
    """Random-texture left views, right views shifted by a piecewise-constant integer disparity, ground truth with a
    band of unknown (inf) pixels; lists train.txt / val.txt."""
    from PIL import Image
    import util
    rng = np.random.default_rng(seed)
    lefts = []
    for i in range(n_pairs):
        d = os.path.join(root, "pair%d" % i)
        os.makedirs(d)
        scene = rng.integers(0, 256, size=(H, W + 32)).astype(np.uint8)
        gt = np.full((H, W), 4.0, np.float32)
        gt[H // 2:] = 9.0
        left = scene[:, 16:16 + W]
        right = np.zeros_like(left)
        for y in range(H):
            s = int(gt[y, 0])
            right[y] = scene[y, 16 + s:16 + s + W]              # right[y, x - s] = left[y, x]
        gt[:, :3] = np.inf                                       # unknown band: never sampled
        Image.fromarray(left, "L").save(os.path.join(d, "im0.png"))
        Image.fromarray(right, "L").save(os.path.join(d, "im1.png"))
        util.writePfm(gt, os.path.join(d, "disp0GT.pfm"))
        lefts.append(os.path.join(d, "im0.png"))
    lists = os.path.join(root, "lists")
    os.makedirs(lists)
    open(os.path.join(lists, "train.txt"), "w").write("\n".join(lefts[:-1]) + "\n")
    open(os.path.join(lists, "val.txt"), "w").write(lefts[-1] + "\n")
    return lists


def test_sampler_rules(tmp_path):
    from datagenerator import ImageDataGenerator
    lists = _write_dataset(str(tmp_path))
    g = ImageDataGenerator(os.path.join(lists, "train.txt"), shuffle=False, rng=np.random.default_rng(3))
    assert g.data_size == 2 and g.pointer == 0
    assert g.right_paths[0].endswith("im1.png") and g.gt_paths[0].endswith("disp0GT.pfm")
    img = g.left_images[0]
    assert abs(float(img.mean())) < 1e-5 and abs(float(img.std()) - 1.0) < 1e-4         # standardised (:88-90)
    left, pos, neg = g.next_batch(32)
    assert g.pointer == 1
    for a in (left, pos, neg):
        assert a.shape == (32, 11, 11, 1) and a.dtype == np.float32
    # positives: the right patch centred within one pixel of the true match has the same centre value as the left one
    # (textures are independent noise, so a different column would not); negatives never do
    Lc, Pc, Nc = left[:, 5, 5, 0], pos[:, 5, 5, 0], neg[:, 5, 5, 0]
    scale_l, scale_r = g.left_images[0].std(), g.right_images[0].std()
    same_pos = np.isclose(Lc * 1.0, Pc, atol=0.2).mean()
    assert same_pos >= 0.4, same_pos          # int(right_col + U(-0.5, 0.5)) is right_col or right_col - 1
    assert np.isclose(Lc, Nc, atol=1e-6).mean() <= 0.1
    # exhausting the list and resetting
    g.next_batch(8)
    with pytest.raises(IndexError):
        g.next_batch(8)
    g.reset_pointer()
    assert g.pointer == 0
    l, r, gt = g.next_pair()
    assert l.shape == r.shape == gt.shape and np.isinf(gt[:, :3]).all()


def test_sampler_offsets_exact(tmp_path):
    """Replays the sampler's geometry: with dataset_pos = 0 the positive patch IS the true match; negatives lie
    1.5 .. 6 px away after int() truncation, on both sides; unknown and occluded pixels are never centres."""
    from datagenerator import ImageDataGenerator
    lists = _write_dataset(str(tmp_path), seed=5)
    g = ImageDataGenerator(os.path.join(lists, "train.txt"), dataset_pos=0.0, rng=np.random.default_rng(11))
    pl, pr = g._padded(g.left_images[0]), g._padded(g.right_images[0])
    left, pos, neg = g.next_batch(40)
    H, W = g.left_images[0].shape
    gt = g.gt_images[0]
    found = 0
    for b in range(40):
        # locate the left patch in the padded left image (random texture: unique)
        hits = [(y, x) for y in range(H) for x in range(W) if np.array_equal(pl[y:y + 11, x:x + 11], left[b, :, :, 0])]
        assert len(hits) == 1
        y, x = hits[0]
        assert np.isfinite(gt[y, x]) and int(gt[y, x]) <= x
        rc = x - int(gt[y, x])
        assert np.array_equal(pr[y:y + 11, rc:rc + 11], pos[b, :, :, 0])
        offs = [c - rc for c in range(W) if np.array_equal(pr[y:y + 11, c:c + 11], neg[b, :, :, 0])]
        assert len(offs) == 1 and 1 <= abs(offs[0]) <= 6, offs
        found += offs[0] < 0
    assert 5 <= found <= 35                                      # both signs occur


def test_momentum_step_is_tensorflows(tmp_path):
    """One Trainer.step equals accum = beta*accum + grad; var -= lr*accum (tf.train.MomentumOptimizer, train.py:105),
    and the loss is mean(max(0, margin - <fl,fr+> + <fl,fr->)) on L2-normalised features (train.py:85-93)."""
    from model import NET
    import train
    rng = np.random.default_rng(0)
    batch = [rng.standard_normal((6, 11, 11, 1)).astype(np.float32) for _ in range(3)]
    net = NET(None, batch_size=6, device="cpu", seed=1)
    ref = NET(None, batch_size=6, device="cpu", seed=1)
    t = train.Trainer(net, 0.05, 0.9, 0.2)
    w0 = [w.detach().clone() for w in net.weights]
    f = [ref(torch.from_numpy(b)).reshape(6, 64) for b in batch]
    want = torch.clamp(0.2 - (f[0] * f[1]).sum(-1) + (f[0] * f[2]).sum(-1), min=0).mean()
    l1 = t.step(*batch)
    assert abs(l1 - float(want)) < 1e-6
    g1 = [(w0[k] - net.weights[k].detach()) / 0.05 for k in range(5)]          # first step: accum = grad
    w1 = [w.detach().clone() for w in net.weights]
    t.opt.zero_grad()
    t.loss(*batch).backward()
    g2 = [net.weights[k].grad.clone() for k in range(5)]
    t.step(*batch)
    for k in range(5):
        accum = 0.9 * g1[k] + g2[k]
        assert torch.allclose(net.weights[k].detach(), w1[k] - 0.05 * accum, atol=1e-6)


def test_train_cli_learns_and_checkpoint_round_trips(tmp_path):
    import tf_checkpoint
    import train
    from model import NET
    lists = _write_dataset(str(tmp_path), n_pairs=4, seed=2)
    log, ck = str(tmp_path / "log"), str(tmp_path / "ck")
    train.main(["--list_dir", lists, "--tensorboard_dir", log, "--checkpoint_dir", ck, "-bs", "32", "-lr", "0.05",
                "--end_epoch", "12", "--print_freq", "1", "--save_freq", "6", "--val_freq", "3", "--seed", "4"])
    pts = [json.loads(x) for x in open(os.path.join(log, "scalars.jsonl"))]
    tr = [p["value"] for p in pts if p["tag"] == "hinge_loss"]
    va = [p for p in pts if p["tag"] == "val_hinge_loss"]
    assert len(tr) == 12 * 3 and [p["step"] for p in va] == [9, 18, 27, 36]    # reference step numbering
    assert np.mean(tr[-6:]) < 0.7 * np.mean(tr[:6]), (tr[:6], tr[-6:])          # it learns
    assert va[-1]["value"] < va[0]["value"]
    files = sorted(os.listdir(ck))
    assert files == ["model_epoch12.ckpt.npz", "model_epoch6.ckpt.npz"]
    layers = tf_checkpoint.load_fast_net_weights(os.path.join(ck, "model_epoch12.ckpt.npz"))
    assert len(layers) == 5 and layers[0][0].shape == (3, 3, 1, 64) and layers[1][0].shape == (3, 3, 64, 64)
    z = np.load(os.path.join(ck, "model_epoch12.ckpt.npz"))
    assert z["conv3/weights/Momentum"].shape == (3, 3, 64, 64)
    # resuming restores weights and momentum: one more epoch from the checkpoint == training one epoch further
    net = NET(None, device="cpu").restore(os.path.join(ck, "model_epoch12.ckpt.npz"))
    assert np.array_equal(net.get_layers()[2][0], layers[2][0])
    train.main(["--list_dir", lists, "--tensorboard_dir", log, "--checkpoint_dir", ck, "-bs", "32", "-lr", "0.05",
                "--resume", os.path.join(ck, "model_epoch12.ckpt.npz"), "--start_epoch", "12", "--end_epoch", "13"])
    assert os.path.isfile(os.path.join(ck, "model_epoch13.ckpt.npz"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, SRC)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import distributed as mgpu
    import train
    from model import NET
    mgpu.init("gloo")
    rng = np.random.default_rng(100 + rank)                     # every rank its own batch
    batch = [rng.standard_normal((4, 11, 11, 1)).astype(np.float32) for _ in range(3)]
    t = train.Trainer(NET(None, batch_size=4, device="cpu", seed=9), 0.1, 0.9, 0.2)
    t.step(*batch)
    t.step(*batch)
    q.put((rank, [w.detach().numpy().copy() for w in t.net.weights], batch))
    mgpu.finalize()


def test_data_parallel_step_averages_gradients_over_gloo():
    """world_size 2: after two steps both ranks hold identical weights, equal to one process stepping on the mean of
    the two ranks' gradients (what the all_reduce over RCCL does on the GPU node)."""
    import train
    from model import NET
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)
    t = train.Trainer(NET(None, batch_size=4, device="cpu", seed=9), 0.1, 0.9, 0.2)
    for _ in range(2):
        t.opt.zero_grad()
        (0.5 * (t.loss(*res[0][2]) + t.loss(*res[1][2]))).backward()
        t.opt.step()
    for k in range(5):
        assert np.allclose(t.net.weights[k].detach().numpy(), res[0][1][k], atol=1e-6)

"""GPU: the match.py drop-in end to end on files - the reference's command line (match.py:14-46), its directory
layout and output files (match.py:99-110, 182-184), checked against the CPU checker fed the same decoded images."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
import tolerances as tol

pytestmark = pytest.mark.gpu


def _write_pair(dirname, H, W, ndisp, seed):
    from PIL import Image
    import synthetic
    os.makedirs(dirname)
    L, R, _, _, _ = synthetic.make_pair(H, W, ndisp, seed=seed)
    for name, img in (("im0.png", L), ("im1.png", R)):
        g = img[:, :, 0]
        g8 = np.clip((g - g.min()) / (g.max() - g.min()) * 255.0, 0, 255).astype(np.uint8)
        Image.fromarray(g8, mode="L").save(os.path.join(dirname, name))
    with open(os.path.join(dirname, "calib.txt"), "w") as f:
        f.write("cam0=[1 0 0; 0 1 0; 0 0 1]\ncam1=[1 0 0; 0 1 0; 0 0 1]\ndoffs=0\nbaseline=100\n"
                "width=%d\nheight=%d\nndisp=%d\nisint=0\nvmin=0\nvmax=%d\ndyavg=0\ndymax=0\n" % (W, H, ndisp, ndisp))


@pytest.mark.parametrize("extra,threshold", [([], 0.999), (["--features", "library"], 0.999)],
                         ids=["default_hand_written_features", "library_features"])
def test_match_cli_writes_reference_outputs(tmp_path, net_layers, extra, threshold):
    import oracle as o
    import util
    data = tmp_path / "data"
    out = tmp_path / "out"
    H, W, D = 40, 64, 16
    rels = ["trainingH/pairA", "trainingH/pairB"]
    for i, rel in enumerate(rels):
        _write_pair(str(data / rel), H, W, D, seed=20 + i)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s/im0.png\n" % (data / rel) for rel in rels))
    cmd = [sys.executable, os.path.join(ROOT, "mc-cnn-python_amd", "src", "match.py"), "-g", "0",
           "--list_file", str(lst), "--resume", os.path.join(GOLDEN_DIR, "mccnn_fast_weights.npz"),
           "--data_dir", str(data), "--save_dir", str(out), "-t", "t1", "-s", "0", "-e", "1"] + extra   # default: bit-exact
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    for rel in rels:
        res = out / "submit_t1" / rel
        img = out / "submit_t1_imgs" / rel
        assert (res / "disp0MCCNN.pfm").is_file() and (res / "timeMCCNN.txt").is_file()
        assert (img / "disp0MCCNN.pgm").is_file()
        assert float((res / "timeMCCNN.txt").read_text().strip()) > 0.0
        disp = util.readPfm(str(res / "disp0MCCNN.pfm"))
        disp = disp[0] if isinstance(disp, tuple) else disp
        disp = np.asarray(disp, np.float32).reshape(H, W)
        # the same decode + standardisation as match.py:118-123, then the CPU checker's whole timed region
        imgs = []
        for name in ("im0.png", "im1.png"):
            g = util.read_gray(str(data / rel / name)).astype(np.float32)
            imgs.append(np.expand_dims((g - np.mean(g, axis=(0, 1))) / np.std(g, axis=(0, 1)), 2))
        want = o.match_pair(imgs[0], imgs[1], D, net_layers)
        close = np.isclose(disp, want, atol=1e-3, equal_nan=True).mean()
        # the SAME threshold for both feature paths: each is ~3e-7 from the checker's float64-accumulating features;
        # everything behind the features is the same bit-exact code
        assert close >= threshold, "%s: only %.4f of pixels within 1e-3 px of the CPU checker" % (rel, close)


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank), on a one-GPU
    box: the two ranks share the GPU and rendezvous over gloo (MCCNN_BENCH_SHARED_GPU=1).  Rank 0 prints ONE JSON
    line with the whole-job rate over the slower rank's time."""
    import json
    env = dict(os.environ, MCCNN_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--config", "cfg1"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "Mdisparities/s" and d["cpu_baseline"] is None
    assert [x["rank"] for x in d["per_rank_device"]] == [0, 1] and d["process_group"] == "gloo x2"
    assert d["dtype"] == "f32" and d["parity"]["final_map_bit_identical"] and d["parity_violations"] == []
    # value = 2 ranks x 256*256*64 voxels x 3 steps / (max-rank seconds): consistent with ms_per_step
    want = 2 * 256 * 256 * 64 * 1e-6 / (d["ms_per_step"] * 1e-3)
    assert abs(d["value"] - want) <= 0.02 * want


def test_bench_launches_itself_for_several_gpus():
    """`python bench.py --gpus 2` started bare (no torchrun, no WORLD_SIZE): the script re-launches itself under
    torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line with both ranks' times - what a driver that
    calls the N-GPU bench like the 1-GPU bench gets.  (Two ranks share the one GPU of the test box.)"""
    import json
    env = dict(os.environ, MCCNN_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config",
           "cfg1", "--no-parity"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["process_group"] == "gloo x2"
    assert abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) <= 0.01 * d["ms_per_step"] + 1e-3


def test_bench_eight_ranks_on_the_shared_gpu():
    """`python bench.py --gpus 8` exactly as the driver will start it on an 8-GPU node - bare, the script launches itself
    under torch.distributed.run on 127.0.0.1 - with all eight ranks sharing the one GPU of the test box (gloo rendezvous,
    MCCNN_BENCH_SHARED_GPU=1: RCCL refuses two ranks per device).  The launcher, the rendezvous, the barrier + gather of
    eight times and eight device records, and the one JSON line with the whole-job rate run end to end
    (/root/reference/src/match.py:85-91: eight independent index windows)."""
    import json
    env = dict(os.environ, MCCNN_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--config",
           "cfg1", "--no-parity"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["process_group"] == "gloo x8"
    assert len(d["per_rank_ms_per_step"]) == 8 and [x["rank"] for x in d["per_rank_device"]] == list(range(8))
    assert abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) <= 0.01 * d["ms_per_step"] + 1e-3
    want = 8 * 256 * 256 * 64 * 1e-6 / (d["ms_per_step"] * 1e-3)
    assert abs(d["value"] - want) <= 0.02 * want


def test_bench_with_library_features():
    """bench.py --library-features: the same bit-exact stages behind MIOpen's float32 convolutions; its plane-major twin
    (same features) still agrees bit for bit."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--config", "cfg1",
           "--library-features", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert d["roofline"]["kernel"] in ("cbca_iter_prog_pair", "cbca_iter_prog_pair_skip", "cbca_iter_prog", "cbca_iter_prog_skip") and "MIOpen" in d["config"]["features"]
    assert d["parity"]["timed_path_equals_kernel_by_kernel"] and d["parity"]["final_map_bit_identical"]
    assert d["parity_violations"] == [] and d["dtype"] == "f32"


@pytest.mark.parametrize("extra,kernels", [([], ("cbca_iter_prog_pair", "cbca_iter_prog_pair_skip", "cbca_iter_prog", "cbca_iter_prog_skip")),
                                           (["--separable-cbca"], ("cbca_iter_pair",))],
                         ids=["matrix_core_cost_volume", "and_separable_aggregation"])
def test_bench_fast_variant_states_and_meets_its_tolerance(extra, kernels):
    """bench.py --fast [--separable-cbca]: the tolerance-checked variants; the line states src/tolerances.py's limits,
    measures them on the timed pair and the script would exit non-zero on a violation."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--config", "cfg1", "--fast",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert d["roofline"]["kernel"] in kernels and d["config"]["variant"].startswith("fast")
    assert "split-f16" in d["dtype"] and d["parity_violations"] == [] and d["exact_variant_ms_per_step"] > 0
    pp = d["parity"]
    assert not tol.fast_violations(pp["pixels"], pp["wta_flips_left"], pp["wta_flips_right"], pp["frac_within_1e-3_px"],
                                   pp["p99_abs_px"])


def test_bench_world_size_one_through_rccl():
    """bench.py with the process group forced on at world size 1 (MCCNN_BENCH_FORCE_DIST=1): backend "nccl" = RCCL is
    initialised with a device id, and the barrier + all_gather of the multi-GPU path run on a device tensor - what an
    8-GPU launch does per rank, minus the peers.  Also checks the JSON contract of the default (hipGraph) launch mode."""
    import json
    env = dict(os.environ, MCCNN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config",
           "cfg1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["process_group"] == "nccl x1", d["process_group"]
    assert d["n_gpus"] == 1 and d["config"]["launch"].startswith("one hipGraph replay per pair; each volume's aggregation")
    # the benchmarked variant is the drop-in default: float32, bit-exact, its twin agrees bit for bit
    assert d["roofline"]["kernel"] in ("cbca_iter_prog_pair", "cbca_iter_prog_pair_skip", "cbca_iter_prog", "cbca_iter_prog_skip") and 0.0 < d["roofline"]["frac"] < 1.5
    assert d["dtype"] == "f32" and d["config"]["variant"].startswith("bit-exact")
    assert d["config"]["features"].startswith("hand-written matrix-core")
    assert d["parity"]["final_map_bit_identical"] and d["parity"]["timed_path_equals_kernel_by_kernel"]
    assert d["parity_violations"] == [] and d["fast_variant_ms_per_step"] > 0
    assert d["fast_variant_parity"]["violations"] == [], d["fast_variant_parity"]
    # round 6: the schedule's side numbers, the natural-statistics scene, pairs in flight, per-chain stage brackets
    assert d["ms_per_step_chains_joined_after_every_stage"] > 0 and d["natural_scene_ms_per_step"] > 0
    assert d["natural_scene"]["unit_region_pixels"]["left"] > 0.8 > d["unit_region_pixels"]["left"]
    for key in ("ms_per_pair_two_in_flight", "ms_per_pair_four_in_flight"):
        assert d[key]["ms_per_pair"] > 0 and d[key]["final_map_equals_the_timed_pair"], d[key]
    assert d["aggregation_stages"]["aggregation_1"]["skip_iterations"] == 1          # refresh, skip
    assert d["aggregation_stages"]["aggregation_2"]["skip_iterations"] == 14         # full, 14 x skip, full + WTA
    assert d["rooflines"]["sgm_pass_one_volume"]["concurrent_launches"] == 2 and d["sgm_stage"]["ms"] > 0
    assert len(d["per_rank_device"]) == 1 and d["per_rank_device"][0]["device"] == 0
    want = 256 * 256 * 64 * 1e-6 / (d["ms_per_step"] * 1e-3)
    assert abs(d["value"] - want) <= 0.02 * want
    assert d["ms_per_step_host_in_host_out"] > 0 and d["ms_per_step_kernel_by_kernel"] > 0


def test_match_cli_two_ranks_write_disjoint_complete_outputs(tmp_path):
    """match.py under torch.distributed.run with two ranks (sharing the one GPU of the test box): rank r takes the pairs
    i = r (mod 2) of the window; together they write every pair exactly once, and what they write equals a single-rank
    run of the same command bit for bit."""
    data = tmp_path / "data"
    H, W, D = 32, 48, 8
    rels = ["setA/p%d" % i for i in range(5)]
    for i, rel in enumerate(rels):
        _write_pair(str(data / rel), H, W, D, seed=40 + i)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s/im0.png\n" % (data / rel) for rel in rels))
    common = ["--list_file", str(lst), "--resume", os.path.join(GOLDEN_DIR, "mccnn_fast_weights.npz"),
              "--data_dir", str(data), "-s", "1", "-e", "4"]
    script = os.path.join(ROOT, "mc-cnn-python_amd", "src", "match.py")
    env = dict(os.environ, MCCNN_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    two = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29635", script] + common + ["--save_dir", str(tmp_path / "two"), "-t", "r"]
    r = subprocess.run(two, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    log = r.stdout.decode()
    import re
    owners = {}
    for rk, idx in re.findall(r"\[(\d+)\] pair (\d+):", log):      # the two ranks' lines may interleave
        owners.setdefault(int(idx), []).append(int(rk))
    assert owners == {1: [0], 2: [1], 3: [0], 4: [1]}, owners          # window [1,4], round-robin from its start
    one = [sys.executable, script] + common + ["-g", "0", "--save_dir", str(tmp_path / "one"), "-t", "r"]
    r = subprocess.run(one, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert not (tmp_path / "two" / "submit_r" / rels[0]).exists()      # index 0 is outside the window
    for rel in rels[1:]:
        for root, name in (("submit_r", "disp0MCCNN.pfm"), ("submit_r_imgs", "disp0MCCNN.pgm")):
            a = (tmp_path / "two" / root / rel / name).read_bytes()
            b = (tmp_path / "one" / root / rel / name).read_bytes()
            assert a == b and len(a) > H * W, (rel, name)
        assert float((tmp_path / "two" / "submit_r" / rel / "timeMCCNN.txt").read_text()) > 0


def test_training_step_on_the_gpu_and_weights_feed_the_matcher(net_layers):
    """train.Trainer on cuda:0 (MIOpen forward + backward): the hinge loss of a fixed batch falls under momentum SGD,
    and the trained weights go straight into the matching pipeline."""
    import _hipabi as hip
    import stereo_device as sd
    import synthetic
    import train
    from model import NET
    rng = np.random.default_rng(0)
    base = rng.standard_normal((48, 11, 11, 1)).astype(np.float32)
    batch = [base, base + 0.05 * rng.standard_normal(base.shape).astype(np.float32),
             rng.standard_normal(base.shape).astype(np.float32)]
    net = NET(None, batch_size=48, device="cuda", seed=3)
    t = train.Trainer(net, 0.02, 0.9, 0.2)
    losses = [t.step(*batch) for _ in range(40)]
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    inf = NET(None, input_patch_size=11, batch_size=1, device="cuda").set_layers(net.get_layers())
    H, W, D = 40, 64, 8
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=5)
    m = sd.StereoMatcher(inf, cv_mode=hip.MCCNN_CV_EXACT, cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER)
    import torch
    out = m.match(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), D)
    assert out.shape == (H, W) and bool(torch.isfinite(out).all())


def test_training_step_matches_float64_evaluation_of_the_reference_graph():
    """(f3) pinned the way a1 is: TWO Trainer.step calls on cuda:0 (MIOpen forward / backward / weight gradients)
    against a float64 evaluation, on the CPU, of what /root/reference/src/train.py:71-106 builds - three weight-shared
    towers of five VALID 3x3 convolutions with ReLU (model.py:51-61), tf.nn.l2_normalize (model.py:64), cosine of the
    unit vectors (train.py:85-87), hinge with margin (train.py:90-93), tf.train.MomentumOptimizer (accum = beta * accum
    + grad; var -= lr * accum, train.py:105-106).  The arithmetic itself lives in TensorFlow (parity unpinned at that
    boundary, like the feature stage); tolerance 1e-5 on both losses and on all ten updated tensors."""
    import torch
    import torch.nn.functional as F
    import train
    from model import NET
    rng = np.random.default_rng(5)
    B, lr, beta, margin = 64, 0.02, 0.9, 0.2
    base = rng.standard_normal((B, 11, 11, 1)).astype(np.float32)
    batches = []
    for _ in range(2):
        batches.append([base + 0.1 * rng.standard_normal(base.shape).astype(np.float32),
                        base + 0.1 * rng.standard_normal(base.shape).astype(np.float32),
                        rng.standard_normal(base.shape).astype(np.float32)])
    net = NET(None, batch_size=B, device="cuda", seed=7)
    start = net.get_layers()                                   # HWIO weights + biases, float32
    t = train.Trainer(net, lr, beta, margin)
    gpu_losses = [t.step(*b) for b in batches]
    got = net.get_layers()

    # float64 evaluation with its own statement of the graph (no code shared with model.NET / train.hinge_loss)
    ws = [torch.tensor(np.transpose(w, (3, 2, 0, 1)).astype(np.float64), requires_grad=True) for w, _ in start]
    bs = [torch.tensor(b.astype(np.float64), requires_grad=True) for _, b in start]
    acc = [torch.zeros_like(p) for p in ws + bs]
    ref_losses = []
    for batch in batches:
        feats = []
        for patches in batch:                                  # the three towers share ws / bs
            x = torch.tensor(patches.astype(np.float64)).permute(0, 3, 1, 2)
            for k in range(5):
                x = F.conv2d(x, ws[k], bs[k])
                if k < 4:
                    x = torch.relu(x)
            x = x.reshape(B, 64)
            feats.append(x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12)))
        pos = (feats[0] * feats[1]).sum(dim=1)
        neg = (feats[0] * feats[2]).sum(dim=1)
        loss = torch.clamp(margin - pos + neg, min=0.0).mean()
        ref_losses.append(float(loss.detach()))
        grads = torch.autograd.grad(loss, ws + bs)
        with torch.no_grad():
            for p, a, g in zip(ws + bs, acc, grads):
                a.mul_(beta).add_(g)
                p.sub_(lr * a)
    assert ref_losses[0] > 0.01, "degenerate test batch: the hinge is inactive"
    for a, b in zip(gpu_losses, ref_losses):
        assert abs(a - b) <= 1e-5, (gpu_losses, ref_losses)
    worst = 0.0
    for k, (w, b) in enumerate(got):
        worst = max(worst, float(np.abs(np.transpose(w, (3, 2, 0, 1)) - ws[k].detach().numpy()).max()),
                    float(np.abs(b - bs[k].detach().numpy()).max()))
    moved = max(float(np.abs(got[k][0] - start[k][0]).max()) for k in range(5))
    assert moved > 1e-4, "the step did not move the weights"
    assert worst <= 1e-5, "updated tensors differ from the float64 evaluation by %g" % worst


def test_match_cli_pairs_in_flight_writes_the_same_files(tmp_path):
    """--pairs_in_flight 2 (two matchers, two streams, pairs of different shapes alternating between them) writes,
    byte for byte, what the default one-pair-at-a-time run writes."""
    data = tmp_path / "data"
    shapes = [(32, 48, 8), (40, 64, 8), (32, 48, 8), (24, 40, 8), (40, 64, 8)]
    rels = ["s/p%d" % i for i in range(len(shapes))]
    for i, (rel, (H, W, D)) in enumerate(zip(rels, shapes)):
        _write_pair(str(data / rel), H, W, D, seed=60 + i)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s/im0.png\n" % (data / rel) for rel in rels))
    script = os.path.join(ROOT, "mc-cnn-python_amd", "src", "match.py")
    common = [sys.executable, script, "--list_file", str(lst), "--resume", os.path.join(GOLDEN_DIR, "mccnn_fast_weights.npz"),
              "--data_dir", str(data), "-s", "0", "-e", "4", "-t", "r"]
    for name, extra in (("one", []), ("two", ["--pairs_in_flight", "2"])):
        r = subprocess.run(common + ["--save_dir", str(tmp_path / name)] + extra, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=900)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
    for rel, (H, W, _) in zip(rels, shapes):
        for root, fn in (("submit_r", "disp0MCCNN.pfm"), ("submit_r_imgs", "disp0MCCNN.pgm")):
            a = (tmp_path / "one" / root / rel / fn).read_bytes()
            b = (tmp_path / "two" / root / rel / fn).read_bytes()
            assert a == b and len(a) > H * W, (rel, fn)
        assert float((tmp_path / "two" / "submit_r" / rel / "timeMCCNN.txt").read_text()) > 0

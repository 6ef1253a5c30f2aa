#!/usr/bin/env python3
"""Benchmark of the stereo-matching hot path (the timed region of the reference's match.py:129-179).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--exact] [--no-cpu-baseline]

A step = one stereo pair pushed through features -> cost volume -> CBCA x2 -> SGM (4 directions x 2 volumes)
-> CBCA x16 -> WTA -> LR check/interpolation -> sub-pixel -> median -> bilateral, with the standardised images and the
network weights already resident in HBM.  Metric: Mdisparities/s = H*W*D / seconds (BASELINE.json).  With N > 1
every rank matches its own pair (pairs are the independent unit; no data-path collective) and rank 0 reports the
aggregate over the slowest rank's time (weak scaling).  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))

CONFIGS = {  # BASELINE.json configs: (H, W, D)
    "cfg1": (256, 256, 64),
    "cfg2": (500, 750, 256),     # Middlebury-v3 half-res: the configuration the metric is quoted on
    "cfg3": (375, 1242, 192),    # KITTI-2015
    "cfg4": (1000, 1500, 400),   # Middlebury-v3 full-res
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--exact", action="store_true",
                    help="bit-exact variants (NumPy-order cost volume, reference-order CBCA) instead of the fast ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="100x150", help="HxW window of the workload timed on the CPU oracle")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="CPU-baseline worker processes (0 = half of the host threads, at most 16)")
    return ap.parse_args()


def cpu_baseline(H, W, D, sample, cores, wpath):
    """The oracle (a scalar C port of the reference's loops, oracle/mccnn_oracle.c) on bounded windows of the same
    workload: `cores` worker processes (oracle/cpu_window.py), one window each, started together; the rate is all
    their voxels over the slowest worker's time.  Reported next to the GPU number; never on the measured path."""
    import subprocess
    sh, sw = [int(x) for x in sample.split("x")]
    sh, sw = min(sh, H), min(max(sw, D + 2), W)
    worker = os.path.join(ROOT, "oracle", "cpu_window.py")
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, worker, str(sh), str(sw), str(D), str(1 + i), wpath],
                              stdout=subprocess.PIPE, env=env) for i in range(cores)]
    secs = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu_baseline worker failed")
        secs.append(float(out.decode().strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    slowest = max(secs)
    return {"value": round(cores * sh * sw * D / slowest / 1e6, 5), "unit": "Mdisparities/s", "cores": cores,
            "kind": "port",
            "sample": "%d workers x one %dx%d window, D=%d (%.1f%% of the %dx%d workload each), slowest %.1f s, "
                      "fastest %.1f s (%.3f Mdisp/s per core), %.0f s wall on %d host threads; the reference's own "
                      "interpreted loops measured 0.0068 Mdisp/s on one core (BASELINE.md)"
                      % (cores, sw, sh, D, 100.0 * sh * sw / (H * W), W, H, slowest, min(secs),
                         sh * sw * D / min(secs) / 1e6, wall, os.cpu_count() or 0)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))

    import numpy as np
    import torch

    import _hipabi as hip
    import distributed as mgpu
    import stereo_device as sd
    import synthetic
    import tf_checkpoint
    from model import NET

    hip.require_device()  # raises without a GPU or without the built library: no fallback
    # MCCNN_BENCH_SHARED_GPU=1 (tests only): ranks share the visible GPUs round-robin and rendezvous over gloo, so the
    # N > 1 control flow of this file can run on a one-GPU box; RCCL refuses two ranks on one device
    shared = os.environ.get("MCCNN_BENCH_SHARED_GPU") == "1"
    device_index = local_rank % torch.cuda.device_count() if shared else local_rank
    torch.cuda.set_device(device_index)
    if shared:
        mgpu.init("gloo")
    else:
        mgpu.init("nccl", torch.device("cuda", device_index))   # RCCL; only the barrier + timing all_gather use it

    H, W, D = CONFIGS[args.config]
    wpath = os.path.join(ROOT, "tests", "golden", "mccnn_fast_weights.npz")
    net = NET(None, input_patch_size=11, batch_size=1, device="cuda", seed=0)
    layers = None
    if os.path.isfile(wpath):
        layers = tf_checkpoint.load_fast_net_weights(wpath)
        net.set_layers(layers)
    else:
        layers = net.get_layers()
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=100 + rank)   # every rank owns a different pair
    dl = torch.from_numpy(L[:, :, 0]).cuda()
    dr = torch.from_numpy(R[:, :, 0]).cuda()
    matcher = sd.StereoMatcher(
        net, cv_mode=hip.MCCNN_CV_EXACT if args.exact else hip.MCCNN_CV_MFMA,
        cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER if args.exact else hip.MCCNN_CBCA_SEPARABLE)

    for _ in range(args.warmup):
        matcher.match(dl, dr, D)
    torch.cuda.synchronize()
    mgpu.barrier()
    timer = sd.StageTimer(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = matcher.match(dl, dr, D, timer=timer)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mgpu.barrier()

    elapsed_max = max(mgpu.gather_elapsed(elapsed))
    if rank != 0:
        mgpu.finalize()
        return

    voxels = H * W * D
    value = world * voxels * args.steps / elapsed_max / 1e6
    stages = {k: float(np.mean(v)) for k, v in timer.summary_ms().items()}          # mean ms per launch / stage
    counts = {k: len(v) // args.steps for k, v in timer.summary_ms().items()}
    per_step = {k: stages[k] * counts[k] for k in stages}                            # ms per step
    vol_bytes = 4.0 * voxels
    # algorithmic bytes per launch (SURVEY 8d / DESIGN.md): one read + one write of every voxel the launch owns
    algo = {
        "cbca_iter": 2 * vol_bytes,       # one iteration on one volume
        "sgm_pass": 2 * 2 * vol_bytes,    # one direction on BOTH volumes (one launch advances left + right)
        "sgm_first_pass": 2 * 2 * vol_bytes,
    }
    # HBM-side bytes per launch cannot be counted live (PMC needs rocprofv3): they come from the committed PMC passes of
    # the same kernels on the same workload (profiles/pmc_traffic.json, see profiles/r01_pmc_summary.md), else null
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(tpath) and not args.exact:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("workload") == args.config:
            traffic = {k: int(v["traffic_bytes"]) for k, v in tj.items() if isinstance(v, dict)}
    rooflines = {}
    for k, b in algo.items():
        if k in stages:
            ach = b / (stages[k] * 1e-3) / 1e9
            rooflines[k] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(k),
                            "launches_per_step": counts[k], "avg_launch_ms": round(stages[k], 4),
                            "algorithmic_bytes_per_launch": int(b)}
    dominant = max(rooflines, key=lambda k: per_step[k]) if rooflines else None
    # SGM as a stage (what north_star's >= 50 % target is quoted on): 4 passes + the two layout changes
    sgm_stage_ms = (per_step.get("sgm_pass", 0.0) + per_step.get("sgm_first_pass", 0.0) +
                    per_step.get("dhw_to_hwd", 0.0) + per_step.get("hwd_to_dhw", 0.0))
    sgm_kern_ms = per_step.get("sgm_pass", 0.0) + per_step.get("sgm_first_pass", 0.0)
    result = {
        "metric": "Mdisparities/s (HxWxD / s) end-to-end match.py timed region",
        "value": round(value, 2), "unit": "Mdisparities/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d synthetic stereo pair, D=%d, one pair per GPU" % (args.config, W, H, D),
                   "variant": "exact" if args.exact else "fast (MFMA cost volume, separable CBCA)",
                   "weights": "converted reference checkpoint" if os.path.isfile(wpath) else "random init"},
        "roofline": dict(rooflines[dominant], kernel=dominant) if dominant else None,
        "rooflines": rooflines,
        "sgm_stage": {"ms": round(sgm_stage_ms, 4), "algorithmic_bytes": int(4 * 2 * 2 * vol_bytes),
                      "achieved_GBs": round(4 * 2 * 2 * vol_bytes / (sgm_stage_ms * 1e-3) / 1e9, 1) if sgm_stage_ms else None,
                      "frac_of_hbm_peak": round(4 * 2 * 2 * vol_bytes / (sgm_stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                      if sgm_stage_ms else None},
        # the SGM kernels alone: four directions, the first one carrying the DHW->HWD layout change; the separate
        # HWD->DHW launches excluded
        "sgm_kernels": {"ms": round(sgm_kern_ms, 4),
                        "frac_of_hbm_peak": round(4 * 2 * 2 * vol_bytes / (sgm_kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                        if sgm_kern_ms else None},
        "stage_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
    }
    if not args.no_cpu_baseline and world == 1:
        cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(16, (os.cpu_count() or 2) // 2))
        if not os.path.isfile(wpath):      # random-init run: hand the workers the same weights through a file
            import tempfile
            wpath = os.path.join(tempfile.mkdtemp(), "weights.npz")
            np.savez(wpath, **{"conv%d/%s" % (k + 1, n): a for k, (w, b) in enumerate(layers)
                               for n, a in (("weights", w), ("biases", b))})
        result["cpu_baseline"] = cpu_baseline(H, W, D, args.cpu_sample, cores, wpath)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    mgpu.finalize()


if __name__ == "__main__":
    main()

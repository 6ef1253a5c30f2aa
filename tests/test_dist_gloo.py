"""CPU, world_size 2 over gloo: the N > 1 structure of bench.py / match.py - pairs sharded across ranks with no data-path
collective, a barrier in front of the timed region, one all_gather of elapsed seconds behind it, whole-job rate over the
slowest rank.  On the GPU node the same code runs over RCCL (backend "nccl")."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mc-cnn-python_amd", "src")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, SRC)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import distributed as mgpu
    r, lr, w = mgpu.init("gloo")
    assert (r, lr, w) == (rank, rank, world)
    mine = mgpu.shard_indices(0, 6, 7, r, w)          # 7 pairs over 2 ranks
    mgpu.barrier()
    elapsed = 0.5 + 0.25 * r                           # a stand-in for the measured wall time of this rank's pairs
    units = len(mine) * 96.0                           # Mvoxels matched by this rank
    all_t = mgpu.gather_elapsed(elapsed)
    all_units = [len(mgpu.shard_indices(0, 6, 7, k, w)) * 96.0 for k in range(w)]
    mgpu.barrier()
    q.put((r, mine, all_t, mgpu.aggregate_throughput(all_units, all_t), units))
    mgpu.finalize()


def test_two_ranks_shard_pairs_and_gather_timings():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, t0, rate0, u0), (r1, mine1, t1, rate1, u1) = results
    assert mine0 == [0, 2, 4, 6] and mine1 == [1, 3, 5]              # disjoint, complete
    assert t0 == t1 == [0.5, 0.75]                                    # every rank sees every rank's time
    assert rate0 == rate1 == (u0 + u1) / 0.75                         # aggregate over the slowest rank


WINDOWS = [(0, 14), (3, 11), (5, 5), (-2, 30), (10, 3), (7, 14)]     # -s / -e as given on the command line, 15 pairs


def _worker8(rank, world, port, q):
    sys.path.insert(0, SRC)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import distributed as mgpu
    mgpu.init("gloo")
    mine = [mgpu.shard_indices(s, e, 15, rank, world) for s, e in WINDOWS]
    mgpu.barrier()
    elapsed = 1.0 + 0.125 * ((rank * 5) % world)       # stand-in wall times; rank 3 is the slowest
    everyone = mgpu.gather_objects({"rank": rank, "device": rank, "mine": mine})
    all_t = mgpu.gather_elapsed(elapsed)
    mgpu.barrier()
    q.put((rank, mine, everyone, all_t))
    mgpu.finalize()


def test_eight_ranks_shard_ragged_windows():
    """The shape of the first real 8-GPU run (cfg5), on CPU over gloo: 8 ranks, a 15-pair list, ragged -s/-e windows
    (match.py:85-91 clips them to the list): every window's pairs are dealt disjointly and completely, at most one pair
    of imbalance, every rank sees every rank's device record and time, the job rate is over the slowest rank."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, SRC)
    import distributed as mgpu
    for wi, (s, e) in enumerate(WINDOWS):
        want = list(range(max(s, 0), min(e, 14) + 1))
        dealt = [results[r][1][wi] for r in range(world)]
        assert sorted(i for d in dealt for i in d) == want, (s, e)               # disjoint + complete
        sizes = [len(d) for d in dealt]
        assert max(sizes) - min(sizes) <= 1
    times = [1.0 + 0.125 * ((r * 5) % world) for r in range(world)]
    for r, _, everyone, all_t in results:
        assert [x["rank"] for x in everyone] == list(range(world)) and everyone[5]["mine"] == results[5][1]
        assert all_t == times
    units = [len(results[r][1][0]) * 96.0 for r in range(world)]
    assert mgpu.aggregate_throughput(units, times) == sum(units) / max(times) == 15 * 96.0 / 1.875


def test_single_process_needs_no_group():
    sys.path.insert(0, SRC)
    import distributed as mgpu
    assert mgpu.gather_elapsed(1.5) == [1.5]
    mgpu.barrier()
    assert mgpu.aggregate_throughput([10.0], [2.0]) == 5.0

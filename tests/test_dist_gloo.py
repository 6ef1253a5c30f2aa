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


def test_single_process_needs_no_group():
    sys.path.insert(0, SRC)
    import distributed as mgpu
    assert mgpu.gather_elapsed(1.5) == [1.5]
    mgpu.barrier()
    assert mgpu.aggregate_throughput([10.0], [2.0]) == 5.0

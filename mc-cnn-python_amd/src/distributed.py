"""One process per GPU, one stereo pair per process: the only multi-GPU structure the hot path has.

The reference parallelises by hand - N copies of match.py with different `-g` and `-s/-e` windows
(/root/reference/src/match.py:17-18, 26-28, 85-91).  Pairs never exchange volume data, so there is no data-path
collective: ranks shard the list of pairs, and the only communication is a barrier in front of a timed region and an
all_gather of one float64 per rank (elapsed seconds) behind it.  Backend "nccl" is RCCL on ROCm (xGMI between the
8 GPUs of a node); "gloo" runs the same code on CPU for the tests.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    """(rank, local_rank, world_size) from the torchrun / torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None, always=False):
    """Joins the process group when WORLD_SIZE > 1 - or, with `always`, also as a group of one, which runs the same
    RCCL initialisation, barrier and all_gather a multi-GPU job does (rendezvous on 127.0.0.1 unless MASTER_ADDR is
    set).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank_world()
    if (world > 1 or always) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_indices(start, end, n_items, rank, world):
    """Pairs of the inclusive window [start, end] (clipped to the list, match.py:85-91) owned by `rank`:
    round-robin, so every GPU gets the same number of pairs +-1 whatever the window."""
    first = max(int(start), 0)
    last = min(int(end), n_items - 1)
    return [i for i in range(first, last + 1) if (i - first) % world == rank]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_elapsed(elapsed_seconds, device=None):
    """all_gather of one float64 per rank -> list of every rank's elapsed seconds (on every rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(elapsed_seconds)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" \
            else torch.device("cpu")
    t = torch.tensor([float(elapsed_seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def gather_objects(obj):
    """all_gather of one small picklable object per rank (which device every rank drives, for the bench line)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def aggregate_throughput(units_per_rank, elapsed_per_rank):
    """Whole-job rate: everything all ranks processed over the slowest rank's time."""
    return float(sum(units_per_rank)) / max(elapsed_per_rank)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()

"""Multi-GPU plumbing for the self-play path: one process per GPU, games sharded statically, NO collective on the data
path (games share nothing — synthesis/src/alpha_zero.rs:181-209). torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only to line the ranks up around a timed region and to combine per-rank scalars."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, local_rank=0):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def step_game_range(step, rank, world, games_per_step):
    """Weak-scaling shard: in step `step`, rank `rank` plays the contiguous block of `games_per_step` global game
    indices [(step*world + rank)*games_per_step, ...). Every global index is played by exactly one rank."""
    first = (step * world + rank) * games_per_step
    return first, games_per_step


def reduce_scalars(dist, device, elapsed, counts):
    """MAX over ranks of the elapsed time, SUM over ranks of the integer counters. `dist` may be None (single rank)."""
    if dist is None:
        return elapsed, list(counts)
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor(list(counts), dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(x) for x in c.tolist()]


def barrier(dist):
    if dist is not None:
        dist.barrier()

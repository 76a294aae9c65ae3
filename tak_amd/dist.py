"""Multi-GPU plumbing for sharded self-play: one process per GPU, games sharded by slot, no data-path
collective.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU in tests)
is used only for the barrier and for reducing the timing / counters that rank 0 reports."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def slot_base(rank, games_per_rank):
    """Global index of a rank's first game: RNG streams are keyed by the global slot, so the union of the
    shards' games is the same set of games whatever the number of ranks."""
    return rank * games_per_rank


def init(backend, rank, world, device=None):
    if world <= 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def reduce_time_and_count(dist, dt_local, count_local, device="cpu"):
    """(max over ranks of the elapsed time, sum over ranks of the work count)"""
    if dist is None:
        return dt_local, count_local
    import torch

    t = torch.tensor([dt_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(count_local)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c.item())

"""Frame sharding across the GPUs of one node.

The hot path has no exchange step: every frame's bev_pool, voxelization and sparse encoder depend on that frame
only (batch index is the slowest-varying term of every output address — bev_pool_cuda.cu:34, the per-sample loop
of bevfusion.py:173, the per-batch spconv grid of spconv_ops.h:60-62).  So the batch is split across ranks, one
process per GPU, and no data-path collective exists; ranks meet only to agree on timing (bench.py) or, in
training, in the dense model's gradient all-reduce (torch.distributed "nccl" == RCCL over xGMI on ROCm).
"""
import os

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def frames_for_rank(n_frames, rank, world):
    """Contiguous, balanced split of frame ids [0, n_frames): the first n_frames % world ranks get one more."""
    base, extra = divmod(int(n_frames), int(world))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (identity when torch.distributed is not initialised)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

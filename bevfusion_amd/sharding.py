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


# ---- per-rank CPU placement ----------------------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs `local_cpulist`)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpus_for_rank(local_rank, local_world, available, near_gpu=None):
    """The CPUs rank `local_rank` of `local_world` ranks on this node should run on: its share of the CPUs that sit next to
    its GPU (`near_gpu[r]` = CPU list of rank r's GPU from sysfs, ranks whose GPUs share a NUMA node split that node's CPUs
    evenly, in rank order), or — without topology information — an even contiguous split of `available`.  Never empty."""
    available = sorted(available)
    if local_world <= 1 or not available:
        return available
    if near_gpu and all(near_gpu.get(r) for r in range(local_world)):
        mine = sorted(set(near_gpu[local_rank]) & set(available))
        sharers = [r for r in range(local_world) if sorted(set(near_gpu[r]) & set(available)) == mine]
        if mine and len(mine) >= len(sharers):
            k = sharers.index(local_rank)
            per = len(mine) // len(sharers)
            return mine[k * per:(k + 1) * per]
    per = max(1, len(available) // local_world)
    lo = min(local_rank * per, len(available) - per)
    return available[lo:lo + per]


def gpu_local_cpus(device_index):
    """CPU list of the NUMA node the GPU hangs off (sysfs `local_cpulist` of its PCI function), None when unknown."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            cpus = parse_cpulist(fh.read())
        return cpus or None
    except Exception:
        return None


def bind_rank_to_cpus(local_rank, local_world):
    """Pin this process (every thread it creates later inherits the mask) to its share of the node's CPUs, next to its GPU when
    sysfs says where that is.  One process per GPU means 8 ranks synthesising inputs, launching and timing on one host: without
    a mask they migrate across sockets and share cores.  Returns a short description for the bench JSON; a single rank keeps the
    whole machine."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        available = sorted(os.sched_getaffinity(0))
        near = {r: gpu_local_cpus(r) for r in range(local_world)} if torch.cuda.is_available() else None
        mine = cpus_for_rank(local_rank, local_world, available, near)
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(mine))))
        return dict(cpus=len(mine), first=mine[0], last=mine[-1], numa_aware=bool(near and near.get(local_rank)))
    except OSError:
        return None

"""CPU, world_size 2 over gloo: the N>1 path of the bench — frame sharding without a data-path collective,
max-over-ranks timing, whole-job throughput."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bevfusion_amd.sharding import frames_for_rank


def test_frames_for_rank_partitions_exactly():
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in frames_for_rank(n, r, world)]
            assert got == list(range(n))
            sizes = [len(frames_for_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bevfusion_amd.sharding import barrier, max_over_ranks, rank_world, sum_over_ranks

    assert rank_world() == (rank, world)
    frames = list(frames_for_rank(9, rank, world))
    # each rank "processes" its own frames (weak scaling: per-rank work independent of the others)
    local = sum(f * f for f in frames)
    barrier()
    elapsed = max_over_ranks(1.0 + rank)          # slowest rank defines the step time
    total_frames = sum_over_ranks(len(frames))
    checksum = sum_over_ranks(local)
    q.put((rank, frames, elapsed, total_frames, checksum))
    dist.destroy_process_group()


def test_two_ranks_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3, 4] and res[1][1] == [5, 6, 7, 8]
    for r in res:
        assert r[2] == 2.0 and r[3] == 9.0 and r[4] == float(sum(i * i for i in range(9)))

"""CPU: the N>1 control flow of bench.py end to end — `python bench.py --gpus 2 --dry-run` with no launcher environment must
spawn two ranks itself (torch.distributed.run on 127.0.0.1), shard the frames, time with barriers + max over ranks and print
ONE JSON line with n_gpus = 2 (`--gpus` used to be ignored).  The GPU work is stubbed, the backend is gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks_weak_scaling():
    res = run_bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "3")
    assert res["n_gpus"] == 2 and res["dry_run"] and res["scaling"] == "weak"
    cfg = res["config"]
    assert cfg["world_size"] == 2 and cfg["frames_per_step"] == 6
    assert sorted(tuple(r["frames"]) for r in cfg["per_rank"]) == [(0, 1, 2), (3, 4, 5)]    # disjoint frame ids per rank
    assert res["ms_per_step"] >= max(r["ms_per_step"] for r in cfg["per_rank"]) * 0.5
    assert abs(res["value"] - 6 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]


def test_global_batch_is_split_with_frames_for_rank():
    res = run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--global-batch", "5")
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert sorted(tuple(r["frames"]) for r in res["config"]["per_rank"]) == [(0, 1, 2), (3, 4)]
    one = run_bench("--dry-run", "--steps", "2", "--warmup", "0", "--global-batch", "5")
    assert one["n_gpus"] == 1 and one["config"]["per_rank"][0]["frames"] == [0, 1, 2, 3, 4]
    # every frame is processed exactly once whatever the rank count: same checksum
    assert abs(one["config"]["checksum"] - res["config"]["checksum"]) < 1e-6 * abs(one["config"]["checksum"])


def test_train_step_dry_run_goes_through_ddp():
    """`--mode train-step --gpus 2 --dry-run`: the gradient path of the N > 1 training step — per-rank
    inputs differ, gradients pass through DistributedDataParallel (gloo here, RCCL on the GPUs) and come out identical on
    both ranks and equal to the mean of the per-rank gradients; the updated weights stay in lock step."""
    res = run_bench("--mode", "train-step", "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "2")
    cfg = res["config"]
    assert res["n_gpus"] == 2 and res["dry_run"] and cfg["ddp"] and cfg["frames_per_step"] == 4
    assert sorted(tuple(r["frames"]) for r in cfg["per_rank"]) == [(0, 1), (2, 3)]
    assert cfg["inputs_differ_across_ranks"] and cfg["gradients_agree_across_ranks"] and cfg["weights_agree_across_ranks"]
    assert cfg["max_abs_grad_minus_mean_of_per_rank_grads"] <= 1e-6
    one = run_bench("--mode", "train-step", "--dry-run", "--steps", "2", "--warmup", "1", "--batch", "2")
    assert one["n_gpus"] == 1 and not one["config"]["ddp"]


def test_configs4_shape_eight_ranks_train_step_dry_run():
    """BASELINE configs[4] as the driver will launch it on an 8-GPU node: 32 frames split over 8 ranks (4 per GPU), training step
    through DistributedDataParallel — here gloo and a host stub.  Every rank gets 4 disjoint frames, the all-reduce sees 8 ranks,
    each rank is pinned to its own share of the CPUs (disjoint masks), gradients and weights agree everywhere."""
    res = run_bench("--mode", "train-step", "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--global-batch", "32")
    cfg = res["config"]
    assert res["n_gpus"] == 8 and res["scaling"] == "strong" and cfg["ddp"] and cfg["rccl_ranks"] == 8
    assert cfg["frames_per_step"] == 32
    assert sorted(tuple(r["frames"]) for r in cfg["per_rank"]) == [tuple(range(4 * r, 4 * r + 4)) for r in range(8)]
    assert cfg["inputs_differ_across_ranks"] and cfg["gradients_agree_across_ranks"] and cfg["weights_agree_across_ranks"]
    assert cfg["max_abs_grad_minus_mean_of_per_rank_grads"] <= 1e-6
    binds = [r["cpu_binding"] for r in cfg["per_rank"]]
    if all(b is not None for b in binds) and len(os.sched_getaffinity(0)) >= 8:
        spans = sorted((b["first"], b["last"]) for b in binds)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans          # disjoint CPU ranges per rank


def test_cpu_share_of_a_rank():
    from bevfusion_amd.sharding import cpus_for_rank, parse_cpulist

    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    avail = list(range(128))
    # two sockets, GPUs 0-3 on the first: each rank gets a quarter of ITS socket's CPUs, no overlap, nothing from the other socket
    sock = {0: list(range(0, 32)) + list(range(64, 96)), 1: list(range(32, 64)) + list(range(96, 128))}
    near = {r: sock[r // 4] for r in range(8)}
    shares = [cpus_for_rank(r, 8, avail, near) for r in range(8)]
    assert all(len(s) == 16 for s in shares) and len(set(sum(shares, []))) == 128
    assert all(set(shares[r]) <= set(sock[r // 4]) for r in range(8))
    # no topology: an even contiguous split; more ranks than CPUs still yields a non-empty share
    assert [cpus_for_rank(r, 2, list(range(8))) for r in range(2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert cpus_for_rank(5, 8, [0, 1, 2]) == [2] and cpus_for_rank(0, 1, [3, 4]) == [3, 4]

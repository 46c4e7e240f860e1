"""GPU: the N > 1 control flow of the REAL benchmark step on a one-GPU box — `python bench.py --gpus 2` with
BEVAMD_BENCH_SHARED_GPU=1 puts both ranks on device 0 and runs the collectives over gloo (RCCL refuses two ranks on one device; the
RCCL path itself is test_gpu_ddp.py).  Everything else is what the driver's 8-GPU launch runs: bench.py re-spawns itself under
torch.distributed.run on 127.0.0.1, every rank gets its own frame ids, builds its own plans and HIP graphs (the pipelined schedule
from 2 frames per step), the timed region is bracketed by barriers, rank 0 prints ONE line with the max over ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BEVAMD_BENCH_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *flags]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0 and "ChildFailedError" in r.stderr:
        # a rank aborted inside the gloo rendezvous / teardown (rare, never reproduced in isolation: tests/test_bench_launch.py):
        # one more try — a deterministic failure fails again
        print("bench.py launch retried after:", r.stderr[-600:], file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("batch", [2, 4])   # both on the software pipeline across steps (two buffer sets per rank)
def test_two_ranks_weak_scaling_on_one_device(batch):
    res = run_bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", str(batch), "--no-cpu-baseline", "--no-extras")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 4 and res["warmup"] == 2
    cfg = res["config"]
    assert cfg["frames_per_step"] == 2 * batch and cfg["frames_per_step_per_gpu"] == batch and cfg["rccl_ranks"] == 2
    per_rank = cfg["per_rank_ms_per_step"]
    per_rank = list(per_rank.values()) if isinstance(per_rank, dict) else list(per_rank)
    assert len(per_rank) == 2 and all(float(t) > 0 for t in per_rank)
    assert res["ms_per_step"] >= max(float(t) for t in per_rank) * 0.999                       # the slowest rank defines the step
    assert abs(res["value"] - 2 * batch * 1e3 / res["ms_per_step"]) <= 1e-3 * res["value"]
    assert res["roofline"]["frac"] > 0 and res["cpu_baseline"] is None


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_two_ranks_strong_scaling_splits_the_global_batch():
    res = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--global-batch", "5", "--no-cpu-baseline", "--no-extras")
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["config"]["frames_per_step"] == 5 and res["config"]["rccl_ranks"] == 2


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_two_ranks_train_step_through_ddp_on_one_device():
    """BASELINE configs[4] in small: the --amp training step on two ranks (2 frames each), gradients through DistributedDataParallel
    (gloo here, RCCL on a multi-GPU node), one line from rank 0."""
    res = run_bench("--mode", "train-step", "--amp", "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2", "--no-cpu-baseline")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    cfg = res["config"]
    assert cfg["frames_per_step"] == 4 and cfg["frames_per_step_per_gpu"] == 2 and cfg["encoder_path"] == "fused-train"
    assert "DistributedDataParallel" in cfg["gradient_allreduce"] and "world 2" in cfg["gradient_allreduce"]
    assert res["value"] > 0 and abs(res["value"] - 4e3 / res["ms_per_step"]) <= 1e-3 * res["value"]

"""bench.py --gpus N started directly (no torchrun): the parent must spawn the N ranks itself before anything touches the
GPU, relay rank 0's ONE JSON line and fail if any rank fails (the driver's N = 1 invocation form
is `python bench.py --gpus 1 …`, and an 8-GPU node would be driven the same way or through torch.distributed.run).
On this machine there is no GPU, so what runs is the launcher plumbing: rendezvous on a free port, barrier, the
max-over-ranks / sum-over-ranks reductions (--rehearse-launch: no engine, no measurement, `value` null)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_direct_invocation_spawns_the_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-launch"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["rehearsal"] is True and out["value"] is None
    assert out["max_over_ranks"] == 2.0 and out["sum_over_ranks"] == 3  # ranks contributed (1 + rank): max 2, sum 3


def test_torchrun_form_still_works():
    from bench import free_port

    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), BENCH, "--gpus", "2", "--rehearse-launch"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_failing_rank_fails_the_launch():
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU: the ranks must fail with 'no GPU visible'")
    # without --rehearse-launch the ranks need a GPU: each exits 2 (the engine has no CPU fallback) and so must the launcher
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2
    assert p.stdout.decode().strip() == ""
    assert "no CPU fallback" in p.stderr.decode()


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-launch"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr.decode()

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
    assert out["ranks_reporting"] == 2


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


def _launch(*extra, timeout=300):
    p = subprocess.run([sys.executable, BENCH, "--rehearse-launch", *extra], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    return p.returncode, lines, p.stderr.decode()


def test_eight_ranks_at_the_real_fan_out():
    """the 8-GPU node's launch shape on the CPU: free port, eight child processes, OMP split, ONE line relayed from rank 0"""
    rc, lines, err = _launch("--gpus", "8")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["max_over_ranks"] == 8.0 and out["sum_over_ranks"] == 36
    # the line explains itself on the day the scaling runs happen: who contributed, and the per-GPU figure beside the whole-job one
    assert out["ranks_reporting"] == 8 and "per_gpu_value" in out
    # round 6 — who sat where, every rank's own time, the reduction preflight, the A/B switches in effect: gathered over the 8 ranks
    dev = out["config"]["devices"]
    assert [d["rank"] for d in dev] == list(range(8)) and len({d["pid"] for d in dev}) == 8
    assert all(set(d) >= {"rank", "hip_device", "pci_bus_id", "name", "cu_count", "host", "pid"} for d in dev)
    assert out["config"]["devices_distinct"] is True and out["config"]["switches_set"] == []
    by = out["ms_per_step_by_rank"]
    assert by["all"] == [1000.0 * (1 + r) for r in range(8)] and by["slowest_rank"] == 7 and by["fastest_rank"] == 0 and by["max"] == 8000.0
    pf = out["preflight_ms_by_rank"]
    assert len(pf["all"]) == 8 and pf["max"] >= pf["min"] >= 0.0


def test_a_switch_in_one_ranks_environment_shows_in_the_line():
    env = dict(_clean_env(), TG_NO_HALO_TOWER="1", TG_NO_FC_GATHER="0")
    p = subprocess.run([sys.executable, BENCH, "--rehearse-launch", "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.strip()][0])
    assert out["config"]["switches_set"] == ["TG_NO_HALO_TOWER=1"]  # `=0` is off and is not listed


def test_two_ranks_on_one_card_are_refused_under_rccl():
    from tak_amd import dist as tdist

    def rec(rank, bus, host="box"):
        return {"rank": rank, "hip_device": rank, "pci_bus_id": bus, "name": "AMD Instinct MI355X", "cu_count": 256, "host": host, "pid": 100 + rank}

    eight = [rec(r, f"0000:{0x05 + 0x10 * r:02x}:00.0") for r in range(8)]
    assert tdist.check_devices(eight, "nccl") is True
    shared = eight[:6] + [rec(6, eight[2]["pci_bus_id"]), eight[7]]
    with pytest.raises(RuntimeError, match=r"ranks share a GPU.*\[2, 6\]"):
        tdist.check_devices(shared, "nccl")
    assert tdist.check_devices(shared, "gloo") is False  # several ranks on one card on purpose: reported, not refused
    # the same bus id on two HOSTS is two cards
    assert tdist.check_devices([rec(0, "0000:05:00.0", "a"), rec(1, "0000:05:00.0", "b")], "nccl") is True
    assert tdist.per_rank_times(None, 0.25) == {"all": [250.0], "min": 250.0, "max": 250.0, "fastest_rank": 0, "slowest_rank": 0}


@pytest.mark.parametrize("world", [2, 8])
def test_a_hung_collective_ends_non_zero_with_the_headline_printed(world):
    """config C5's watchdog, acted out without a GPU: one rank never joins the collective → after --train-timeout rank 0 prints
    the line it has, marked train_c5_failed, and the launch ends with EXIT_C5_FAILED instead of hanging or looking green"""
    import time

    from bench import EXIT_C5_FAILED

    t0 = time.time()
    rc, lines, err = _launch("--gpus", str(world), "--train-timeout", "8", "--rehearse-hang-rank", str(world - 1), timeout=240)
    assert rc == EXIT_C5_FAILED, (rc, err[-2000:])
    assert time.time() - t0 < 120
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["train_c5_failed"] is True and "timed out" in out["extra"]["train_c5"]["error"] and out["n_gpus"] == world
    assert out["extra"]["train_c5"]["stage"] == "rehearsed collective"  # the line names the stage the phase hung in
    assert "did not finish within 8 s" in err


def test_a_rank_that_fails_before_the_collective_releases_the_others():
    """one rank raises in its local stage: every other rank learns it from the agreement all-reduce and leaves too — nobody waits
    in the collective for the watchdog (which is set far away here)"""
    import time

    from bench import EXIT_C5_FAILED

    t0 = time.time()
    rc, lines, err = _launch("--gpus", "4", "--train-timeout", "600", "--rehearse-fail-rank", "2", timeout=240)
    assert rc == EXIT_C5_FAILED, (rc, err[-2000:])
    assert time.time() - t0 < 120, "the ranks waited for the watchdog"
    out = json.loads(lines[0])
    assert out["train_c5_failed"] is True and "rehearsed local failure" not in out["extra"]["train_c5"]["error"]  # rank 0 saw "another rank failed"
    assert "another rank failed" in out["extra"]["train_c5"]["error"]
    assert "rehearsed local failure" in err


@pytest.mark.parametrize("world,bad", [(2, 1), (8, 5)])
def test_a_rank_that_fails_inside_the_collective_phase_does_not_cost_the_headline(world, bad):
    """a rank != 0 raises behind the last agreement (inside tg_train, between two reductions) while rank 0 waits in
    the collective.  Its non-zero exit makes the launcher end every rank — so it posts the failure on the rendezvous store and
    leaves only once rank 0 (whose polling thread sees the post) has printed the marked headline.  Far inside the watchdog."""
    import time

    from bench import EXIT_C5_FAILED

    t0 = time.time()
    rc, lines, err = _launch("--gpus", str(world), "--train-timeout", "600", "--rehearse-fail-in-collective-rank", str(bad), timeout=240)
    assert rc == EXIT_C5_FAILED, (rc, err[-2000:])
    assert time.time() - t0 < 120, "waited for the watchdog"
    assert len(lines) == 1, (lines, err[-2000:])
    out = json.loads(lines[0])
    assert out["train_c5_failed"] is True and out["n_gpus"] == world
    assert f"rank {bad}" in out["extra"]["train_c5"]["error"] and "rehearsed failure between two reductions" in out["extra"]["train_c5"]["error"]

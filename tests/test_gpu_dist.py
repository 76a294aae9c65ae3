"""The multi-GPU path driven through the PRODUCT (libtakgpu), on one card: SURVEY.md §8(e) partitions self-play by game
(rank r owns global slots [r·G, (r+1)·G), TgSearchConfig.slot_base, weights replicated, no data-path collective) and
trains data-parallel with one gradient all-reduce per optimiser step.  Two OS processes — two ranks, both on device 0,
torch.distributed/gloo as the host transport (RCCL refuses two ranks on one device; with one GPU per rank the same
code path runs tg_train_comm_init instead of the host hook) — must reproduce a single process that owns all the games."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import torch_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, BLOCKS, FILTERS, G, PLIES = 5, 2, 32, 6, 110
KW = dict(rollouts=16, noise_plies=8, exploit_plies=6, total_games=0, seed=13, arena_nodes=1 << 14, max_examples=1 << 13)


def _keyed(h, s, m, v):
    return {(int(h["game_id"][i]), s[i].tobytes()): (m[i].tobytes(), v[i].tobytes(), float(h["result"][i])) for i in range(len(h))}


def test_slot_base_shard_equals_the_slots_of_a_full_run(orc):
    # one process: an engine that owns slots G…2G-1 (slot_base = G) against one that owns 0…2G-1, hash evaluator and network
    import tak_amd

    net = torch_ref.make_net(N, BLOCKS, FILTERS, "fc5", seed=2)
    for evaluator in (tak_amd.EVAL_HASH, tak_amd.EVAL_RESNET):
        runs = []
        for games, base in ((2 * G, 0), (G, G), (G, 0)):
            e = tak_amd.Engine(N, res_blocks=BLOCKS, filters=FILTERS, evaluator=evaluator, max_batch=64)
            if evaluator == tak_amd.EVAL_RESNET:
                e.load_state_dict(torch_ref.abi_tensors(net))
            e.selfplay_create(games, slot_base=base, **KW)
            e.selfplay_step(PLIES)
            runs.append((e.selfplay_stats(), e.selfplay_drain(1 << 13), e.search_states()))
            e.close()
        (full_st, full_ex, full_roots), (hi_st, hi_ex, hi_roots), (lo_st, lo_ex, lo_roots) = runs
        assert full_st["examples"] > 0 and full_st["examples"] == hi_st["examples"] + lo_st["examples"]
        assert full_st["expansions"] == hi_st["expansions"] + lo_st["expansions"]
        slot = full_ex[0]["game_id"] & 0xFFFFF
        for part, keep in ((hi_ex, slot >= G), (lo_ex, slot < G)):
            # the shard's examples ARE the full run's examples of its slots, in the same order
            assert np.array_equal(part[0], full_ex[0][keep])
            for a, b in zip(part[1:], full_ex[1:]):
                assert np.array_equal(a, b[keep])
        assert np.array_equal(full_roots[G:], hi_roots) and np.array_equal(full_roots[:G], lo_roots)
        if evaluator == tak_amd.EVAL_HASH:  # and the oracle agrees on the shard
            okw = {k: v for k, v in KW.items() if k not in ("arena_nodes", "max_examples")}
            sp = orc.SelfPlay(N, G, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH, slot_base=G, **okw)
            sp.step(PLIES)
            oh, os_, om, ov = sp.drain(1 << 13)
            assert np.array_equal(oh, hi_ex[0]) and np.array_equal(os_, hi_ex[1]) and np.array_equal(ov, hi_ex[3])


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import tak_amd
import torch_ref
from tak_amd import dist as tdist
rank, world, _ = tdist.env_rank()
dist = tdist.init("gloo", rank, world)
N, BLOCKS, FILTERS, G, PLIES = {cfg!r}
KW = {kw!r}
net = torch_ref.make_net(N, BLOCKS, FILTERS, "fc5", seed=2)
e = tak_amd.Engine(N, res_blocks=BLOCKS, filters=FILTERS, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
e.load_state_dict(torch_ref.abi_tensors(net))
# ---- self-play shard: no collective on the data path ----
e.selfplay_create(G, slot_base=tdist.slot_base(rank, G), **KW)
dist.barrier()
e.selfplay_step(PLIES)
st = e.selfplay_stats()
hdr, states, moves, visits = e.selfplay_drain(1 << 13)
_, total = tdist.reduce_time_and_count(dist, 1.0, st["expansions"])
# ---- data-parallel training on the rank's own examples: Σ grads over ranks ÷ world → Adam, every `chunks_in_step` chunks ----
count = 16
k = (min(len(hdr), 64) // count) * count
k = int(tdist.reduce_min(dist, k))          # the same number of whole chunks on every rank
e.train_create(learning_rate=1e-3, weight_decay=1e-2, chunk_size=count, chunks_in_step=2)
e.train_set_allreduce(tdist.host_allreduce_hook(dist), world)
lp, lz, steps = e.train(states[:k], hdr["n_moves"][:k], moves[:k], visits[:k], hdr["result"][:k], seed=5)
shapes = {{torch_ref.abi_name(n_): tuple(v.shape) for n_, v in net.named_parameters()}}
params = {{n_: e.train_get_tensor(n_, s) for n_, s in shapes.items()}}
e.train_commit()
bn = e.train_get_tensor("bn0.running_mean", (FILTERS,))
pe = e.policy_eval(states[:4])
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), hdr=hdr, states=states, moves=moves, visits=visits, total=total,
         local=st["expansions"], k=k, steps=steps, lp=lp, bn=bn, pol=pe[0], ev=pe[1], **{{"p_" + n_: v for n_, v in params.items()}})
e.close()
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_selfplay_and_training(orc):
    import tak_amd

    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(WORKER.format(root=ROOT, out=tmp, cfg=(N, BLOCKS, FILTERS, G, PLIES), kw=KW))
        from bench import free_port

        # a free port per run: a fixed one collides with a concurrent pytest or with the leftovers of a killed run
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0")) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        shards = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(2)]
    # self-play: the union of the two ranks' examples = a single engine that owns all 2G slots
    net = torch_ref.make_net(N, BLOCKS, FILTERS, "fc5", seed=2)
    e = tak_amd.Engine(N, res_blocks=BLOCKS, filters=FILTERS, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    e.load_state_dict(torch_ref.abi_tensors(net))
    e.selfplay_create(2 * G, slot_base=0, **KW)
    e.selfplay_step(PLIES)
    single = _keyed(*e.selfplay_drain(1 << 13))
    assert e.selfplay_stats()["expansions"] == int(shards[0]["total"]) == int(shards[0]["local"]) + int(shards[1]["local"])
    e.close()
    union = {}
    for sh in shards:
        assert len(sh["hdr"]) > 0
        union.update(_keyed(sh["hdr"], sh["states"], sh["moves"], sh["visits"]))
    assert union == single
    # training: both ranks took the same number of steps and hold bit-identical parameters, BatchNorm statistics and outputs
    assert int(shards[0]["steps"]) == int(shards[1]["steps"]) >= 1 and int(shards[0]["k"]) == int(shards[1]["k"])
    for key in shards[0]:
        if key.startswith("p_") or key in ("bn",):
            assert np.array_equal(shards[0][key], shards[1][key]), key
    assert float(shards[0]["lp"]) != float(shards[1]["lp"])  # … although they trained on different examples
    changed = max(np.abs(shards[0]["p_" + torch_ref.abi_name(k)] - v.detach().numpy()).max() for k, v in net.named_parameters())
    assert 0 < changed <= 1e-3 * int(shards[0]["steps"]) * 1.001


def test_bench_gpus_2_runs_unaided_on_one_card():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns both ranks before touching the GPU, the ranks
    shard the games, and config C5's data-parallel training step runs with a gradient all-reduce per optimiser step (through
    the host hook: RCCL refuses two ranks on one device — on a multi-GPU node the same code calls tg_train_comm_init)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TAK_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--games", "256", "--rollouts", "24",
           "--blocks", "2", "--filters", "64", "--train-games", "256", "--train-blocks", "2", "--train-filters", "64", "--train-steps", "2",
           "--train-chunk", "40", "--train-chunks-in-step", "2", "--train-example-rollouts", "8", "--profile-every", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["expansions_timed"] == 2 * 256 * 25  # both shards: (1 root evaluation + 24 rollouts) per game and ply
    c5 = out["extra"]["train_c5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["optimizer_steps"] == 2 and c5["parameters_identical_on_all_ranks"] is True
    assert c5["gradient_allreduce"]["count"] == 2 and c5["gradient_allreduce"]["ms_per_step_rank0"] > 0
    assert c5["value"] > 0 and c5["selfplay_c5net"]["value"] > 0
    # round 6: the line proves where the ranks sat and how each of them fared — here both on the one card (gloo: reported, not refused)
    dev = out["config"]["devices"]
    assert [d["rank"] for d in dev] == [0, 1] and dev[0]["pci_bus_id"] == dev[1]["pci_bus_id"] and len(dev[0]["pci_bus_id"]) >= 7
    assert all(d["cu_count"] == 256 and d["name"] and d["arch"].startswith("gfx950") and d["hip_device"] == 0 for d in dev)
    assert dev[0]["pid"] != dev[1]["pid"]
    assert out["config"]["devices_distinct"] is False and out["config"]["backend"] == "gloo" and out["config"]["switches_set"] == []
    by = out["ms_per_step_by_rank"]
    assert len(by["all"]) == 2 and abs(by["max"] - out["ms_per_step"]) < 1e-6 and by["min"] > 0
    pf = c5["gradient_allreduce"]["preflight_ms_by_rank"]
    assert len(pf["all"]) == 2 and pf["min"] > 0 and c5["gradient_allreduce"]["preflight_ms"] == pf["max"]
    assert len(c5["ms_per_optimizer_step_by_rank"]["all"]) == 2 and len(c5["selfplay_c5net"]["ms_per_step_by_rank"]["all"]) == 2


def test_bench_gpus_4_runs_unaided_on_one_card():
    """the same at a wider fan-out: four ranks (four child processes) on the one card.  The 8-GPU node's fan-out itself — port,
    eight children, OMP split, stdout relay, failure propagation, watchdog — is rehearsed without a GPU in
    tests/test_bench_launcher.py; on a one-GPU box at most 6 processes may use the card at once (pool rule), and this test's
    own process is one of them, so four ranks is what fits."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TAK_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--games", "128", "--rollouts", "16",
           "--blocks", "2", "--filters", "64", "--train-games", "128", "--train-blocks", "2", "--train-filters", "64", "--train-steps", "2",
           "--train-chunk", "24", "--train-chunks-in-step", "2", "--train-example-rollouts", "8", "--profile-every", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["value"] > 0 and "train_c5_failed" not in out
    assert out["config"]["expansions_timed"] == 4 * 128 * 17
    c5 = out["extra"]["train_c5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 4 and c5["optimizer_steps"] == 2 and c5["parameters_identical_on_all_ranks"] is True
    assert c5["gradient_allreduce"]["count"] == 2
    assert c5["gradient_allreduce"]["rccl"]["attached"] == 2 and c5["gradient_allreduce"]["rccl"]["world_size"] == 4  # the host hook


def test_device_info_and_switches_of_this_process():
    """tg_device_info: what bench.py's config.devices is built from — the card's PCI bus id as HIP names it, 256 CUs, gfx950 — and
    tg_debug_switches: nothing is switched in a test run that did not set a switch"""
    import tak_amd

    e = tak_amd.Engine(5, evaluator=tak_amd.EVAL_DUMMY, max_batch=1)
    info = e.device_info()
    e.close()
    assert info["hip_device"] == 0 and info["cu_count"] == 256 and info["arch"].startswith("gfx950") and info["total_mem"] > 200 << 30
    dom, bus, rest = info["pci_bus_id"].split(":")
    assert len(dom) == 4 and len(bus) == 2 and "." in rest and info["name"]
    from tak_amd import dist as tdist

    rec = tdist.device_record(3, None)
    assert rec["rank"] == 3 and rec["pci_bus_id"] is None
    assert tak_amd.debug_switches() == [s for s in tak_amd.debug_switches() if s.split("=")[0] in os.environ]


COMM_INFO = r"""
import json, sys
sys.path.insert(0, {root!r})
import numpy as np
{preload}
import tak_amd
from oracle import oracle as orc
e = tak_amd.Engine(5, res_blocks=1, filters=32, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
e.init_random(seed=1)
e.train_create(learning_rate=1e-3, chunk_size=8, chunks_in_step=1)
before = e.train_comm_info()
e.train_comm_init(0, 1, tak_amd.comm_unique_id())
info = e.train_comm_info()
preflight_ms = e.train_comm_preflight()   # the communicator's first collective: 4 bytes, before any training work
sts = orc.random_positions(5, 64, seed=2, max_plies=40, half_komi=4)
sts = sts[orc.result(5, sts) == 0][:8]
mv, cnt = orc.movegen(5, sts)
visits = np.zeros((8, 512), np.uint32)
for i in range(8):
    visits[i, : cnt[i]] = 1
lp, lz, stepped = e.train_chunk(sts, cnt.astype(np.int32), mv, visits, np.zeros(8, np.float32))   # one all-reduced optimiser step
ms, count = e.train_comm_stats()
# tg_train with a step after EVERY chunk: chunk k + 1 and its step are enqueued before chunk k is collected, and still every
# reduction is counted (two event pairs used in turn)
rep = lambda a: np.concatenate([a] * 5)
_, _, steps5 = e.train(rep(sts), rep(cnt.astype(np.int32)), rep(mv), rep(visits), np.zeros(40, np.float32), seed=3)
ms5, count5 = e.train_comm_stats()
e.close()
print("INFO", json.dumps(dict(before=before, info=info, stepped=bool(stepped), reductions=int(count), torch="torch" in sys.modules,
                              preflight_ms=preflight_ms, steps5=int(steps5), reductions5=int(count5), ms5=ms5)))
"""


@pytest.mark.parametrize("with_torch", [True, False])
def test_comm_info_names_the_rccl_that_is_bound(with_torch):
    """ONE copy of RCCL per process, chosen on purpose (train.hip, rccl_load): in a process that has imported torch — bench.py's
    ranks, whose torch.distributed backend is RCCL too — libtakgpu binds the librccl.so.1 torch already mapped
    (torch/lib/librccl.so) instead of loading a second one; a process without one (the Rust host) loads /opt/rocm's.  World size 1
    on the one card: ncclCommCount = 1, and an optimiser step goes through ncclAllReduce of that library."""
    import json

    env = {k: v for k, v in os.environ.items() if not k.startswith("TG_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    code = COMM_INFO.format(root=ROOT, preload="import torch" if with_torch else "")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("INFO")][-1][5:])
    assert r["torch"] == with_torch
    assert r["before"]["attached"] == 0 and r["before"]["nccl_count"] == -1 and r["before"]["world_size"] == 1
    info = r["info"]
    assert info["attached"] == 1 and info["world_size"] == 1 and info["rank"] == 0
    assert info["nccl_count"] == 1 and info["nccl_rank"] == 0 and info["nccl_version"] > 20000
    assert r["stepped"] and r["reductions"] == 1
    assert r["preflight_ms"] > 0
    assert r["steps5"] == 5 and r["reductions5"] == 1 + 5 and r["ms5"] > 0  # no step dropped from the statistics
    if with_torch:
        assert info["lib_was_mapped"] == 1 and info["lib_path"].endswith(os.path.join("torch", "lib", "librccl.so")), info
    else:
        assert info["lib_was_mapped"] == 0 and "torch" not in info["lib_path"] and "librccl.so" in info["lib_path"], info

"""N>1 path on CPU: two gloo ranks each play their shard of the games (the oracle stands in for the GPU
engine); the union of the shards' examples must equal a single-process run of all the games — the
property that makes self-play shard without any data-path collective."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from tak_amd import dist as tdist
from oracle import oracle as orc
rank, world, _ = tdist.env_rank()
dist = tdist.init("gloo", rank, world)
G = 3
sp = orc.SelfPlay(4, G, evaluator=orc.EVAL_HASH, seed=11, rollouts=12, noise_plies=6, exploit_plies=4, total_games=0,
                  slot_base=tdist.slot_base(rank, G))
dist.barrier()
t0 = time.perf_counter()
sp.step(14)
dt_local = time.perf_counter() - t0
dist.barrier()
st = sp.stats()
dt, total = tdist.reduce_time_and_count(dist, dt_local, st["expansions"])
hdr, states, moves, visits = sp.drain(100000)
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), hdr=hdr, states=states, moves=moves, visits=visits,
         total=total, dt=dt, local=st["expansions"])
dist.destroy_process_group()
"""


def test_two_rank_sharding_equals_single_process(orc):
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(WORKER.format(root=ROOT, out=tmp))
        from bench import free_port

        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1")
        procs = []
        for r in range(2):
            e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, script], env=e))
        for p in procs:
            assert p.wait(timeout=300) == 0
        shards = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(2)]
    assert int(shards[0]["total"]) == int(shards[0]["local"]) + int(shards[1]["local"]) == int(shards[1]["total"])
    assert float(shards[0]["dt"]) == float(shards[1]["dt"]) > 0
    # single process, all six slots
    sp = orc.SelfPlay(4, 6, evaluator=orc.EVAL_HASH, seed=11, rollouts=12, noise_plies=6, exploit_plies=4, total_games=0)
    sp.step(14)
    hdr, states, moves, visits = sp.drain(100000)
    assert len(hdr) > 0

    def keyed(h, s, m, v):
        return {(int(h["game_id"][i]), s[i].tobytes()): (m[i].tobytes(), v[i].tobytes(), float(h["result"][i])) for i in range(len(h))}

    single = keyed(hdr, states, moves, visits)
    union = {}
    for sh in shards:
        union.update(keyed(sh["hdr"], sh["states"], sh["moves"], sh["visits"]))
    assert union == single


DP_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from tak_amd import dist as tdist
rank, world, _ = tdist.env_rank()
dist = tdist.init("gloo", rank, world)
uid = tdist.broadcast_unique_id(dist, lambda: bytes((7 * i + 3) % 256 for i in range(128)) if rank == 0 else b"")
b, e = tdist.training_shard(10_345, rank, world, 500)
np.savez(os.path.join({out!r}, f"dp{{rank}}.npz"), uid=np.frombuffer(uid, np.uint8), shard=np.array([b, e]))
dist.destroy_process_group()
"""


def test_data_parallel_host_plumbing():
    """The host side of data-parallel training (config C5): the RCCL unique id reaches every rank, and the ranks
    train on disjoint shards of equally many whole chunks."""
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "dp_worker.py")
        open(script, "w").write(DP_WORKER.format(root=ROOT, out=tmp))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(__import__("bench").free_port()), OMP_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=300) == 0
        out = [np.load(os.path.join(tmp, f"dp{r}.npz")) for r in range(2)]
    want = bytes((7 * i + 3) % 256 for i in range(128))
    assert out[0]["uid"].tobytes() == want and out[1]["uid"].tobytes() == want
    (b0, e0), (b1, e1) = out[0]["shard"], out[1]["shard"]
    assert (b0, e0, b1, e1) == (0, 5000, 5000, 10000)  # 20 whole chunks of 500 → 10 per rank, 345 dropped

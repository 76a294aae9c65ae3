import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, tak_amd, torch_ref
import test_gpu_train as T
from oracle import oracle as orc
ok = True
def check(name, cond):
    global ok
    print(("ok  " if cond else "BAD ") + name, flush=True); ok &= bool(cond)
n = 5
net = torch_ref.make_net(n, 1, 128, "fc5", seed=2)
tensors = torch_ref.abi_tensors(net)
e = tak_amd.Engine(n, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=48)
e.load_state_dict(tensors)
sts = orc.random_positions(n, 400, seed=1, max_plies=50, half_komi=4)
sts = sts[orc.result(n, sts) == 0][:300]
# policy_eval: n = 0, n > max_batch (chunks of 48 through the split tower), n = max_batch + 1
p0, v0 = e.policy_eval(sts[:0]); check("policy_eval of zero positions", p0.shape == (0, 1575) and v0.shape == (0,))
p, v = e.policy_eval(sts); pr, vr = torch_ref.forward(net, orc.encode(n, sts))
check("policy_eval of 300 positions in chunks of 48", np.abs(p - pr).max() <= 1e-4 and np.abs(v - vr).max() <= 1e-4)
p49, v49 = e.policy_eval(sts[:49]); check("49 = max_batch + 1", np.array_equal(p49, p[:49]) and np.array_equal(v49, v[:49]))
# training: fewer examples than a chunk → no chunk, no step; exactly one chunk; forward at Bmax
e.train_create(chunk_size=16, chunks_in_step=2)
ex = T._examples(orc, n, 40, seed=4)
lp, lz, steps = e.train(*[x[:15] for x in ex], seed=1); check("tg_train with fewer examples than a chunk: nothing happens", steps == 0 and lp == 0.0 and lz == 0.0)
w0 = e.train_get_tensor("value.weight", (1, 128 * 25))
lp, lz, steps = e.train(*[x[:16] for x in ex], seed=1); check("one chunk of a two-chunk step: losses, no step", steps == 0 and lp > 0 and np.array_equal(w0, e.train_get_tensor("value.weight", (1, 128 * 25))))
lp, lz, steps = e.train(*[x[:32] for x in ex], seed=1); check("two chunks: one step, parameters moved", steps == 1 and not np.array_equal(w0, e.train_get_tensor("value.weight", (1, 128 * 25))))
a_states, pi = e.augment_examples(*[x[:16] for x in ex[:4]])
logp, ev = e.train_forward(a_states); check("train_forward at 8 x chunk_size positions", logp.shape == (128, 1575) and np.isfinite(logp).all() and abs(np.exp(logp).sum(1) - 1).max() < 1e-4)
try:
    e.train_forward(np.concatenate([a_states, a_states[:1]])); check("train_forward beyond 8 x chunk_size is refused", False)
except tak_amd.TgError as ex_:
    check("train_forward beyond 8 x chunk_size is refused", ex_.code == -1)
try:
    e.train_chunk(*[x[:17] for x in ex]); check("train_chunk beyond chunk_size is refused", False)
except tak_amd.TgError as ex_:
    check("train_chunk beyond chunk_size is refused", ex_.code == -1)
bad = [x.copy() for x in ex]; bad[3][5, :] = 0
try:
    e.train(*[x[:32] for x in bad], seed=1); check("an example without visits is refused before any chunk", False)
except tak_amd.TgError as ex_:
    check("an example without visits is refused before any chunk", ex_.code == -1 and "without visits" in str(ex_))
e.train_commit()
p2, v2 = e.policy_eval(sts[:20]); check("after commit the inference network is the trained one", not np.array_equal(p2, p[:20]) and np.isfinite(p2).all())
# self-play: drain with a small cap, repeatedly; ring overrun is counted, not silent
e.selfplay_create(32, arena_nodes=1 << 12, seed=3, rollouts=6, max_examples=64)
e.selfplay_step(60)
st = e.selfplay_stats()
got = 0
while True:
    h, s_, m_, v_ = e.selfplay_drain(7)
    got += len(h)
    if len(h) == 0: break
check(f"drain in pieces of 7: {got} drained + {st['dropped_examples']} dropped = {st['examples']} emitted", got + st["dropped_examples"] == st["examples"] and st["dropped_examples"] > 0)
e.close()
# search: one game, batch 16 (Player's shape), games = 1
e = tak_amd.Engine(6, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=16)
net6 = torch_ref.make_net(6, 1, 128, "conv", seed=3); e.load_state_dict(torch_ref.abi_tensors(net6))
ev = tak_amd.Engine(6, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=300); ev.load_state_dict(torch_ref.abi_tensors(net6))
root = orc.random_positions(6, 30, seed=9, max_plies=30, half_komi=4); root = root[orc.result(6, root) == 0][:1]
pad = np.concatenate([root] * 300)
def padded(st):
    k = len(st); p_, v_ = ev.policy_eval(np.concatenate([st, pad[:300 - k]])); return p_[:k], v_[:k]
e.search_create(1, arena_nodes=1 << 18, batch=16); e.search_reset(root); e.search_run(25)
s = orc.Search(6, head=orc.HEAD_CONV, py_eval=padded, batch=16); s.reset(root); s.run(25)
a, b = e.search_dump(0), s.dump(0)
check("one game x 16 virtual rollouts on a 6x6 128-filter network: tree = oracle's", len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names))
e.close(); ev.close()
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)

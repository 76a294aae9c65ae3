"""Call shapes at the edges of the C ABI's contracts (round 6; first run as scripts/api_edge_cases.py): zero and over-sized batches, fewer
examples than a chunk, calls beyond a trainer's capacity, a bad example, draining in small pieces with a ring overrun, one game with
Player's batch of 16 on a 6×6 128-filter network."""
import numpy as np
import pytest

import torch_ref

pytestmark = pytest.mark.gpu


def test_policy_eval_and_training_edges(orc):
    import tak_amd
    import test_gpu_train as T

    n = 5
    net = torch_ref.make_net(n, 1, 128, "fc5", seed=2)
    e = tak_amd.Engine(n, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=48)
    e.load_state_dict(torch_ref.abi_tensors(net))
    sts = orc.random_positions(n, 400, seed=1, max_plies=50, half_komi=4)
    sts = sts[orc.result(n, sts) == 0][:300]
    p0, v0 = e.policy_eval(sts[:0])
    assert p0.shape == (0, 1575) and v0.shape == (0,)                      # net5.rs:121-123: an empty batch is an empty answer
    p, v = e.policy_eval(sts)                                               # 300 positions through an engine of 48: chunks, the split tower
    pr, vr = torch_ref.forward(net, orc.encode(n, sts))
    assert np.abs(p - pr).max() <= 1e-4 and np.abs(v - vr).max() <= 1e-4
    p49, v49 = e.policy_eval(sts[:49])
    assert np.array_equal(p49, p[:49]) and np.array_equal(v49, v[:49])
    e.train_create(chunk_size=16, chunks_in_step=2)
    ex = T._examples(orc, n, 40, seed=4)
    w0 = e.train_get_tensor("value.weight", (1, 128 * 25))
    assert e.train(*[x[:15] for x in ex], seed=1) == (0.0, 0.0, 0)          # chunks_exact: fewer examples than a chunk → nothing (network.rs:53)
    lp, lz, steps = e.train(*[x[:16] for x in ex], seed=1)
    assert steps == 0 and lp > 0 and np.array_equal(w0, e.train_get_tensor("value.weight", (1, 128 * 25)))
    lp, lz, steps = e.train(*[x[:32] for x in ex], seed=1)
    assert steps == 1 and not np.array_equal(w0, e.train_get_tensor("value.weight", (1, 128 * 25)))
    a_states, pi = e.augment_examples(*[x[:16] for x in ex[:4]])
    logp, ev = e.train_forward(a_states)
    assert logp.shape == (128, 1575) and np.isfinite(logp).all() and np.abs(np.exp(logp).sum(1) - 1).max() < 1e-4
    with pytest.raises(tak_amd.TgError) as err:
        e.train_forward(np.concatenate([a_states, a_states[:1]]))           # beyond 8 × chunk_size positions
    assert err.value.code == -1
    with pytest.raises(tak_amd.TgError) as err:
        e.train_chunk(*[x[:17] for x in ex])                                 # beyond chunk_size examples
    assert err.value.code == -1
    bad = [x.copy() for x in ex]
    bad[3][5, :] = 0
    with pytest.raises(tak_amd.TgError) as err:
        e.train(*[x[:32] for x in bad], seed=1)                              # an example without visits: refused before any chunk runs
    assert err.value.code == -1 and "without visits" in str(err.value)
    e.train_commit()
    p2, _ = e.policy_eval(sts[:20])
    assert not np.array_equal(p2, p[:20]) and np.isfinite(p2).all()          # the inference network is the trained one
    e.selfplay_create(32, arena_nodes=1 << 12, seed=3, rollouts=6, max_examples=64)
    e.selfplay_step(60)
    st = e.selfplay_stats()
    got = 0
    while True:
        h = e.selfplay_drain(7)[0]
        got += len(h)
        if len(h) == 0:
            break
    assert got + st["dropped_examples"] == st["examples"] and st["dropped_examples"] > 0   # a ring overrun is counted, not silent
    e.close()


def test_one_game_with_players_batch_on_a_wide_6x6_network(orc):
    import tak_amd

    net6 = torch_ref.abi_tensors(torch_ref.make_net(6, 1, 128, "conv", seed=3))
    e = tak_amd.Engine(6, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=16)
    e.load_state_dict(net6)
    ev = tak_amd.Engine(6, res_blocks=1, filters=128, evaluator=tak_amd.EVAL_RESNET, max_batch=300)
    ev.load_state_dict(net6)
    root = orc.random_positions(6, 30, seed=9, max_plies=30, half_komi=4)
    root = root[orc.result(6, root) == 0][:1]
    pad = np.concatenate([root] * 300)

    def padded(st):  # the oracle's evaluator sees every leaf batch inside 300 positions: the one-workgroup-per-position tower
        k = len(st)
        p_, v_ = ev.policy_eval(np.concatenate([st, pad[: 300 - k]]))
        return p_[:k], v_[:k]

    e.search_create(1, arena_nodes=1 << 18, batch=16)
    e.search_reset(root)
    e.search_run(25)
    s = orc.Search(6, head=orc.HEAD_CONV, py_eval=padded, batch=16)
    s.reset(root)
    s.run(25)
    a, b = e.search_dump(0), s.dump(0)
    assert len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)
    e.close()
    ev.close()

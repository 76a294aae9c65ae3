"""Behavioural checks of the oracle's MCTS against alpha-tak/src/search/tests.rs:37-72 (DummyNet)."""
import numpy as np


def _best_move(root, g=0):
    c = root["counts"][g]
    v = root["visits"][g, :c]
    # max_by_key → last maximum (play.rs:56)
    return root["moves"][g, c - 1 - int(np.argmax(v[::-1]))]


def test_win_in_one(orc):
    n = 3
    st = orc.from_ptn(n, ["a3", "c3", "c2", "a2"])
    s = orc.Search(n, head=orc.HEAD_CONV, evaluator=orc.EVAL_DUMMY)
    s.reset(st)
    assert s.run(1000) == 0
    mv = _best_move(s.root())
    st2, status = orc.play(n, st, [mv])
    assert status[0] == 0
    assert orc.result(n, st2)[0] == 1  # Winner { White, road }


def test_prevent_win_in_two(orc):
    n = 3
    st = orc.from_ptn(n, ["a3", "c3", "c2"])
    s = orc.Search(n, evaluator=orc.EVAL_DUMMY)
    s.reset(st)
    s.run(1000)
    mv = _best_move(s.root())
    assert s.play([mv]) == 0
    st = s.states()
    assert orc.result(n, st)[0] == 0
    s.run(1000)
    mv = _best_move(s.root())
    st2, status = orc.play(n, st, [mv])
    assert status[0] == 0 and orc.result(n, st2)[0] == 0


def test_tree_invariants_hash_eval(orc):
    n = 5
    sts = orc.random_positions(n, 8, seed=3, max_plies=30, half_komi=4)
    sts = sts[orc.result(n, sts) == 0]
    s = orc.Search(n, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH)
    s.reset(sts)
    s.run(200)
    r = s.root()
    for g in range(len(sts)):
        c = r["counts"][g]
        # every rollout adds exactly one visit to the root; children's visits sum to root-1 (first visit expands)
        assert r["root_visits"][g] == 200
        assert r["visits"][g, :c].sum() == 199
        rec = s.dump(g)
        assert rec[0]["visits"] == 200 and (rec["virtual_visits"] == 0).all()
    exp, ev = s.counters()
    assert exp == 200 * len(sts) and ev <= exp


def test_selfplay_small(orc):
    sp = orc.SelfPlay(4, games=6, evaluator=orc.EVAL_HASH, rollouts=20, total_games=10, seed=7)
    for _ in range(200):
        sp.step(1)
        _, alive = sp.states()
        if not alive.any():
            break
    st = sp.stats()
    assert st["games_finished"] >= 10 - 6 + 1
    hdr, states, moves, visits = sp.drain(100000)
    assert len(hdr) == st["examples"] > 0
    assert set(np.unique(hdr["result"])).issubset({-1.0, 0.0, 1.0})
    # determinism
    sp2 = orc.SelfPlay(4, games=6, evaluator=orc.EVAL_HASH, rollouts=20, total_games=10, seed=7)
    for _ in range(200):
        sp2.step(1)
        if not sp2.states()[1].any():
            break
    hdr2, states2, moves2, visits2 = sp2.drain(100000)
    assert np.array_equal(hdr, hdr2) and np.array_equal(states, states2) and np.array_equal(visits, visits2)


def test_dirichlet_noise_distribution(orc):
    """apply_dirichlet (alpha-tak/src/search/noise.rs:6-16) draws from rand_distr::Dirichlet with thread_rng — only the
    DISTRIBUTION can be compared.  The counter-based replacement must have Dirichlet(α·1) marginals: each component is
    Beta(α, (K-1)α): mean 1/K, variance (1/K)(1-1/K)/(Kα+1); components sum to 1."""
    import numpy as np

    alpha, k, draws = 0.2, 30, 4000
    x = np.stack([orc.dirichlet(k, alpha, 77, g, 0, 3) for g in range(draws)]).astype(np.float64)
    assert np.all(x >= 0) and np.allclose(x.sum(1), 1.0, atol=1e-5)
    mean, var = 1.0 / k, (1.0 / k) * (1 - 1.0 / k) / (k * alpha + 1)
    # standard error of the mean of one component over `draws` samples = sqrt(var/draws); 5 sigma over 30 components
    assert np.all(np.abs(x.mean(0) - mean) < 5 * np.sqrt(var / draws))
    assert abs(x.var(0).mean() - var) < 0.05 * var
    # against an independent sampler (numpy's Dirichlet): the mean of the largest component and the quantiles of one component
    ref = np.random.default_rng(0).dirichlet([alpha] * k, 20000)
    assert abs(x.max(1).mean() - ref.max(1).mean()) < 0.01
    qs = [0.5, 0.75, 0.9, 0.97]
    assert np.allclose(np.quantile(x.ravel(), qs), np.quantile(ref.ravel(), qs), rtol=0.08, atol=2e-4)
    # different streams (slot, generation, ply) are different draws; the same key is the same draw
    assert not np.array_equal(orc.dirichlet(k, alpha, 77, 0, 0, 3), orc.dirichlet(k, alpha, 77, 0, 1, 3))
    assert np.array_equal(orc.dirichlet(k, alpha, 77, 5, 2, 9), orc.dirichlet(k, alpha, 77, 5, 2, 9))

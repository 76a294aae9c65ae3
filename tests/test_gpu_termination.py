"""GPU parity on the rarely reached branches of Game::result / Game::play (reference tak/src/game.rs:211-267,
tile.rs:28-63): the 50-reversible-plies draw, very tall stacks (the high half of the u64 colour word), every
ending of a game on every board size including 6×6, and a sweep over a million distinct positions.
Bit-exact against the CPU oracle through the C ABI."""
import numpy as np
import pytest

import posgen

pytestmark = pytest.mark.gpu

RES = {"ongoing": 0, "white_road": 1, "white_flat": 2, "black_road": 3, "black_flat": 4, "draw": 5, "draw_reversible": 6}


@pytest.fixture(scope="module")
def engines():
    import tak_amd

    es = {n: tak_amd.Engine(n, evaluator=tak_amd.EVAL_HASH, max_batch=16384,
                            policy_head=tak_amd.HEAD_FC5 if n == 5 else tak_amd.HEAD_CONV) for n in (3, 4, 5, 6)}
    yield es
    for e in es.values():
        e.close()


def _assert_movegen_equal(gm, gc, om, oc):
    assert np.array_equal(gc, oc)
    live = np.arange(gm.shape[1])[None, :] < gc[:, None]
    assert np.array_equal(np.where(live, gm, 0), np.where(live, om, 0))  # same moves in the same order


def _play_every_move(e, orc, n, sts, cap=None):
    """Game::play of every legal move of every state: states, status and the result afterwards"""
    om, oc = orc.movegen(n, sts)
    rep = np.repeat(np.arange(len(sts)), oc)
    mv = np.concatenate([om[i, : oc[i]] for i in range(len(sts))]) if len(sts) else np.zeros(0, np.uint16)
    if cap and len(rep) > cap:
        pick = np.random.default_rng(0).choice(len(rep), cap, replace=False)
        rep, mv = rep[pick], mv[pick]
    g_states, g_status = e.play(sts[rep], mv)
    o_states, o_status = orc.play(n, sts[rep], mv)
    assert not o_status.any() and np.array_equal(g_status, o_status)
    assert np.array_equal(g_states, o_states)
    g_res, o_res = e.result(g_states), orc.result(n, o_states)
    assert np.array_equal(g_res, o_res)
    return o_states, o_res, mv


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_reversible_plies_boundary(engines, orc, n):
    # game.rs:211-218 (a placement resets the counter, a spread adds one) and :257-261 (≥ 50 → Draw{reversible_plies: true},
    # but only after roads and flat counts).  Roots with the counter just below / at / above the limit and at the u8 edge.
    e = engines[n]
    base = orc.random_positions(n, 3000, seed=41 + n, max_plies=80 if n >= 5 else 24, half_komi=4)
    base = base[(posgen.header(base, "ply") >= 2)]
    seen = set()
    for rev in (48, 49, 50, 51, 254, 255):
        sts = posgen.with_header(base, reversible_plies=rev)
        g, o = e.result(sts), orc.result(n, sts)
        assert np.array_equal(g, o)
        seen |= set(o.tolist())
        if rev >= 50:
            assert (o == RES["draw_reversible"]).sum() > len(sts) // 2  # the branch is taken unless a road / flat count ends it
        _assert_movegen_equal(*e.movegen(sts), *orc.movegen(n, sts))
        ongoing = sts[o == 0][:300] if rev < 50 else sts[:300]  # (play does not look at the result: also from ended positions)
        after, res, mv = _play_every_move(e, orc, n, ongoing, cap=60000)
        spread = (mv >> 8) != 0
        want = np.where(spread, (rev + 1) & 0xFF, 0)
        assert np.array_equal(posgen.header(after, "reversible_plies"), want)
        if rev == 49:  # place vs spread as the 50th reversible ply
            assert (res[spread] == RES["draw_reversible"]).any() and not (res[~spread] == RES["draw_reversible"]).any()
        seen |= set(res.tolist())
    assert RES["draw_reversible"] in seen and RES["ongoing"] in seen


@pytest.mark.parametrize("n", [5, 6])
def test_search_sees_the_reversible_draw(engines, orc, n):
    # trees rooted one and two spreads away from the 50-ply draw: terminal children carry Draw{reversible} and a reward of 0
    e = engines[n]
    base = orc.random_positions(n, 400, seed=3 + n, max_plies=60, half_komi=4)
    base = base[(posgen.header(base, "ply") >= 2) & (orc.result(n, base) == 0)][:48]
    roots = np.concatenate([posgen.with_header(base[:24], reversible_plies=49), posgen.with_header(base[24:], reversible_plies=48)])
    games = len(roots)
    e.search_create(games, arena_nodes=1 << 14, seed=9)
    e.search_reset(roots)
    s = orc.Search(n, head=orc.HEAD_FC5 if n == 5 else orc.HEAD_CONV, evaluator=orc.EVAL_HASH, seed=9)
    s.reset(roots)
    e.search_run(96)
    s.run(96)
    found = 0
    for g in range(games):
        a, b = e.search_dump(g), s.dump(g)
        assert len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names), f"tree {g} differs"
        found += int((b["result"] == RES["draw_reversible"]).sum())
    assert found > games  # the draw was reached inside the trees


@pytest.mark.parametrize("n", [5, 6])
def test_tall_stacks(engines, orc, n):
    # stacks of 33…44 (5×5) / 33…62 (6×6) stones: colour bits ≥ 32 set, carries taken from the top of such a stack, the deep
    # planes of game_repr, TPS text and the 8 symmetries (tile.rs:28-63, repr/board.rs:36-46, tps.rs, symm.rs)
    import tak_amd

    e = engines[n]
    S, Cc = posgen.STONES[n]
    sts = np.concatenate([posgen.tall_stack_states(n, 1500, seed=17 * n),
                          posgen.tall_stack_states(n, 500, seed=18 * n, lo=2 * S - 2)])
    hs = posgen.heights(sts, n).max(axis=1)
    assert hs.min() >= 33 and hs.max() == 2 * S + 1  # every flat of both colours under one capstone
    words = sts[:, : 8 * n * n].view(np.uint64)
    assert (words >> np.uint64(32)).any() and (words >> np.uint64(2 * S)).any()
    _assert_movegen_equal(*e.movegen(sts), *orc.movegen(n, sts))
    assert np.array_equal(e.result(sts), orc.result(n, sts))
    assert np.array_equal(e.encode(sts), orc.encode(n, sts))
    after, _, _ = _play_every_move(e, orc, n, sts[:600], cap=120000)
    assert np.array_equal(e.encode(after[:4000]), orc.encode(n, after[:4000]))
    for st in sts[:200]:
        text = tak_amd.format_tps(n, st)
        assert text == orc.to_tps(n, st)
        back = tak_amd.parse_tps(n, text)  # reserves derived from the board, half_komi 0, reversible 0
        want = posgen.with_header(st[None], half_komi=0, reversible_plies=0)[0]
        want = posgen.with_header(want[None], ply=2 * (posgen.header(st[None], "ply") // 2) + posgen.header(st[None], "to_move"))[0]
        assert np.array_equal(back[: 9 * (25 if n <= 5 else 36)], want[: 9 * (25 if n <= 5 else 36)])
    # Example::to_tensors on them: 8 images of the state and of the visit distribution
    sub = sts[orc.result(n, sts) == 0][:256]
    om, oc = orc.movegen(n, sub)
    rng = np.random.default_rng(5)
    vis = np.where(np.arange(om.shape[1])[None, :] < oc[:, None], rng.integers(1, 50, om.shape), 0).astype(np.uint32)
    head = orc.HEAD_FC5 if n == 5 else orc.HEAD_CONV
    gs, gp = e.augment_examples(sub, oc, om, vis)
    os_, op = orc.augment(n, head, sub, oc, om, vis)
    assert np.array_equal(gs, os_) and np.array_equal(gp, op)


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_every_ending_of_whole_games(engines, orc, n):
    # whole games steered (by the oracle's move classes) into each ending of Game::result: roads of either colour, flat
    # counts on a full board and on exhausted reserves under every komi parity (half_komi % 2 branch), both draws.
    e = engines[n]
    d = posgen.terminal_mix(orc, n, per_style=3000, seed=n)
    final, prev, mv, res = d["final"], d["prev"], d["move"], d["result"]
    counts = np.bincount(res, minlength=7)
    need = 1000 if n >= 5 else 300
    assert counts[0] == 0 and (counts[1:] >= need).all(), counts  # ≥ 1000 terminal positions per result code on 5×5 and 6×6
    assert np.array_equal(e.result(final), res)
    assert not e.result(prev).any()                                # the game was still running one ply earlier …
    g_states, g_status = e.play(prev, mv)                          # … and the last move ends it the same way
    assert not g_status.any() and np.array_equal(g_states, final)
    _assert_movegen_equal(*e.movegen(final), *orc.movegen(n, final))
    assert np.array_equal(e.encode(final[:3000]), orc.encode(n, final[:3000]))
    # endings by kind: full board vs exhausted reserves, odd and even half-komi among the flat counts and draws
    flat = np.isin(res, (2, 4, 5))
    full = (posgen.heights(final, n) > 0).all(axis=1)
    out_of = ((posgen.header(final, "white_stones") == 0) & (posgen.header(final, "white_caps") == 0)) | \
             ((posgen.header(final, "black_stones") == 0) & (posgen.header(final, "black_caps") == 0))
    assert (flat & full).sum() > 100 and (flat & out_of & ~full).sum() > 100
    hk = posgen.header(final, "half_komi").astype(int)
    assert ((res == 5) & (hk % 2 == 0)).sum() > 100 and not ((res == 5) & (hk % 2 != 0)).any()
    assert ((res == 4) & (hk % 2 != 0)).sum() > 100
    assert posgen.heights(final, n).max() > (12 if n >= 5 else 5)


@pytest.mark.parametrize("n", [5, 6])
def test_million_distinct_positions(engines, orc, n):
    # movegen (order-exact), result and game_repr on ≥ 1 M distinct positions per size
    e = engines[n]
    parts = [orc.random_positions(n, 700_000, seed=100 + n, max_plies=140 if n == 5 else 200, half_komi=4),
             orc.random_positions(n, 450_000, seed=200 + n, max_plies=60, half_komi=0)]
    d = posgen.terminal_mix(orc, n, per_style=6000, seed=50 + n)
    sts = np.concatenate(parts + [d["final"], d["prev"]])
    sts = np.unique(sts.view(np.dtype((np.void, sts.shape[1]))).ravel()).view(np.uint8).reshape(-1, sts.shape[1])
    assert len(sts) >= 1_000_000, len(sts)
    sts = sts[np.random.default_rng(n).permutation(len(sts))[:1_000_000]]
    res_g = np.zeros(len(sts), np.uint8)
    for lo in range(0, len(sts), 1 << 16):
        chunk = sts[lo: lo + (1 << 16)]
        _assert_movegen_equal(*e.movegen(chunk), *orc.movegen(n, chunk))
        res_g[lo: lo + len(chunk)] = e.result(chunk)
        assert np.array_equal(res_g[lo: lo + len(chunk)], orc.result(n, chunk))
        assert np.array_equal(e.encode(chunk), orc.encode(n, chunk))
    assert len(set(res_g.tolist())) == 7  # every result code occurs in the sweep

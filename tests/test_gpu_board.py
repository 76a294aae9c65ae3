"""GPU parity of the board kernels (through the C ABI) against the CPU oracle and against the
reference's own known-answer values.  Bit-exact: integer / byte work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    import tak_amd

    es = {n: tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=4096,
                            policy_head=tak_amd.HEAD_FC5 if n == 5 else tak_amd.HEAD_CONV) for n in (3, 4, 5, 6)}
    yield es
    for e in es.values():
        e.close()


def _positions(orc, n, count, seed):
    a = orc.random_positions(n, count, seed=seed, max_plies=90 if n >= 5 else 30, half_komi=4)
    b = orc.random_positions(n, count // 4, seed=seed + 1, max_plies=6, half_komi=0)
    return np.concatenate([a, b, orc.new_game(n)[None]])


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_movegen_result_encode_match_oracle(engines, orc, n):
    e = engines[n]
    sts = _positions(orc, n, 6000, seed=11 * n)
    gm, gc = e.movegen(sts)
    om, oc = orc.movegen(n, sts)
    assert np.array_equal(gc, oc)
    for i in range(len(sts)):
        assert np.array_equal(gm[i, : gc[i]], om[i, : oc[i]]), i  # same moves in the same order
    assert np.array_equal(e.result(sts), orc.result(n, sts))
    sub = sts[:1500]
    assert np.array_equal(e.encode(sub), orc.encode(n, sub))  # planes are exact (0/1 and the f64-divided fcd)


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_play_matches_oracle(engines, orc, n):
    e = engines[n]
    sts = _positions(orc, n, 3000, seed=5 * n)
    sts = sts[orc.result(n, sts) == 0]
    om, oc = orc.movegen(n, sts)
    rng = np.random.default_rng(n)
    # every legal move of the first 200 positions, one random legal move for the rest
    rep, mv = [], []
    for i in range(len(sts)):
        ks = range(oc[i]) if i < 200 else [rng.integers(oc[i])]
        for k in ks:
            rep.append(i)
            mv.append(om[i, k])
    rep, mv = np.array(rep), np.array(mv, np.uint16)
    g_states, g_status = e.play(sts[rep], mv)
    o_states, o_status = orc.play(n, sts[rep], mv)
    assert not g_status.any() and not o_status.any()
    assert np.array_equal(g_states, o_states)
    assert np.array_equal(e.result(g_states), orc.result(n, o_states))


@pytest.mark.parametrize("n", [3, 5, 6])
def test_play_error_codes_match_oracle(engines, orc, n):
    # arbitrary (mostly illegal) move codes: same PlayError as the oracle, state untouched on error
    e = engines[n]
    sts = _positions(orc, n, 1500, seed=77 + n)
    rng = np.random.default_rng(3)
    sq = rng.integers(0, n * n + 2, len(sts))
    kind = rng.integers(0, 4, len(sts))
    pat = np.where(rng.random(len(sts)) < 0.4, 0, rng.integers(1, 256, len(sts)))
    mv = (sq | (kind << 6) | (pat << 8)).astype(np.uint16)
    g_states, g_status = e.play(sts, mv)
    o_states, o_status = orc.play(n, sts, mv)
    assert np.array_equal(g_status, o_status)
    assert len(set(g_status.tolist())) >= 6  # several distinct error kinds exercised
    assert np.array_equal(g_states, o_states)


def test_move_index_matches_oracle(engines, orc):
    for n in (3, 4, 5, 6):
        e = engines[n]
        sts = _positions(orc, n, 2000, seed=9 + n)
        om, oc = orc.movegen(n, sts)
        mv = np.unique(np.concatenate([om[i, : oc[i]] for i in range(len(sts))]))
        assert np.array_equal(e.move_index(mv), orc.move_index(n, mv))
    # the whole legacy table on 5x5
    table = orc.legacy5_table()
    codes = np.array([orc.parse_move(5, s) for s in table], np.uint16)
    assert np.array_equal(engines[5].move_index(codes), np.arange(1575))


def test_gpu_perft_reference_kats(engines, orc, kats):
    # tak/tests/perft.rs: the reference's own node counts, reproduced by the GPU movegen/play/result kernels
    for k in kats["perft"]:
        st = orc.from_ptn(k["n"], k["moves"])
        got = int(engines[k["n"]].perft(st, k["depth"])[0])
        assert got == k["count"], (k["name"], k["depth"], got)


def test_gpu_wins_and_tps_kats(engines, orc, kats):
    res = {"Ongoing": 0, "WhiteRoad": 1, "WhiteFlat": 2, "BlackRoad": 3, "BlackFlat": 4, "Draw": 5, "DrawReversible": 6}
    for k in kats["wins"]:
        n = k["n"]
        e = engines[n]
        st = orc.new_game(n)
        for ptn in k["moves"]:  # play the whole game on the GPU
            st, status = e.play(st, [orc.parse_move(n, ptn)])
            assert status[0] == 0
        st = st[0].copy()
        st[e.sb - 16 + 8] = np.uint8(k["half_komi"] & 0xFF)
        assert e.result(st)[0] == res[k["result"]], k["name"]
    k = kats["tps"]
    e = engines[6]
    st = orc.new_game(6)
    for ptn in k["moves"]:
        st, status = e.play(st, [orc.parse_move(6, ptn)])
        assert status[0] == 0, ptn
    assert orc.to_tps(6, st[0]) == k["tps"]  # stack order after 107 plies of GPU spreads / smashes


def test_gpu_repr_kat(engines, orc, kats):
    k = kats["repr"]
    st = orc.from_ptn(5, k["moves"])
    st[engines[5].sb - 16 + 1] = 0  # board_repr(&board, Color::White)
    planes = engines[5].encode(st)[0]
    want = np.array([int(c) for c in k["bits"]], np.float32).reshape(k["planes"], 5, 5)
    assert np.array_equal(planes[: k["planes"]], want) and not planes[k["planes"] : 26].any()


def test_empty_and_ragged_batches(engines, orc):
    e = engines[5]
    assert e.movegen(np.zeros((0, 256), np.uint8))[1].shape == (0,)
    assert e.result(np.zeros((0, 256), np.uint8)).shape == (0,)
    sts = _positions(orc, 5, 5000, seed=1)[:4097 + 13]  # crosses the max_batch chunk boundary
    assert np.array_equal(e.result(sts), orc.result(5, sts))
    assert np.array_equal(e.movegen(sts)[1], orc.movegen(5, sts)[1])

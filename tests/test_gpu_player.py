"""`Player` mirror (alpha-tak/src/player.rs) on the engine: the reference's two behavioural tests (search/tests.rs:37-72)
driven through the Player API, a full game with example collection, and the examples' value targets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_win_in_one_and_prevent_win_in_two(orc):
    import tak_amd

    e = tak_amd.Engine(3, evaluator=tak_amd.EVAL_DUMMY, max_batch=64, policy_head=tak_amd.HEAD_CONV)
    game = orc.from_ptn(3, ["a3", "c3", "c2", "a2"])
    p = tak_amd.Player(e, batch=100, save_examples=False, game=game)
    for _ in range(10):
        p.rollout()
    mv = p.pick_move(True)
    st, status = orc.play(3, game, [mv])
    assert status[0] == 0 and orc.result(3, st)[0] == 1  # white completes a road (search/tests.rs:37-52)
    game = orc.from_ptn(3, ["a3", "c3", "c2"])
    p = tak_amd.Player(e, batch=100, save_examples=False, game=game)
    for _ in range(10):
        p.rollout()
    st, _ = orc.play(3, game, [p.pick_move(True)])
    # black's best reply must not leave a win in one (search/tests.rs:54-72)
    lm, lc = orc.movegen(3, st)
    nxt, status = orc.play(3, np.repeat(st, lc[0], axis=0), lm[0, : lc[0]])
    assert not (orc.result(3, nxt)[status == 0] == 1).any()
    e.close()


def test_full_game_with_examples(orc):
    import tak_amd

    n = 4
    e = tak_amd.Engine(n, evaluator=tak_amd.EVAL_HASH, max_batch=64, policy_head=tak_amd.HEAD_CONV)
    game = orc.new_game(n, half_komi=4)
    p = tak_amd.Player(e, batch=16, save_examples=True, game=game, seed=5)
    plies = 0
    while orc.result(n, game)[0] == 0 and plies < 120:
        if plies < 4:
            p.add_noise(0.2, 0.3)
        for _ in range(3):
            p.rollout()
        mv = p.pick_move(plies >= 6)
        legal, cnt = orc.movegen(n, game)
        assert mv in legal[0, : cnt[0]]
        p.play_move(mv)
        game = orc.play(n, game, [mv])[0][0]
        assert np.array_equal(p.state(), game)  # the engine's root state follows the oracle's game
        plies += 1
    res = int(orc.result(n, game)[0])
    assert res != 0
    states, n_moves, moves, visits, results = p.get_examples(res)
    assert len(states) == plies and (n_moves > 0).all()
    white = 1.0 if res in (1, 2) else -1.0 if res in (3, 4) else 0.0
    for i in range(plies):
        to_move = states[i][256 - 16 + 1]
        assert results[i] == (white if to_move == 0 else -white)
        assert visits[i, : n_moves[i]].sum() > 0
    assert p.get_examples(res)[0].shape[0] == 0  # taken
    e.close()

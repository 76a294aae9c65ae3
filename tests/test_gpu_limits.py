"""Per-game capacities in self-play (include/takgpu.h TG_LIMIT_*): the reference's heap structures have none
(train/src/self_play.rs:108-259 — a Vec of examples per game, boxed tree nodes), the engine's fixed-size staging areas and
tables do.  A game that runs into one is retired ALONE — its examples discarded, its slot restarted as the next generation,
TgSelfPlayStats.aborted_games counting it — and the other games must not notice: every game that completes is, example for
example, the game the unlimited run plays under the same (slot, generation) key.  The capacities are reached here by
lowering them (TgSelfPlayConfig.max_game_plies, TgSearchConfig.visit_limit): a natural 513-ply game cannot be forced, the
code path (stage_example / the exploration-rate table bound in select) is the same."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, G, PLIES = 5, 48, 150
KW = dict(rollouts=20, noise_plies=12, exploit_plies=8, total_games=0, seed=21, arena_nodes=1 << 14, max_examples=1 << 15)


def _games(drain):
    hdr, states, moves, visits = drain
    out = {}
    for i in range(len(hdr)):
        out.setdefault(int(hdr["game_id"][i]), []).append((states[i].tobytes(), moves[i].tobytes(), visits[i].tobytes(), float(hdr["result"][i])))
    return out


def _run(**over):
    import tak_amd

    e = tak_amd.Engine(N, evaluator=tak_amd.EVAL_HASH, max_batch=64)
    e.selfplay_create(G, **{**KW, **over})
    e.selfplay_step(PLIES)
    st = e.selfplay_stats()          # raises if the engine carries a sticky error
    games = _games(e.selfplay_drain(1 << 15))
    e.selfplay_step(1)               # … and it keeps going
    e.close()
    return st, games


def test_a_game_past_max_game_plies_is_retired_alone(orc):
    full_st, full = _run()
    assert full_st["aborted_games"] == 0 and full_st["alive_games"] == G
    lengths = sorted(len(v) for v in full.values())
    limit = lengths[len(lengths) // 2]  # the median game length: about half of the games are longer
    assert 8 <= limit < 200
    st, cut = _run(max_game_plies=limit)
    assert st["aborted_games"] > 0 and st["alive_games"] == G and st["dropped_examples"] == 0
    assert st["games_finished"] == len(cut) and st["examples"] == sum(len(v) for v in cut.values())
    assert st["white_wins"] + st["black_wins"] + st["draws"] == st["games_finished"]  # a retired game is no result
    assert max(len(v) for v in cut.values()) <= limit
    # The same game under the same (slot, generation) key, whatever happened to the slot before it.  (One thing does carry
    # over, in the reference too: a slot recycled by the instant-win scan — phase (b), after the opening phase (a) of that
    # ply — starts its next game WITHOUT the forced opening.  So a game whose predecessor ended differently in the two runs
    # — instant win in one, retired at the move choice in the other — may start from another position: then it is a
    # different game; from the same first position it must be the same game.)
    common = set(cut) & set(full)
    same_start = [gid for gid in common if cut[gid][0][0] == full[gid][0][0]]
    assert len(same_start) >= 10 and len(same_start) >= len(common) // 2
    for gid in same_start:
        assert cut[gid] == full[gid], f"game {gid:#x} differs from the unlimited run"
    for gid in common - set(same_start):
        assert (gid >> 20) > 0 and (gid - (1 << 20)) not in cut  # its predecessor was retired in the limited run
    # games of the unlimited run that fit under the limit and started early enough were all played by the limited run too
    short_first = {gid for gid, v in full.items() if (gid >> 20) == 0 and len(v) <= limit}
    assert short_first and short_first <= set(cut)
    # more games get started when long ones are cut short: later generations appear
    assert max(g >> 20 for g in cut) >= max(g >> 20 for g in full)
    # and the oracle (no capacities, as the reference) agrees on every completed game
    okw = {k: v for k, v in KW.items() if k not in ("arena_nodes", "max_examples")}
    sp = orc.SelfPlay(N, G, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH, **okw)
    sp.step(PLIES)
    ref = _games(sp.drain(1 << 15))
    assert ref == full
    assert sp.stats()["aborted_games"] == 0


def test_a_game_past_the_visit_table_is_retired_alone():
    import tak_amd

    # 20 rollouts + the root evaluation per ply, plus the visits the kept subtree brings along (tree reuse): a root whose kept
    # subtree is well visited passes a table of 24 – 48 entries; find a length at which some games are retired and some finish
    for limit in (48, 40, 32, 28, 24):
        st, games = _run(visit_limit=limit)
        assert st["alive_games"] == G and st["games_finished"] == len(games)
        if st["aborted_games"] > 0 and len(games) > 0:
            break
    else:
        pytest.fail("no table length retired some games and let others finish")
    full_st, full = _run()
    for gid in set(games) & set(full):
        if games[gid][0][0] == full[gid][0][0]:  # (same first position: see the note on the opening in the test above)
            assert games[gid] == full[gid]
    # a caller-driven search has nobody to restart the game: the capacity stays a sticky engine error there
    e = tak_amd.Engine(N, evaluator=tak_amd.EVAL_HASH, max_batch=64)
    e.search_create(4, arena_nodes=1 << 14, visit_limit=64)
    start = np.zeros((4, e.sb), np.uint8)
    hdr = e.sb - 16
    start[:, hdr + 0] = N
    start[:, hdr + 4], start[:, hdr + 5], start[:, hdr + 6], start[:, hdr + 7] = 21, 1, 21, 1
    start[:, hdr + 8] = 4
    e.search_reset(start)
    e.search_run(100)
    with pytest.raises(tak_amd.TgError) as err:
        e.search_counters()
    assert err.value.code == -9 and "TG_LIMIT_VISITS" in str(err.value)  # TG_ERR_LIMIT
    e.close()


def test_limits_are_validated():
    import tak_amd

    e = tak_amd.Engine(N, evaluator=tak_amd.EVAL_HASH, max_batch=64)
    with pytest.raises(tak_amd.TgError):
        e.selfplay_create(4, **{**KW, "max_game_plies": 513})
    with pytest.raises(tak_amd.TgError):
        e.selfplay_create(4, **{**KW, "visit_limit": 8})
    e.close()

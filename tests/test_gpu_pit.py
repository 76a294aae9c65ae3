"""Pit (tg_pit ≙ `pit`, train/src/pit.rs:15-96): the GPU match against a CPU replay of the same loop on the oracle's
scalar MCTS (two trees per game, one per network), with the deterministic test evaluators so that whole games are
bit-identical; plus the symmetry property that two identical networks split every pair of games."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RNG_PIT_CORNER, RNG_PIT_RANDOM = 8, 9


def _oracle_pit(orc, n, evals, pairs, rollouts, idle, random_plies, komi, seed, max_plies=0, batch=1):
    head = orc.HEAD_FC5 if n == 5 else orc.HEAD_CONV
    op = np.stack([orc.new_game(n, half_komi=2 * komi) for _ in range(pairs)])
    for ply in range(2 + random_plies):
        if ply == 0:
            mv = np.zeros(pairs, np.uint16)
        elif ply == 1:
            mv = np.array([n * n - 1 if orc.philox(seed, p, 0, 1 | (RNG_PIT_CORNER << 16), 0)[0] & 1 else (n - 1) * n for p in range(pairs)], np.uint16)
        else:
            lm, lc = orc.movegen(n, op)
            mv = np.zeros(pairs, np.uint16)
            for p in range(pairs):
                ok = [int(m) for m in lm[p, : lc[p]] if (m >> 8) == 0 and ((m >> 6) & 3) != 1]
                r = orc.philox(seed, p, 0, ply | (RNG_PIT_RANDOM << 16), 0)
                x = (int(r[0]) << 32) | int(r[1])
                mv[p] = ok[(x * len(ok)) >> 64]
        op, status = orc.play(n, op, mv)
        assert not status.any()
    G = 2 * pairs
    states = np.repeat(op, 2, axis=0)
    # an evaluator is one of the oracle's deterministic kinds or a callable (states → policy, eval): a network behind tg_policy_eval
    trees = [orc.Search(n, head=head, py_eval=ev, seed=seed, batch=batch) if callable(ev) else orc.Search(n, head=head, evaluator=ev, seed=seed, batch=batch)
             for ev in evals]
    for t in trees:
        t.reset(states)
    alive = np.ones(G, bool)
    final = np.zeros(G, np.int64)
    wins = losses = draws = plies = 0
    sb = states.shape[1]
    while True:
        res = orc.result(n, states)
        for g in range(G):
            if alive[g] and res[g] != 0:
                alive[g] = False
                final[g] = res[g]
                if res[g] in (5, 6):
                    draws += 1
                elif (res[g] in (1, 2)) == (g % 2 == 0):
                    wins += 1
                else:
                    losses += 1
        if not alive.any() or (max_plies and plies >= max_plies):
            break
        to_move = states[:, sb - 16 + 1]
        new_to_move = alive & ((to_move == 0) == (np.arange(G) % 2 == 0))
        act = [new_to_move, alive & ~new_to_move]
        roots = []
        for k in range(2):
            trees[k].run(rollouts, act[k].astype(np.uint8))
            trees[k].run(idle, act[1 - k].astype(np.uint8))
            roots.append(trees[k].root())
        chosen = np.zeros(G, np.uint16)
        for g in range(G):
            if alive[g]:
                r = roots[0 if act[0][g] else 1]
                v = r["visits"][g, : r["counts"][g]]
                best = len(v) - 1 - int(np.argmax(v[::-1]))  # the LAST maximum (play.rs:54-57)
                chosen[g] = r["moves"][g, best]
        for t in trees:
            t.play(chosen, alive.astype(np.uint8))
        states = trees[0].states()
        plies += 1
    return dict(wins=wins, losses=losses, draws=draws, plies=plies, unfinished=int(alive.sum()), **_reference_tally(final, pairs))


def _reference_tally(final, pairs):
    """`pit` plays the openings one after the other, White game then Black game, and breaks once the verdict is known
    (pit.rs:20-23): the counts it returns, from the per-game results of all games."""
    w = l = d = played = 0
    for i in range(pairs):
        if w > pairs + pairs // 10 or l > pairs - pairs // 10:
            break
        played += 1
        for c in range(2):
            r = final[2 * i + c]
            if r == 0:
                continue
            if r in (5, 6):
                d += 1
            elif (r in (1, 2)) == (c == 0):
                w += 1
            else:
                l += 1
    return dict(ref_wins=w, ref_losses=l, ref_draws=d, ref_pairs=played)


@pytest.mark.parametrize("n,pairs,rollouts,batch", [(4, 6, 40, 1), (5, 5, 30, 1), (5, 4, 6, 8), (4, 24, 8, 1)])
def test_pit_matches_oracle_replay(orc, n, pairs, rollouts, batch):
    import tak_amd

    head = tak_amd.HEAD_FC5 if n == 5 else tak_amd.HEAD_CONV
    new = tak_amd.Engine(n, evaluator=tak_amd.EVAL_HASH, max_batch=64, policy_head=head)
    old = tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=64, policy_head=head)
    kw = dict(pairs=pairs, rollouts=rollouts, batch=batch, idle_rollouts=4 if batch == 1 else 1, random_plies=2, komi=2, seed=11, max_plies=60)
    got = tak_amd.pit(new, old, arena_nodes=1 << 16, **kw)
    want = _oracle_pit(orc, n, (orc.EVAL_HASH, orc.EVAL_DUMMY), kw["pairs"], kw["rollouts"], kw["idle_rollouts"], 2, 2, 11, max_plies=60,
                       batch=batch)
    for k in ("wins", "losses", "draws", "plies", "unfinished", "ref_wins", "ref_losses", "ref_draws", "ref_pairs"):
        assert got[k] == want[k], (got, want)
    assert got["wins"] + got["losses"] + got["draws"] + got["unfinished"] == 2 * pairs
    if got["wins"] + got["losses"]:
        assert abs(got["win_rate"] - got["wins"] / (got["wins"] + got["losses"])) < 1e-12
    new.close()
    old.close()


def test_pit_of_two_networks_matches_oracle_replay(orc):
    """The pit as the reference uses it — two NETWORKS (train/src/pit.rs:15-96, main.rs:100) — at a width where both engines run the
    small-batch kernels (2 · 3 pairs · 4 virtual rollouts = 24 leaves: k_tower_split on 128 filters): game for game against the replay
    on the oracle's MCTS, whose two evaluators are two MORE engines that see every leaf batch padded to 300 positions (the
    one-workgroup-per-position tower)."""
    import tak_amd

    import torch_ref

    n, blocks, filters, pairs, rollouts, batch = 5, 2, 128, 3, 10, 4
    nets = [torch_ref.abi_tensors(torch_ref.make_net(n, blocks, filters, "fc5", seed=s)) for s in (1, 2)]

    def engine(tensors, max_batch):
        e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=max_batch)
        e.load_state_dict(tensors)
        return e

    new, old = engine(nets[0], 64), engine(nets[1], 64)
    ev = [engine(nets[0], 320), engine(nets[1], 320)]
    pad = np.stack([orc.new_game(n, half_komi=4)] * 300)

    def padded(e):
        def f(st):
            k = len(st)
            p, v = e.policy_eval(np.concatenate([st, pad[: 300 - k]]))
            return p[:k], v[:k]
        return f

    kw = dict(pairs=pairs, rollouts=rollouts, batch=batch, idle_rollouts=1, random_plies=2, komi=2, seed=5, max_plies=24)
    got = tak_amd.pit(new, old, arena_nodes=1 << 16, **kw)
    want = _oracle_pit(orc, n, (padded(ev[0]), padded(ev[1])), pairs, rollouts, 1, 2, 2, 5, max_plies=24, batch=batch)
    for k in ("wins", "losses", "draws", "plies", "unfinished", "ref_wins", "ref_losses", "ref_draws", "ref_pairs"):
        assert got[k] == want[k], (got, want)
    for e in (new, old, *ev):
        e.close()


def test_identical_networks_split_every_pair(orc):
    import tak_amd

    a = tak_amd.Engine(5, evaluator=tak_amd.EVAL_HASH, max_batch=64)
    b = tak_amd.Engine(5, evaluator=tak_amd.EVAL_HASH, max_batch=64)
    r = tak_amd.pit(a, b, pairs=16, rollouts=12, batch=2, idle_rollouts=12, seed=3, max_plies=80)
    # same evaluator, same budget on both sides → the two games of a pair are the same game with the roles swapped
    assert r["wins"] == r["losses"] and r["draws"] % 2 == 0 and r["unfinished"] % 2 == 0
    with pytest.raises(tak_amd.TgError):
        tak_amd.pit(a, a)
    a.close()
    b.close()


def test_early_exit_tally_of_the_reference():
    """pit.rs:20-23 on hand-made results: with 10 openings the loop breaks before an opening once wins > 11 or losses > 9."""
    white_wins, black_wins = 1, 3  # result codes: road wins
    final = np.array([white_wins, black_wins] * 10)  # the new network wins both games of every opening
    assert _reference_tally(final, 10) == dict(ref_wins=12, ref_losses=0, ref_draws=0, ref_pairs=6)
    final = np.array([black_wins, white_wins] * 10)  # … loses both
    assert _reference_tally(final, 10) == dict(ref_wins=0, ref_losses=10, ref_draws=0, ref_pairs=5)
    final = np.array([white_wins, white_wins] * 10)  # one each: never breaks
    assert _reference_tally(final, 10) == dict(ref_wins=10, ref_losses=10, ref_draws=0, ref_pairs=10)

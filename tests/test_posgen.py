"""CPU checks of the test-input generators (tests/posgen.py, oracle.playouts): the inputs the GPU parity tests of
tests/test_gpu_termination.py feed to both implementations are what they claim to be."""
import numpy as np
import pytest

import posgen


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_steered_games_reach_every_ending(orc, n):
    d = posgen.terminal_mix(orc, n, per_style=400, seed=n)
    final, prev, mv, res = d["final"], d["prev"], d["move"], d["result"]
    counts = np.bincount(res, minlength=7)
    assert counts[0] == 0 and (counts[1:] > 20).all(), counts
    assert np.array_equal(orc.result(n, final), res) and not orc.result(n, prev).any()
    after, status = orc.play(n, prev, mv)
    assert not status.any() and np.array_equal(after, final)
    hk = posgen.header(final, "half_komi").astype(int)
    assert not ((res == 5) & (hk % 2 != 0)).any()  # game.rs:246-255: an odd half-komi never draws on flats


@pytest.mark.parametrize("n", [5, 6])
def test_tall_stack_states_are_positions(orc, n):
    S, _ = posgen.STONES[n]
    sts = posgen.tall_stack_states(n, 300, seed=n, lo=33)
    hs = posgen.heights(sts, n)
    assert hs.max(axis=1).min() >= 33 and hs.max() <= 2 * S + 1
    # reserves + stones on the board = the starting supply, per colour; colour bits stay below the height
    slots = 25 if n <= 5 else 36
    words = sts[:, : 8 * slots].view(np.uint64)[:, : n * n]
    assert not (words >> hs.astype(np.uint64)).any()
    black = np.array([[bin(int(w)).count("1") for w in row] for row in words])
    tops = sts[:, 8 * slots: 8 * slots + n * n] >> 6
    top_black = ((words >> (np.maximum(hs, 1) - 1).astype(np.uint64)) & np.uint64(1)).astype(bool) & (hs > 0)
    caps_b = ((tops == 2) & top_black).sum(axis=1)
    caps_w = ((tops == 2) & ~top_black & (hs > 0)).sum(axis=1)
    assert np.array_equal(posgen.header(sts, "black_stones") + black.sum(axis=1) - caps_b, np.full(len(sts), S))
    assert np.array_equal(posgen.header(sts, "white_stones") + (hs.sum(axis=1) - black.sum(axis=1)) - caps_w, np.full(len(sts), S))
    # the oracle plays every legal move from them without an error, and TPS text round-trips
    om, oc = orc.movegen(n, sts)
    rep = np.repeat(np.arange(len(sts)), oc)
    mv = np.concatenate([om[i, : oc[i]] for i in range(len(sts))])
    _, status = orc.play(n, sts[rep], mv)
    assert not status.any()

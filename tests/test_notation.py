"""Text formats of the product (PTN, TPS, example lines) — host-only, no GPU needed — against the oracle's
restatement and the reference's TPS known-answer string."""
import numpy as np
import pytest

import tak_amd


def test_ptn_and_tps_match_oracle(orc, kats):
    for n in (3, 4, 5, 6):
        sts = orc.random_positions(n, 300, seed=40 + n, max_plies=70 if n >= 5 else 20, half_komi=4)
        mv, cnt = orc.movegen(n, sts)
        for i in range(len(sts)):
            assert tak_amd.format_tps(n, sts[i]) == orc.to_tps(n, sts[i])
            for m in mv[i, : cnt[i]]:
                text = tak_amd.format_move(n, m)
                assert text == orc.format_move(n, m)
                assert tak_amd.parse_move(n, text) == m == orc.parse_move(n, text)
    k = kats["tps"]
    st = orc.from_ptn(k["n"], k["moves"])
    assert tak_amd.format_tps(6, st) == k["tps"]  # tak/tests/tps.rs:19-23
    # explicit-count and run-length forms parse too
    assert tak_amd.parse_move(5, "1d3<") == tak_amd.parse_move(5, "d3<")
    assert np.array_equal(tak_amd.parse_tps(5, "x5/x5/x5/x5/x5 1 1"), orc.new_game(5))


def test_tps_roundtrip_reserves(orc, kats):
    # tak/tests/tps.rs:26-55: Game → Tps → Game keeps board, to_move, ply and reserves on every ply
    for seed in kats["seeds"][:4]:
        st = orc.new_game(5)
        while orc.result(5, st)[0] == 0:
            mv, cnt = orc.movegen(5, st)
            st, status = orc.play(5, st, [mv[0, seed % int(cnt[0])]])
            st = st[0]
            assert status[0] == 0
            back = tak_amd.parse_tps(5, tak_amd.format_tps(5, st))
            a, b = st.copy(), back.copy()
            a[256 - 16 + 9] = 0  # reversible_plies is not part of a TPS
            assert np.array_equal(a, b)


def test_example_line_roundtrip(orc):
    n = 5
    sp = orc.SelfPlay(n, 4, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH, rollouts=12, total_games=6, seed=2)
    for _ in range(300):
        sp.step(1)
        if not sp.states()[1].any():
            break
    hdr, states, moves, visits = sp.drain(10000)
    assert len(hdr) > 10
    seen_results = set()
    for i in range(len(hdr)):
        k = int(hdr["n_moves"][i])
        line = tak_amd.format_example(n, states[i], moves[i, :k], visits[i, :k], float(hdr["result"][i]))
        fields = line.split(";")
        assert len(fields) == 8 and fields[0] == orc.to_tps(n, states[i]) and fields[5] == "4"
        seen_results.add(fields[6])
        st, mv, vs, res = tak_amd.parse_example(n, line)
        want = states[i].copy()
        want[256 - 16 + 9] = 0
        assert np.array_equal(st, want) and np.array_equal(mv, moves[i, :k]) and np.array_equal(vs, visits[i, :k])
        assert res == hdr["result"][i] and np.signbit(res) == np.signbit(hdr["result"][i])
    assert seen_results <= {"1", "-1", "0", "-0"} and len(seen_results) >= 2


def test_example_file_roundtrip(orc, tmp_path):
    """a `.data` file (self_play.rs:98, one Example per line) written from drained examples loads back into the arrays tg_train takes"""
    n = 5
    sp = orc.SelfPlay(n, 3, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH, rollouts=10, total_games=3, seed=9)
    for _ in range(300):
        sp.step(1)
        if not sp.states()[1].any():
            break
    hdr, states, moves, visits = sp.drain(10000)
    path = tmp_path / "games.data"
    with open(path, "w") as f:
        for i in range(len(hdr)):
            k = int(hdr["n_moves"][i])
            f.write(tak_amd.format_example(n, states[i], moves[i, :k], visits[i, :k], float(hdr["result"][i])) + "\n")
    st, nm, mv, vs, res = tak_amd.read_examples(n, path)
    want = states.copy()
    want[:, 256 - 16 + 9] = 0  # reversible_plies is not part of the text format
    assert len(st) == len(hdr) > 5 and np.array_equal(st, want) and np.array_equal(nm, hdr["n_moves"])
    assert np.array_equal(mv, moves) and np.array_equal(vs, visits) and np.array_equal(res, hdr["result"])


def test_parsers_survive_garbage(orc):
    """tg_parse_move / tg_parse_tps / tg_parse_example are host code fed with files from elsewhere: random and mutated input
    must come back as an error (or a valid parse), never crash or write out of bounds"""
    import random

    rnd = random.Random(5)
    n = 5
    sp = orc.SelfPlay(n, 2, head=orc.HEAD_FC5, evaluator=orc.EVAL_HASH, rollouts=8, total_games=2, seed=4)
    for _ in range(200):
        sp.step(1)
        if not sp.states()[1].any():
            break
    hdr, states, moves, visits = sp.drain(1000)
    lines = []
    for i in range(min(len(hdr), 20)):
        k = int(hdr["n_moves"][i])
        lines.append(tak_amd.format_example(n, states[i], moves[i, :k], visits[i, :k], float(hdr["result"][i])))
    alphabet = "abcdefgh12345678SC+-<>x/,;:. \t-0129"
    ok = bad = 0
    for trial in range(6000):
        kind = trial % 3
        if kind == 0:
            text = "".join(rnd.choice(alphabet) for _ in range(rnd.randrange(0, 40)))
        else:
            text = list(rnd.choice(lines))
            for _ in range(rnd.randrange(1, 6)):
                pos = rnd.randrange(len(text))
                op = rnd.randrange(3)
                if op == 0:
                    text[pos] = rnd.choice(alphabet)
                elif op == 1:
                    del text[pos]
                else:
                    text.insert(pos, rnd.choice(alphabet) * rnd.randrange(1, 30))
            text = "".join(text)
        for size in (5, 6):
            for fn in (tak_amd.parse_move, tak_amd.parse_tps, tak_amd.parse_example):
                try:
                    fn(size, text)
                    ok += 1
                except tak_amd.TgError:
                    bad += 1
    assert bad > 1000 and ok > 0
    # very long inputs
    for fn in (tak_amd.parse_move, tak_amd.parse_tps, tak_amd.parse_example):
        with pytest.raises(tak_amd.TgError):
            fn(5, "x5," * 100000)
        with pytest.raises(tak_amd.TgError):
            fn(5, lines[0].split(";")[0] + ";" + "9" * 5000)

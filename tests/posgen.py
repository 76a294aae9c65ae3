"""Test-input generators for the rarely reached corners of the rules (test infrastructure).

* tall_stack_states: packed states built directly (not by play) around one very tall stack, so that the high
  half of the u64 colour word, carries from the top of a 33…62-stone stack, the deep `game_repr` planes and the
  TPS / augmentation paths are exercised.  The states satisfy tg_search_reset's host-side check (heights, colour
  bits, reserves) — the format's invariants, which is what both implementations are specified on.
* with_header: copies of states with header bytes (reversible_plies, half_komi, …) overwritten.
* terminal_mix: whole games from oracle.playouts steered towards every ending of Game::result.
"""
import numpy as np

STONES = {3: (10, 0), 4: (15, 0), 5: (21, 1), 6: (30, 1)}
H_TO_MOVE, H_PLY, H_WS, H_WC, H_BS, H_BC, H_KOMI, H_REV = 1, 2, 4, 5, 6, 7, 8, 9  # byte offsets inside TgHeader


def state_bytes(n):
    return 256 if n <= 5 else 384


def distinct_positions(orc, n, count, seed, max_plies=60, half_komi=4):
    """`count` DIFFERENT ongoing positions (round 6: the full-batch tests used to tile ≈ 2 930 distinct positions to 4096 rows — every
    row was compared, but a quarter of them were repeats): random play-outs are drawn in rounds of 1.5 × what is still missing,
    finished games dropped, duplicates removed by content, until `count` are there."""
    have, seen = [], set()
    rnd = 0
    while len(have) < count:
        batch = orc.random_positions(n, max(256, int(1.5 * (count - len(have))) + 64), seed=seed + 7919 * rnd, max_plies=max_plies, half_komi=half_komi)
        batch = batch[orc.result(n, batch) == 0]
        for st in batch:
            key = st.tobytes()
            if key not in seen:
                seen.add(key)
                have.append(st)
                if len(have) == count:
                    break
        rnd += 1
        assert rnd < 64, "distinct_positions: the generator keeps repeating itself"
    return np.stack(have)


def with_header(states, **fields):
    """copy of `states` with TgHeader fields replaced: to_move, ply, white_stones, …, half_komi, reversible_plies"""
    out = np.array(states, np.uint8, copy=True).reshape(-1, states.shape[-1])
    h = out.shape[1] - 16
    for k, v in fields.items():
        v = np.asarray(v)
        if k == "ply":
            out[:, h + H_PLY] = (v & 0xFF).astype(np.uint8)
            out[:, h + H_PLY + 1] = ((v >> 8) & 0xFF).astype(np.uint8)
        else:
            off = {"to_move": H_TO_MOVE, "white_stones": H_WS, "white_caps": H_WC, "black_stones": H_BS, "black_caps": H_BC,
                   "half_komi": H_KOMI, "reversible_plies": H_REV}[k]
            out[:, h + off] = (v.astype(np.int64) & 0xFF).astype(np.uint8)
    return out


def header(states, field):
    states = np.asarray(states).reshape(-1, states.shape[-1])
    h = states.shape[1] - 16
    if field == "ply":
        return states[:, h + H_PLY].astype(np.int32) | (states[:, h + H_PLY + 1].astype(np.int32) << 8)
    off = {"to_move": H_TO_MOVE, "white_stones": H_WS, "white_caps": H_WC, "black_stones": H_BS, "black_caps": H_BC,
           "half_komi": H_KOMI, "reversible_plies": H_REV}[field]
    v = states[:, h + off]
    return v.astype(np.int8) if field == "half_komi" else v


def heights(states, n):
    slots = 25 if n <= 5 else 36
    return states.reshape(-1, states.shape[-1])[:, 8 * slots: 8 * slots + n * n] & 63


def tall_stack_states(n, count, seed, lo=33, hi=None):
    """`count` packed states, each with one stack of height in [lo, hi] (default hi = every stone of both supplies)
    plus a few small stacks around it; reserves are what is left of the supplies."""
    S, Cc = STONES[n]
    hi = hi or 2 * (S + Cc)
    rng = np.random.default_rng(seed)
    sb, slots, nsq = state_bytes(n), (25 if n <= 5 else 36), n * n
    out = np.zeros((count, sb), np.uint8)
    for i in range(count):
        H = int(rng.integers(lo, hi + 1))
        avail = {0: [S, Cc], 1: [S, Cc]}  # colour → [stones, caps] still in reserve
        st64 = np.zeros(slots, np.uint64)
        meta = np.zeros(slots, np.uint8)

        def build(height, want_top=None):
            """colours bottom→top and the top piece type, drawn from what the supplies still hold"""
            cols = []
            for k in range(height):
                last = k == height - 1
                choices = [c for c in (0, 1) if avail[c][0] > 0 or (last and avail[c][1] > 0)]
                if not choices:
                    break
                c = int(rng.choice(choices))
                top = 0
                if last:
                    kinds = ([0, 1] if avail[c][0] > 0 else []) + ([2] if avail[c][1] > 0 else [])
                    top = int(rng.choice(kinds)) if want_top is None or want_top not in kinds else want_top
                if top == 2:
                    avail[c][1] -= 1
                else:
                    avail[c][0] -= 1
                cols.append(c)
                if last:
                    return cols, top
            return cols, 0

        squares = rng.permutation(nsq)
        cols, top = build(H, want_top=int(rng.integers(0, 3)))
        sq0 = int(squares[0])
        st64[sq0] = sum(np.uint64(c) << np.uint64(k) for k, c in enumerate(cols)) if cols else np.uint64(0)
        meta[sq0] = len(cols) | (top << 6)
        keep_reserves = rng.random() < 0.7  # most states stay Ongoing: both colours keep a few stones in hand
        for sq in squares[1: 1 + int(rng.integers(2, nsq - 1))]:
            if keep_reserves and min(avail[0][0], avail[1][0]) <= 2:
                break
            h = int(rng.integers(0, 5))
            cols, top = build(h)
            if not cols:
                continue
            st64[sq] = sum(np.uint64(c) << np.uint64(k) for k, c in enumerate(cols))
            meta[sq] = len(cols) | (top << 6)
        out[i, : 8 * slots] = st64.view(np.uint8)
        out[i, 8 * slots: 9 * slots] = meta
        to_move = int(rng.integers(0, 2))
        ply = 2 * int(rng.integers(20, 200)) + to_move
        hdr = out[i, sb - 16:]
        hdr[0] = n
        hdr[H_TO_MOVE] = to_move
        hdr[H_PLY], hdr[H_PLY + 1] = ply & 0xFF, ply >> 8
        hdr[H_WS], hdr[H_WC], hdr[H_BS], hdr[H_BC] = avail[0][0], avail[0][1], avail[1][0], avail[1][1]
        hdr[H_KOMI] = np.uint8(int(rng.integers(-5, 7)) & 0xFF)
        hdr[H_REV] = int(rng.integers(0, 50))
    return out


def terminal_mix(orc, n, per_style=3000, seed=1):
    """final states / previous states / last moves / results of whole steered games (every ending of Game::result)"""
    parts = []
    for style in range(5):
        for avoid in (False, True):
            parts.append(orc.playouts(n, per_style, seed=seed * 100 + style * 2 + int(avoid), style=style, avoid_roads=avoid,
                                      max_plies=700))
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

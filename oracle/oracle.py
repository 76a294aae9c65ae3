"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so, the CPU restatement of the reference's hot path
(rules, encoder, move index, MCTS, self-play driver).  Imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product (tak_amd) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

TG_MAX_MOVES = 512
EVAL_CALLBACK, EVAL_DUMMY, EVAL_HASH = 0, 1, 2
HEAD_FC5, HEAD_CONV = 0, 1

RESULT_NAMES = ["Ongoing", "WhiteRoad", "WhiteFlat", "BlackRoad", "BlackFlat", "Draw", "DrawReversible"]

NODE_RECORD = np.dtype(
    [("move", "<u2"), ("n_children", "<u2"), ("visits", "<u4"), ("virtual_visits", "<u4"), ("result", "<u4"),
     ("prior_bits", "<u4"), ("q_bits", "<u4")]
)
EXAMPLE_HEADER = np.dtype([("game_id", "<i4"), ("n_moves", "<i4"), ("result", "<f4"), ("reserved", "<i4")])


class SelfPlayConfig(C.Structure):
    _fields_ = [
        ("rollouts", C.c_int32), ("noise_plies", C.c_int32), ("exploit_plies", C.c_int32),
        ("noise_alpha", C.c_float), ("noise_ratio", C.c_float), ("komi", C.c_int32),
        ("total_games", C.c_int32), ("max_examples", C.c_int32),
    ]


class SelfPlayStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "games_finished", "examples", "expansions", "evals", "plies", "white_wins", "black_wins", "draws", "instant_wins",
        "dropped_examples", "aborted_games", "alive_games")]  # = TgSelfPlayStats (include/takgpu.h)

    def as_dict(self):
        # dropped_examples / aborted_games are always 0 here: the restatement keeps every example in a Vec and has no
        # capacities a game could exceed, as the reference
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


def build():
    """Compile the oracle (gcc).  Building the checker is not using it."""
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        l = C.CDLL(_LIB_PATH)
        l.orc_state_bytes.restype = C.c_size_t
        l.orc_perft.restype = C.c_uint64
        l.orc_perft.argtypes = [C.c_int, C.c_void_p, C.c_int]
        l.orc_search_new.restype = C.c_void_p
        l.orc_search_new.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint64]
        l.orc_selfplay_new.restype = C.c_void_p
        l.orc_selfplay_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                       C.c_uint64, C.c_void_p, C.c_uint32]
        l.orc_random_positions.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        l.orc_playouts.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_seeded_game.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        l.orc_state_hash.restype = C.c_uint64
        l.orc_state_hash.argtypes = [C.c_int, C.c_void_p]
        l.orc_hash_policy.restype = C.c_float
        l.orc_hash_policy.argtypes = [C.c_uint64, C.c_uint32]
        l.orc_hash_eval.restype = C.c_float
        l.orc_hash_eval.argtypes = [C.c_uint64]
        l.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        l.orc_dirichlet.argtypes = [C.c_int, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        for name in ("orc_search_free", "orc_selfplay_free"):
            getattr(l, name).argtypes = [C.c_void_p]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def state_bytes(n):
    return int(lib().orc_state_bytes(n))


def input_channels(n):
    return int(lib().orc_input_channels(n))


def policy_size(n, head):
    return int(lib().orc_policy_size(n, head))


def new_game(n, half_komi=0):
    out = np.zeros(state_bytes(n), np.uint8)
    lib().orc_new_game(n, half_komi, _p(out))
    return out


def from_ptn(n, moves, half_komi=0):
    """Game::from_ptn_moves.  `moves` is a list of PTN strings (or one space-separated string)."""
    if not isinstance(moves, str):
        moves = " ".join(moves)
    out = np.zeros(state_bytes(n), np.uint8)
    err = lib().orc_from_ptn_moves(n, half_komi, moves.encode(), _p(out))
    if err:
        raise ValueError(f"from_ptn_moves failed with code {err}")
    return out


def parse_move(n, ptn):
    code = lib().orc_parse_move(n, ptn.encode())
    if code < 0:
        raise ValueError(f"bad PTN {ptn!r}")
    return code


def format_move(n, code):
    buf = C.create_string_buffer(32)
    lib().orc_format_move(n, int(code), buf, 32)
    return buf.value.decode()


def to_tps(n, state):
    buf = C.create_string_buffer(1024)
    lib().orc_to_tps(n, _p(np.ascontiguousarray(state)), buf, 1024)
    return buf.value.decode()


def _states(n, states):
    states = np.ascontiguousarray(states, np.uint8).reshape(-1, state_bytes(n))
    return states, states.shape[0]


def play(n, states, moves):
    """Batch Game::play (safe_play semantics).  Returns (new_states, status)."""
    states, k = _states(n, states)
    states = states.copy()
    moves = np.ascontiguousarray(moves, np.uint16).reshape(k)
    status = np.zeros(k, np.uint8)
    lib().orc_play(n, k, _p(states), _p(moves), _p(status))
    return states, status


def movegen(n, states):
    states, k = _states(n, states)
    moves = np.zeros((k, TG_MAX_MOVES), np.uint16)
    counts = np.zeros(k, np.int32)
    lib().orc_movegen(n, k, _p(states), _p(moves), _p(counts))
    return moves, counts


def result(n, states):
    states, k = _states(n, states)
    out = np.zeros(k, np.uint8)
    lib().orc_result(n, k, _p(states), _p(out))
    return out


def encode(n, states):
    states, k = _states(n, states)
    out = np.zeros((k, input_channels(n), n, n), np.float32)
    lib().orc_encode(n, k, _p(states), _p(out))
    return out


def move_index(n, moves):
    moves = np.ascontiguousarray(moves, np.uint16).ravel()
    out = np.zeros(moves.size, np.int32)
    lib().orc_move_index(n, moves.size, _p(moves), _p(out))
    return out


def augment(n, head, states, n_moves, moves, visits):
    """Example::to_tensors: (8k states, 8k × P policy targets) in the order of tak/src/symm.rs."""
    states, k = _states(n, states)
    psize = policy_size(n, head)
    n_moves = np.ascontiguousarray(n_moves, np.int32)
    moves = np.ascontiguousarray(moves, np.uint16).reshape(k, TG_MAX_MOVES)
    visits = np.ascontiguousarray(visits, np.uint32).reshape(k, TG_MAX_MOVES)
    out = np.zeros((k * 8, state_bytes(n)), np.uint8)
    pi = np.zeros((k * 8, psize), np.float32)
    lib().orc_augment(n, k, psize, _p(states), _p(n_moves), _p(moves), _p(visits), _p(out), _p(pi))
    return out, pi


def perft(n, state, depth):
    return int(lib().orc_perft(n, _p(np.ascontiguousarray(state, np.uint8)), depth))


def legacy5_table():
    buf = C.create_string_buffer(1 << 16)
    cnt = lib().orc_legacy5_table(buf, 1 << 16)
    strings = buf.value.decode().split("\n")
    assert len(strings) == cnt
    return strings


def random_positions(n, count, seed, max_plies=60, half_komi=0):
    out = np.zeros((count, state_bytes(n)), np.uint8)
    lib().orc_random_positions(n, count, seed, max_plies, half_komi, _p(out))
    return out


def playouts(n, count, seed, style=0, half_komi=127, avoid_roads=False, max_plies=1000):
    """Whole pseudo-random games steered towards particular endings (see orc_playouts).  Returns a dict:
    final / prev states, the last move, the final result and the highest stack of the final position."""
    sb = state_bytes(n)
    final = np.zeros((count, sb), np.uint8)
    prev = np.zeros((count, sb), np.uint8)
    move = np.zeros(count, np.uint16)
    res = np.zeros(count, np.uint8)
    mh = np.zeros(count, np.uint8)
    lib().orc_playouts(n, count, seed, half_komi, style, int(avoid_roads), max_plies, _p(final), _p(prev), _p(move), _p(res), _p(mh))
    return dict(final=final, prev=prev, move=move, result=res, max_height=mh)


def seeded_game(n, seed):
    out = np.zeros(state_bytes(n), np.uint8)
    res = np.zeros(1, np.uint8)
    plies = lib().orc_seeded_game(n, seed, _p(out), _p(res))
    if plies < 0:
        raise ValueError(f"play error {-plies}")
    return out, int(res[0]), plies


def state_hash(n, state):
    return int(lib().orc_state_hash(n, _p(np.ascontiguousarray(state, np.uint8))))


def philox(seed, c0, c1, c2, c3):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(seed, c0, c1, c2, c3, _p(out))
    return out


def dirichlet(k, alpha, seed, slot, generation, ply):
    out = np.zeros(k, np.float32)
    lib().orc_dirichlet(k, alpha, seed, slot, generation, ply, _p(out))
    return out


def _wrap_eval(n, psize, py_eval):
    """py_eval(states[k, bytes]) -> (policy[k, P], eval[k]) wrapped as the C callback."""
    sb = state_bytes(n)

    def cb(ctx, k, st, pol, ev):
        states = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint8)), shape=(k, sb))
        p, e = py_eval(states.copy())
        np.ctypeslib.as_array(C.cast(pol, C.POINTER(C.c_float)), shape=(k, psize))[:] = np.asarray(p, np.float32).reshape(k, psize)
        np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_float)), shape=(k,))[:] = np.asarray(e, np.float32).reshape(k)

    return EVAL_FN(cb)


class Search:
    """Lock-step MCTS over independent trees (reference Node + the loop of self_play.rs:181-210)."""

    def __init__(self, n, head=HEAD_CONV, evaluator=EVAL_DUMMY, py_eval=None, base=500.0, init=4.0, seed=0, batch=1):
        self.n = n
        self.psize = policy_size(n, head)
        self._cb = _wrap_eval(n, self.psize, py_eval) if py_eval is not None else None
        kind = EVAL_CALLBACK if py_eval is not None else evaluator
        self.h = lib().orc_search_new(n, kind, C.cast(self._cb, C.c_void_p) if self._cb else None, None, self.psize, base, init, seed)
        lib().orc_search_batch(C.c_void_p(self.h), int(batch))  # virtual rollouts per tree and iteration (Player's batching)
        self.games = 0

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # (at interpreter shutdown the module's globals may be gone already)
            lib().orc_search_free(self.h)
            self.h = None

    def reset(self, states):
        states, k = _states(self.n, states)
        self.games = k
        lib().orc_search_reset(C.c_void_p(self.h), k, _p(states))

    def run(self, iters, active=None):
        a = np.ascontiguousarray(active, np.uint8) if active is not None else None
        return lib().orc_search_run(C.c_void_p(self.h), iters, _p(a))

    def apply_noise(self, noise, ratio, active=None):
        a = np.ascontiguousarray(active, np.uint8) if active is not None else None
        noise = np.ascontiguousarray(noise, np.float32).reshape(self.games, TG_MAX_MOVES)
        lib().orc_search_apply_noise(C.c_void_p(self.h), _p(noise), C.c_float(ratio), _p(a))

    def apply_dirichlet(self, alpha, ratio, active=None):
        a = np.ascontiguousarray(active, np.uint8) if active is not None else None
        lib().orc_search_apply_dirichlet(C.c_void_p(self.h), C.c_float(alpha), C.c_float(ratio), _p(a))

    def root(self):
        g = self.games
        moves = np.zeros((g, TG_MAX_MOVES), np.uint16)
        visits = np.zeros((g, TG_MAX_MOVES), np.uint32)
        prior = np.zeros((g, TG_MAX_MOVES), np.float32)
        q = np.zeros((g, TG_MAX_MOVES), np.float32)
        counts = np.zeros(g, np.int32)
        rv = np.zeros(g, np.uint32)
        rq = np.zeros(g, np.float32)
        lib().orc_search_root(C.c_void_p(self.h), _p(moves), _p(visits), _p(prior), _p(q), _p(counts), _p(rv), _p(rq))
        return dict(moves=moves, visits=visits, prior=prior, q=q, counts=counts, root_visits=rv, root_q=rq)

    def play(self, moves, active=None):
        a = np.ascontiguousarray(active, np.uint8) if active is not None else None
        moves = np.ascontiguousarray(moves, np.uint16).reshape(self.games)
        return lib().orc_search_play(C.c_void_p(self.h), _p(moves), _p(a))

    def states(self):
        out = np.zeros((self.games, state_bytes(self.n)), np.uint8)
        lib().orc_search_states(C.c_void_p(self.h), _p(out))
        return out

    def dump(self, game, capacity=1 << 20):
        rec = np.zeros(capacity, NODE_RECORD)
        nrec = C.c_size_t(0)
        rc = lib().orc_search_dump(C.c_void_p(self.h), game, _p(rec), C.c_size_t(capacity), C.byref(nrec))
        assert rc == 0, rc
        return rec[: nrec.value].copy()

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().orc_search_counters(C.c_void_p(self.h), C.byref(a), C.byref(b))
        return a.value, b.value


class SelfPlay:
    """self_play_parallel (train/src/self_play.rs:96-262) with runtime parameters."""

    def __init__(self, n, games, head=HEAD_CONV, evaluator=EVAL_DUMMY, py_eval=None, base=500.0, init=4.0, seed=0,
                 rollouts=400, noise_plies=80, exploit_plies=40, noise_alpha=0.2, noise_ratio=0.3, komi=2,
                 total_games=0, slot_base=0):
        self.n = n
        self.games = games
        self.psize = policy_size(n, head)
        self._cb = _wrap_eval(n, self.psize, py_eval) if py_eval is not None else None
        kind = EVAL_CALLBACK if py_eval is not None else evaluator
        cfg = SelfPlayConfig(rollouts, noise_plies, exploit_plies, noise_alpha, noise_ratio, komi, total_games, 0)
        self.h = lib().orc_selfplay_new(n, games, kind, C.cast(self._cb, C.c_void_p) if self._cb else None, None,
                                        self.psize, base, init, seed, C.byref(cfg), slot_base)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_selfplay_free(self.h)
            self.h = None

    def set_threads(self, threads):
        """OpenMP threads for the per-game MCTS phases (results are identical for any count)."""
        lib().orc_selfplay_threads(C.c_void_p(self.h), int(threads))

    def step(self, plies=1):
        return lib().orc_selfplay_step(C.c_void_p(self.h), plies)

    def stats(self):
        s = SelfPlayStats()
        lib().orc_selfplay_stats(C.c_void_p(self.h), C.byref(s))
        return s.as_dict()

    def drain(self, cap=4096):
        hdr = np.zeros(cap, EXAMPLE_HEADER)
        states = np.zeros((cap, state_bytes(self.n)), np.uint8)
        moves = np.zeros((cap, TG_MAX_MOVES), np.uint16)
        visits = np.zeros((cap, TG_MAX_MOVES), np.uint32)
        k = C.c_int32(0)
        lib().orc_selfplay_drain(C.c_void_p(self.h), cap, _p(hdr), _p(states), _p(moves), _p(visits), C.byref(k))
        k = k.value
        return hdr[:k].copy(), states[:k].copy(), moves[:k].copy(), visits[:k].copy()

    def states(self):
        out = np.zeros((self.games, state_bytes(self.n)), np.uint8)
        alive = np.zeros(self.games, np.uint8)
        lib().orc_selfplay_states(C.c_void_p(self.h), _p(out), _p(alive))
        return out, alive

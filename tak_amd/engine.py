"""ctypes binding of libtakgpu.so (C ABI: include/takgpu.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAKGPU_LIB") or os.path.join(_HERE, "libtakgpu.so")  # TAKGPU_LIB: probe builds (scripts/probes)

TG_ABI_VERSION = 5
TG_MAX_MOVES = 512
HEAD_FC5, HEAD_CONV = 0, 1
EVAL_RESNET, EVAL_DUMMY, EVAL_HASH = 0, 1, 2

NODE_RECORD = np.dtype(
    [("move", "<u2"), ("n_children", "<u2"), ("visits", "<u4"), ("virtual_visits", "<u4"), ("result", "<u4"),
     ("prior_bits", "<u4"), ("q_bits", "<u4")]
)
EXAMPLE_HEADER = np.dtype([("game_id", "<i4"), ("n_moves", "<i4"), ("result", "<f4"), ("reserved", "<i4")])


class TgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"takgpu error {code}: {msg}")
        self.code = code


class TgConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "abi_version", "device", "board_size", "res_blocks", "filters", "policy_head", "evaluator", "max_batch")]


class TgSearchConfig(C.Structure):
    _fields_ = [("games", C.c_int32), ("arena_nodes", C.c_int32), ("exploration_base", C.c_float),
                ("exploration_init", C.c_float), ("seed", C.c_uint64), ("slot_base", C.c_uint32), ("batch", C.c_uint32),
                ("visit_limit", C.c_int32), ("reserved", C.c_int32)]


class TgProfile(C.Structure):
    _fields_ = [("conv_launches", C.c_uint64), ("conv_ms", C.c_double), ("forwards", C.c_uint64), ("forward_ms", C.c_double),
                ("conv_rows", C.c_int64), ("conv_flops", C.c_int64), ("conv_flops_executed", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TgSelfPlayConfig(C.Structure):
    _fields_ = [("rollouts", C.c_int32), ("noise_plies", C.c_int32), ("exploit_plies", C.c_int32),
                ("noise_alpha", C.c_float), ("noise_ratio", C.c_float), ("komi", C.c_int32),
                ("total_games", C.c_int32), ("max_examples", C.c_int32), ("max_game_plies", C.c_int32), ("reserved", C.c_int32)]


class TgSelfPlayStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "games_finished", "examples", "expansions", "evals", "plies", "white_wins", "black_wins", "draws", "instant_wins",
        "dropped_examples", "aborted_games", "alive_games")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TgTrainConfig(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("bn_momentum", C.c_float), ("bn_eps", C.c_float), ("chunk_size", C.c_int32),
                ("chunks_in_step", C.c_int32), ("reserved", C.c_int32)]


class TgPitConfig(C.Structure):
    _fields_ = [("pairs", C.c_int32), ("rollouts", C.c_int32), ("idle_rollouts", C.c_int32), ("random_plies", C.c_int32),
                ("komi", C.c_int32), ("max_plies", C.c_int32), ("arena_nodes", C.c_int32), ("batch", C.c_int32),
                ("seed", C.c_uint64)]


class TgPitResult(C.Structure):
    _fields_ = [("wins", C.c_uint32), ("losses", C.c_uint32), ("draws", C.c_uint32), ("unfinished", C.c_uint32),
                ("plies", C.c_uint32), ("reserved", C.c_uint32), ("win_rate", C.c_double),
                ("ref_wins", C.c_uint32), ("ref_losses", C.c_uint32), ("ref_draws", C.c_uint32), ("ref_pairs", C.c_uint32),
                ("ref_win_rate", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class TgCommInfo(C.Structure):
    _fields_ = [("attached", C.c_int32), ("world_size", C.c_int32), ("rank", C.c_int32), ("nccl_count", C.c_int32),
                ("nccl_rank", C.c_int32), ("nccl_version", C.c_int32), ("lib_was_mapped", C.c_int32), ("reserved", C.c_int32),
                ("lib_path", C.c_char * 256)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k not in ("reserved", "lib_path")}
        d["lib_path"] = self.lib_path.decode(errors="replace")
        return d


class TgDeviceInfo(C.Structure):
    _fields_ = [("hip_device", C.c_int32), ("cu_count", C.c_int32), ("clock_khz", C.c_int32), ("reserved", C.c_int32),
                ("total_mem", C.c_uint64), ("pci_bus_id", C.c_char * 32), ("name", C.c_char * 128), ("arch", C.c_char * 64)]

    def as_dict(self):
        return {"hip_device": int(self.hip_device), "pci_bus_id": self.pci_bus_id.decode(errors="replace"),
                "name": self.name.decode(errors="replace"), "arch": self.arch.decode(errors="replace"),
                "cu_count": int(self.cu_count), "clock_khz": int(self.clock_khz), "total_mem": int(self.total_mem)}


# every symbol include/takgpu.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "tg_state_bytes", "tg_engine_create", "tg_engine_destroy", "tg_last_error", "tg_sync", "tg_stream", "tg_device_info", "tg_debug_switches",
    "tg_input_channels", "tg_policy_size", "tg_movegen", "tg_play", "tg_result", "tg_encode", "tg_move_index",
    "tg_perft", "tg_net_set_tensor", "tg_net_init_random", "tg_net_get_tensor", "tg_net_finalize", "tg_net_set_precision", "tg_policy_eval", "tg_forward_mcts", "tg_policy_eval_dev",
    "tg_search_create", "tg_search_reset", "tg_search_run", "tg_search_apply_dirichlet", "tg_search_apply_noise",
    "tg_search_root", "tg_search_play", "tg_search_states", "tg_search_dump", "tg_search_counters", "tg_search_pool",
    "tg_selfplay_create", "tg_selfplay_step", "tg_selfplay_stats", "tg_selfplay_drain",
    "tg_profile_enable", "tg_profile_read", "tg_board_pass_bench",
    "tg_augment_examples",
    "tg_train_create", "tg_train_chunk", "tg_train", "tg_train_step", "tg_train_forward", "tg_train_get_tensor",
    "tg_train_get_grad", "tg_train_debug_capture", "tg_train_debug_read", "tg_train_commit", "tg_comm_unique_id", "tg_train_comm_init", "tg_train_set_allreduce",
    "tg_train_grad_buffer", "tg_train_comm_stats", "tg_train_comm_info", "tg_train_comm_preflight", "tg_train_order", "tg_pit",
    "tg_format_move", "tg_parse_move", "tg_format_tps", "tg_parse_tps", "tg_format_example", "tg_parse_example",
]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def device_to_host(d_ptr, count, stream=None):
    """`count` floats of device memory as a numpy array (after synchronising `stream`); HIP runtime through the
    library's own dependency (no torch involved)."""
    lib = load_library()
    if stream is not None:
        rc = lib.hipStreamSynchronize(C.c_void_p(stream))
        assert rc == 0, f"hipStreamSynchronize: {rc}"
    out = np.empty(count, np.float32)
    rc = lib.hipMemcpy(_p(out), C.c_void_p(d_ptr), C.c_size_t(count * 4), 2)  # hipMemcpyDeviceToHost
    assert rc == 0, f"hipMemcpy D2H: {rc}"
    return out


def host_to_device(d_ptr, array):
    a = np.ascontiguousarray(array, np.float32)
    rc = load_library().hipMemcpy(C.c_void_p(d_ptr), _p(a), C.c_size_t(a.size * 4), 1)  # hipMemcpyHostToDevice
    assert rc == 0, f"hipMemcpy H2D: {rc}"


def build_library():
    """Compile the HIP sources for gfx950 into tak_amd/libtakgpu.so (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], check=True)


_lib = None


def load_library():
    """Load libtakgpu.so.  Raises (never falls back) if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: build it with tak_amd.build_library() / __graft_entry__.build()")
        l = C.CDLL(LIB_PATH)
        l.tg_state_bytes.restype = C.c_size_t
        l.tg_last_error.restype = C.c_char_p
        l.tg_stream.restype = C.c_void_p
        l.tg_stream.argtypes = [C.c_void_p]
        l.tg_engine_create.argtypes = [C.c_void_p, C.c_void_p]
        l.tg_engine_destroy.argtypes = [C.c_void_p]
        l.tg_engine_destroy.restype = None
        for name in ABI_SYMBOLS:
            getattr(l, name)
        _lib = l
    return _lib


def state_bytes(n):
    return 256 if n <= 5 else 384


def input_channels(n):
    return int(load_library().tg_input_channels(n))


def policy_size(n, head):
    return int(load_library().tg_policy_size(n, head))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _text_check(rc):
    if rc < 0:
        raise TgError(rc, load_library().tg_last_error().decode())
    return rc


# ---- text formats (host only; no GPU needed): PTN, TPS, example lines of alpha-tak/src/example.rs:81-133 ----
def format_move(n, code):
    buf = C.create_string_buffer(32)
    _text_check(load_library().tg_format_move(n, C.c_uint16(int(code)), buf, C.c_size_t(32)))
    return buf.value.decode()


def parse_move(n, text):
    out = C.c_uint16(0)
    _text_check(load_library().tg_parse_move(n, text.encode(), C.byref(out)))
    return out.value


def format_tps(n, state):
    st = np.ascontiguousarray(state, np.uint8)
    buf = C.create_string_buffer(2048)
    _text_check(load_library().tg_format_tps(n, _p(st), buf, C.c_size_t(2048)))
    return buf.value.decode()


def parse_tps(n, text):
    st = np.zeros(state_bytes(n), np.uint8)
    _text_check(load_library().tg_parse_tps(n, text.encode(), _p(st)))
    return st


def format_example(n, state, moves, visits, result):
    st = np.ascontiguousarray(state, np.uint8)
    mv = np.ascontiguousarray(moves, np.uint16)
    vs = np.ascontiguousarray(visits, np.uint32)
    buf = C.create_string_buffer(1 << 14)
    _text_check(load_library().tg_format_example(n, _p(st), int(mv.size), _p(mv), _p(vs), C.c_float(float(result)), buf, C.c_size_t(1 << 14)))
    return buf.value.decode()


def parse_example(n, line):
    st = np.zeros(state_bytes(n), np.uint8)
    mv = np.zeros(TG_MAX_MOVES, np.uint16)
    vs = np.zeros(TG_MAX_MOVES, np.uint32)
    k = C.c_int32(0)
    res = C.c_float(0)
    _text_check(load_library().tg_parse_example(n, line.encode(), _p(st), TG_MAX_MOVES, _p(mv), _p(vs), C.byref(k), C.byref(res)))
    return st, mv[: k.value].copy(), vs[: k.value].copy(), res.value


def read_examples(n, paths):
    """Load `.data` example files as train/src/main.rs:69-80 does (one Example per line, alpha-tak/src/example.rs:100-133)
    → (states [k, bytes], n_moves [k], moves [k, TG_MAX_MOVES], visits [k, TG_MAX_MOVES], results [k]): the layout tg_train takes."""
    if isinstance(paths, (str, bytes, os.PathLike)):
        paths = [paths]
    rows = []
    for path in paths:
        with open(path) as f:
            for line in f.read().split("\n"):
                if line.strip():
                    rows.append(parse_example(n, line))
    k = len(rows)
    states = np.zeros((k, state_bytes(n)), np.uint8)
    n_moves = np.zeros(k, np.int32)
    moves = np.zeros((k, TG_MAX_MOVES), np.uint16)
    visits = np.zeros((k, TG_MAX_MOVES), np.uint32)
    results = np.zeros(k, np.float32)
    for i, (st, mv, vs, res) in enumerate(rows):
        states[i], n_moves[i], results[i] = st, len(mv), res
        moves[i, : len(mv)], visits[i, : len(mv)] = mv, vs
    return states, n_moves, moves, visits, results


def comm_unique_id():
    """ncclGetUniqueId on this rank (128 bytes); broadcast it to the other ranks with any host transport."""
    uid = np.zeros(128, np.uint8)
    rc = load_library().tg_comm_unique_id(_p(uid))
    if rc:
        raise TgError(rc, load_library().tg_last_error().decode())
    return uid.tobytes()


def debug_switches():
    """the TG_* A/B switches that are ON in this process's environment, as the library reads them → ["NAME=value", …];
    [] on a measured run (tg_debug_switches; needs no GPU)"""
    buf = C.create_string_buffer(4096)
    n = load_library().tg_debug_switches(buf, C.c_size_t(4096))
    out = buf.value.decode().split()
    assert n == len(out), (n, out)
    return out


def train_order(seed, n):
    """the permutation tg_train(seed) visits n examples in (tg_train_order)"""
    order = np.zeros(n, np.int32)
    _text_check(load_library().tg_train_order(C.c_uint64(seed), n, _p(order)))
    return order


def pit(new, old, pairs=128, rollouts=50, batch=16, idle_rollouts=1, random_plies=2, komi=2, max_plies=0, arena_nodes=0, seed=0):
    """`pit(new, old)` of train/src/pit.rs on two engines (one per weight set) → dict(wins, losses, draws, win_rate, …; the
    ref_* entries are the counts with the reference's early exit, pit.rs:20-23, applied);
    `rollouts` iterations of `batch` virtual rollouts per move (ROLLOUTS × BATCH_SIZE); 2·pairs·batch ≤ max_batch"""
    cfg = TgPitConfig(pairs, rollouts, idle_rollouts, random_plies, komi, max_plies, arena_nodes, batch, seed)
    res = TgPitResult()
    rc = load_library().tg_pit(new.h, old.h, C.byref(cfg), C.byref(res))
    if rc:
        raise TgError(rc, load_library().tg_last_error().decode())
    return res.as_dict()


def tensor_shapes(n, res_blocks, filters, policy_head):
    """{ABI tensor name: shape} in the creation order of net5.rs:29-62 / net6.rs:29-57 (tch layouts, include/takgpu.h)."""
    F, cin, P = filters, input_channels(n), policy_size(n, policy_head)
    out = {}

    def conv(name, o, i):
        out[name + ".weight"], out[name + ".bias"] = (o, i, 3, 3), (o,)

    def bn(name):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out[f"{name}.{leaf}"] = (F,)

    conv("conv0", F, cin)
    bn("bn0")
    for i in range(res_blocks):
        conv(f"res{i}.conv1", F, F)
        conv(f"res{i}.conv2", F, F)
        bn(f"res{i}.bn1")
        bn(f"res{i}.bn2")
    if policy_head == HEAD_FC5:
        out["policy.weight"], out["policy.bias"] = (P, F * n * n), (P,)
    else:
        conv("policy", P // (n * n), F)
    out["value.weight"], out["value.bias"] = (1, F * n * n), (1,)
    return out


def _mask(active):
    return np.ascontiguousarray(active, np.uint8) if active is not None else None


class Engine:
    """One engine per GPU (reference: one `NET` + the statics of alpha-tak/src/lib.rs:21-23)."""

    def __init__(self, board_size=5, res_blocks=6, filters=64, policy_head=None, evaluator=EVAL_RESNET, max_batch=4096,
                 device=0):
        self.lib = load_library()
        if policy_head is None:
            policy_head = HEAD_FC5 if board_size == 5 else HEAD_CONV
        self.n = board_size
        self.head = policy_head
        self.max_batch = max_batch
        self.cfg = TgConfig(TG_ABI_VERSION, device, board_size, res_blocks, filters, policy_head, evaluator, max_batch)
        self.h = C.c_void_p(None)
        self._check(self.lib.tg_engine_create(C.byref(self.cfg), C.byref(self.h)))
        self.sb = state_bytes(board_size)
        self.cin = input_channels(board_size)
        self.psize = policy_size(board_size, policy_head)
        self.games = 0

    def device_info(self):
        """the card this engine runs on as the HIP runtime names it: {hip_device, pci_bus_id, name, arch, cu_count, …}"""
        info = TgDeviceInfo()
        self._check(self.lib.tg_device_info(self.h, C.byref(info)))
        return info.as_dict()

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.tg_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise TgError(rc, self.lib.tg_last_error().decode())

    def _states(self, states):
        states = np.ascontiguousarray(states, np.uint8).reshape(-1, self.sb)
        return states, states.shape[0]

    def sync(self):
        self._check(self.lib.tg_sync(self.h))

    @property
    def stream(self):
        return self.lib.tg_stream(self.h)

    # ---- Game::{possible_moves, play, result}, game_repr, move_index -------------------------------
    def movegen(self, states):
        states, k = self._states(states)
        moves = np.zeros((k, TG_MAX_MOVES), np.uint16)
        counts = np.zeros(k, np.int32)
        self._check(self.lib.tg_movegen(self.h, k, _p(states), _p(moves), _p(counts)))
        return moves, counts

    def play(self, states, moves, check=False):
        states, k = self._states(states)
        states = states.copy()
        moves = np.ascontiguousarray(moves, np.uint16).reshape(k)
        status = np.zeros(k, np.uint8)
        rc = self.lib.tg_play(self.h, k, _p(states), _p(moves), _p(status))
        if rc != 0 and (check or rc != -8):
            self._check(rc)
        return states, status

    def result(self, states):
        states, k = self._states(states)
        out = np.zeros(k, np.uint8)
        self._check(self.lib.tg_result(self.h, k, _p(states), _p(out)))
        return out

    def encode(self, states):
        states, k = self._states(states)
        out = np.zeros((k, self.cin, self.n, self.n), np.float32)
        self._check(self.lib.tg_encode(self.h, k, _p(states), _p(out)))
        return out

    def move_index(self, moves):
        moves = np.ascontiguousarray(moves, np.uint16).ravel()
        out = np.zeros(moves.size, np.int32)
        self._check(self.lib.tg_move_index(self.h, moves.size, _p(moves), _p(out)))
        return out

    def augment_examples(self, states, n_moves, moves, visits):
        """Example::to_tensors: 8 symmetric states and policy targets per example."""
        states, k = self._states(states)
        n_moves = np.ascontiguousarray(n_moves, np.int32)
        moves = np.ascontiguousarray(moves, np.uint16).reshape(k, TG_MAX_MOVES)
        visits = np.ascontiguousarray(visits, np.uint32).reshape(k, TG_MAX_MOVES)
        out = np.zeros((k * 8, self.sb), np.uint8)
        pi = np.zeros((k * 8, self.psize), np.float32)
        self._check(self.lib.tg_augment_examples(self.h, k, _p(states), _p(n_moves), _p(moves), _p(visits), _p(out), _p(pi)))
        return out, pi

    # ---- training step (Network::train / train_inner, alpha-tak/src/model/network.rs:37-97) ----
    def train_create(self, learning_rate=1e-4, weight_decay=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, bn_momentum=0.1,
                     bn_eps=1e-5, chunk_size=500, chunks_in_step=20):
        cfg = TgTrainConfig(learning_rate, weight_decay, beta1, beta2, eps, bn_momentum, bn_eps, chunk_size, chunks_in_step, 0)
        self._check(self.lib.tg_train_create(self.h, C.byref(cfg)))
        self.chunk_size = chunk_size

    def _examples(self, states, n_moves, moves, visits, results):
        states, k = self._states(states)
        n_moves = np.ascontiguousarray(n_moves, np.int32)
        moves = np.ascontiguousarray(moves, np.uint16).reshape(k, TG_MAX_MOVES)
        visits = np.ascontiguousarray(visits, np.uint32).reshape(k, TG_MAX_MOVES)
        results = np.ascontiguousarray(results, np.float32).reshape(k)
        return states, k, n_moves, moves, visits, results

    def train_chunk(self, states, n_moves, moves, visits, results):
        """train_inner on one chunk → (loss_p, loss_z, stepped)"""
        states, k, n_moves, moves, visits, results = self._examples(states, n_moves, moves, visits, results)
        lp, lz, did = C.c_float(0), C.c_float(0), C.c_int32(0)
        self._check(self.lib.tg_train_chunk(self.h, k, _p(states), _p(n_moves), _p(moves), _p(visits), _p(results),
                                            C.byref(lp), C.byref(lz), C.byref(did)))
        return lp.value, lz.value, bool(did.value)

    def train(self, states, n_moves, moves, visits, results, seed=0):
        """Network::train over a set of examples → (mean loss_p, mean loss_z, optimiser steps)"""
        states, k, n_moves, moves, visits, results = self._examples(states, n_moves, moves, visits, results)
        lp, lz, steps = C.c_float(0), C.c_float(0), C.c_int32(0)
        self._check(self.lib.tg_train(self.h, k, _p(states), _p(n_moves), _p(moves), _p(visits), _p(results), C.c_uint64(seed),
                                      C.byref(lp), C.byref(lz), C.byref(steps)))
        return lp.value, lz.value, steps.value

    def train_step(self):
        self._check(self.lib.tg_train_step(self.h))

    def train_forward(self, states):
        """forward_training: (log_softmax policy, eval) with BatchNorm on the batch statistics"""
        states, k = self._states(states)
        logp = np.zeros((k, self.psize), np.float32)
        ev = np.zeros(k, np.float32)
        self._check(self.lib.tg_train_forward(self.h, k, _p(states), _p(logp), _p(ev)))
        return logp, ev

    def train_get_tensor(self, name, shape):
        out = np.zeros(shape, np.float32)
        self._check(self.lib.tg_train_get_tensor(self.h, name.encode(), _p(out), C.c_size_t(out.size)))
        return out

    def train_get_grad(self, name, shape):
        out = np.zeros(shape, np.float32)
        self._check(self.lib.tg_train_get_grad(self.h, name.encode(), _p(out), C.c_size_t(out.size)))
        return out

    def train_debug_capture(self, layer):
        """keep dy / dz / dx of conv layer `layer` in the backward pass of the next chunks (layer < 0: stop)"""
        self._check(self.lib.tg_train_debug_capture(self.h, int(layer)))

    def train_debug_read(self, what, layer, shape):
        """an intermediate tensor of the last training chunk / forward (include/takgpu.h tg_train_debug_read)"""
        out = np.zeros(shape, np.float32)
        self._check(self.lib.tg_train_debug_read(self.h, what.encode(), int(layer), _p(out), C.c_size_t(out.size)))
        return out

    def train_commit(self):
        self._check(self.lib.tg_train_commit(self.h))

    def train_set_allreduce(self, fn, world):
        """Route the gradient / BatchNorm-statistics reduction through fn(d_ptr, count, stream) -> 0 | error instead of
        RCCL (tg_train_set_allreduce).  fn must leave the SUM over all `world` ranks in the device buffer.  None removes it."""
        if fn is None:
            self._allreduce_cb = None
            self._check(self.lib.tg_train_set_allreduce(self.h, None, None, 1))
            return

        def cb(ctx, d_buf, count, stream):
            try:
                return int(fn(d_buf, count, stream) or 0)
            except Exception:  # an exception must not unwind through the C caller
                import traceback

                traceback.print_exc()
                return 1

        self._allreduce_cb = ALLREDUCE_FN(cb)  # kept alive as long as the engine uses it
        self._check(self.lib.tg_train_set_allreduce(self.h, self._allreduce_cb, None, int(world)))

    def train_grad_buffer(self):
        """(device address, float count) of the flat gradient buffer"""
        ptr, cnt = C.c_void_p(0), C.c_size_t(0)
        self._check(self.lib.tg_train_grad_buffer(self.h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def train_comm_stats(self):
        """(total milliseconds, count) of the gradient all-reduces issued by optimiser steps so far (HIP events on the engine stream)"""
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._check(self.lib.tg_train_comm_stats(self.h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def train_comm_info(self):
        """what is attached to the optimiser step's reduction, as the library sees it: ncclCommCount / ncclCommUserRank of the
        communicator, ncclGetVersion, the file ncclAllReduce was bound from, whether that copy was already mapped"""
        info = TgCommInfo()
        self._check(self.lib.tg_train_comm_info(self.h, C.byref(info)))
        return info.as_dict()

    def train_comm_preflight(self):
        """one float summed over the ranks through the optimiser step's reduction, before any training work is enqueued →
        wall-clock milliseconds of the round trip (0.0 on a single-rank trainer); raises if the sum is not world_size"""
        ms = C.c_double(0)
        self._check(self.lib.tg_train_comm_preflight(self.h, C.byref(ms)))
        return ms.value

    def train_comm_init(self, rank, world, unique_id):
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        assert uid.size == 128
        self._check(self.lib.tg_train_comm_init(self.h, rank, world, _p(uid)))

    def perft(self, states, depth):
        states, k = self._states(states)
        out = np.zeros(k, np.uint64)
        self._check(self.lib.tg_perft(self.h, k, _p(states), depth, _p(out)))
        return out

    # ---- Network<N> -------------------------------------------------------------------------------
    def set_tensor(self, name, array):
        a = np.ascontiguousarray(array, np.float32)
        self._check(self.lib.tg_net_set_tensor(self.h, name.encode(), _p(a), C.c_size_t(a.size)))

    def set_precision(self, precision):
        """"f32" (exact, default) or "bf16x3" (split-bf16 tower); takes effect at the next load_state_dict / finalize"""
        self._check(self.lib.tg_net_set_precision(self.h, {"f32": 0, "bf16x3": 1}[precision]))

    def init_random(self, seed=0, finalize=True):
        """Network::default(): tch's default initialisers drawn from Philox(seed) (tg_net_init_random)."""
        self._check(self.lib.tg_net_init_random(self.h, C.c_uint64(seed)))
        if finalize:
            self._check(self.lib.tg_net_finalize(self.h))

    def get_tensor(self, name, shape):
        """the tensor as last set / initialised / committed (tg_net_get_tensor) — what Network::save writes"""
        out = np.zeros(shape, np.float32)
        self._check(self.lib.tg_net_get_tensor(self.h, name.encode(), _p(out), C.c_size_t(out.size)))
        return out

    def state_dict(self):
        """{name: array} of every tensor as last set / initialised / committed (Network::save's content)"""
        return {k: self.get_tensor(k, shp) for k, shp in tensor_shapes(self.n, self.cfg.res_blocks, self.cfg.filters, self.head).items()}

    def load_state_dict(self, tensors):
        """tensors: {name: array} with the names of include/takgpu.h (tch layouts)."""
        for k, v in tensors.items():
            self.set_tensor(k, np.asarray(v, np.float32))
        self._check(self.lib.tg_net_finalize(self.h))

    def policy_eval(self, states):
        """Network::policy_eval: states → (policy [k, P] softmax, eval [k] tanh)."""
        states, k = self._states(states)
        policy = np.zeros((k, self.psize), np.float32)
        ev = np.zeros(k, np.float32)
        self._check(self.lib.tg_policy_eval(self.h, k, _p(states), _p(policy), _p(ev)))
        return policy, ev

    def forward_mcts(self, planes):
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1, self.cin, self.n, self.n)
        k = planes.shape[0]
        policy = np.zeros((k, self.psize), np.float32)
        ev = np.zeros(k, np.float32)
        self._check(self.lib.tg_forward_mcts(self.h, k, _p(planes), _p(policy), _p(ev)))
        return policy, ev

    def policy_eval_dev(self, n, d_states, d_policy, d_eval):
        self._check(self.lib.tg_policy_eval_dev(self.h, n, C.c_void_p(d_states), C.c_void_p(d_policy), C.c_void_p(d_eval)))

    # ---- Node / search ----------------------------------------------------------------------------
    def search_create(self, games, arena_nodes=1 << 16, base=500.0, init=4.0, seed=0, slot_base=0, batch=1, visit_limit=0):
        """batch: virtual rollouts per tree and iteration (Player's batching); games * batch <= max_batch"""
        cfg = TgSearchConfig(games, arena_nodes, base, init, seed, slot_base, batch, visit_limit, 0)
        self._check(self.lib.tg_search_create(self.h, C.byref(cfg)))
        self.games = games

    def search_reset(self, states):
        states, k = self._states(states)
        assert k == self.games
        self._check(self.lib.tg_search_reset(self.h, _p(states)))

    def search_run(self, iters, active=None):
        self._check(self.lib.tg_search_run(self.h, iters, _p(_mask(active))))

    def search_apply_dirichlet(self, alpha, ratio, active=None):
        self._check(self.lib.tg_search_apply_dirichlet(self.h, C.c_float(alpha), C.c_float(ratio), _p(_mask(active))))

    def search_apply_noise(self, noise, ratio, active=None):
        noise = np.ascontiguousarray(noise, np.float32).reshape(self.games, TG_MAX_MOVES)
        self._check(self.lib.tg_search_apply_noise(self.h, _p(noise), C.c_float(ratio), _p(_mask(active))))

    def search_root(self):
        g = self.games
        moves = np.zeros((g, TG_MAX_MOVES), np.uint16)
        visits = np.zeros((g, TG_MAX_MOVES), np.uint32)
        prior = np.zeros((g, TG_MAX_MOVES), np.float32)
        q = np.zeros((g, TG_MAX_MOVES), np.float32)
        counts = np.zeros(g, np.int32)
        rv = np.zeros(g, np.uint32)
        rq = np.zeros(g, np.float32)
        self._check(self.lib.tg_search_root(self.h, _p(moves), _p(visits), _p(prior), _p(q), _p(counts), _p(rv), _p(rq)))
        return dict(moves=moves, visits=visits, prior=prior, q=q, counts=counts, root_visits=rv, root_q=rq)

    def search_play(self, moves, active=None):
        moves = np.ascontiguousarray(moves, np.uint16).reshape(self.games)
        self._check(self.lib.tg_search_play(self.h, _p(moves), _p(_mask(active))))

    def search_states(self):
        out = np.zeros((self.games, self.sb), np.uint8)
        self._check(self.lib.tg_search_states(self.h, _p(out)))
        return out

    def search_dump(self, game, capacity=1 << 20):
        rec = np.zeros(capacity, NODE_RECORD)
        nrec = C.c_size_t(0)
        self._check(self.lib.tg_search_dump(self.h, game, _p(rec), C.c_size_t(capacity), C.byref(nrec)))
        return rec[: nrec.value].copy()

    def search_counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.tg_search_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- measurement hooks ----------------------------------------------------------------------
    def search_pool(self):
        """node-pool occupancy: dict(total, in_use, peak) in nodes"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.tg_search_pool(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"total": a.value, "in_use": b.value, "peak": c.value}

    def profile_enable(self, sample_every):
        self._check(self.lib.tg_profile_enable(self.h, sample_every))

    def profile_read(self):
        p = TgProfile()
        self._check(self.lib.tg_profile_read(self.h, C.byref(p)))
        return p.as_dict()

    def board_pass_bench(self, states, moves, reps=10):
        """Fused play → result → movegen-count → encode pass on device-resident inputs (micro-benchmark)."""
        states, k = self._states(states)
        moves = np.ascontiguousarray(moves, np.uint16).reshape(k)
        ms = C.c_double(0)
        out_states = np.zeros_like(states)
        res = np.zeros(k, np.uint8)
        cnt = np.zeros(k, np.int32)
        self._check(self.lib.tg_board_pass_bench(self.h, k, _p(states), _p(moves), reps, C.byref(ms), _p(out_states), _p(res), _p(cnt)))
        return ms.value, out_states, res, cnt

    # ---- self_play_parallel ------------------------------------------------------------------------
    def selfplay_create(self, games, arena_nodes=0, base=500.0, init=4.0, seed=0, rollouts=400, noise_plies=80,
                        exploit_plies=40, noise_alpha=0.2, noise_ratio=0.3, komi=2, total_games=0, max_examples=1 << 16,
                        slot_base=0, max_game_plies=0, visit_limit=0):
        scfg = TgSearchConfig(games, arena_nodes, base, init, seed, slot_base, 0, visit_limit, 0)
        cfg = TgSelfPlayConfig(rollouts, noise_plies, exploit_plies, noise_alpha, noise_ratio, komi, total_games, max_examples,
                               max_game_plies, 0)
        self._check(self.lib.tg_selfplay_create(self.h, C.byref(scfg), C.byref(cfg)))
        self.games = games

    def selfplay_step(self, plies=1):
        self._check(self.lib.tg_selfplay_step(self.h, plies))

    def selfplay_stats(self):
        s = TgSelfPlayStats()
        self._check(self.lib.tg_selfplay_stats(self.h, C.byref(s)))
        return s.as_dict()

    def write_examples(self, path, cap=1 << 16):
        """Drain finished examples and append them to `path` in the reference's `.data` text format."""
        hdr, states, moves, visits = self.selfplay_drain(cap)
        with open(path, "a") as f:
            for i in range(len(hdr)):
                k = int(hdr["n_moves"][i])
                f.write(format_example(self.n, states[i], moves[i, :k], visits[i, :k], float(hdr["result"][i])) + "\n")
        return len(hdr)

    def selfplay_drain(self, cap=4096):
        hdr = np.zeros(cap, EXAMPLE_HEADER)
        states = np.zeros((cap, self.sb), np.uint8)
        moves = np.zeros((cap, TG_MAX_MOVES), np.uint16)
        visits = np.zeros((cap, TG_MAX_MOVES), np.uint32)
        k = C.c_int32(0)
        self._check(self.lib.tg_selfplay_drain(self.h, cap, _p(hdr), _p(states), _p(moves), _p(visits), C.byref(k)))
        k = k.value
        return hdr[:k].copy(), states[:k].copy(), moves[:k].copy(), visits[:k].copy()

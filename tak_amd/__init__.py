"""tak_amd — MI355X-native batched Tak self-play engine (host-side binding of libtakgpu.so).

The compute lives in hand-written HIP kernels behind the C ABI of include/takgpu.h; this package is
the thin Python mirror of the reference's operator surface (`Network::policy_eval`, `Node`,
`self_play_parallel`, `Game::{play, possible_moves, result}`) used by tests and bench.py.
There is no CPU fallback: importing works anywhere, but creating an Engine needs a GPU and the
built library, and fails loudly otherwise.
"""
from .engine import (  # noqa: F401
    Engine, TgError, HEAD_FC5, HEAD_CONV, EVAL_RESNET, EVAL_DUMMY, EVAL_HASH, TG_MAX_MOVES,
    state_bytes, input_channels, policy_size, load_library, build_library, LIB_PATH,
    format_move, parse_move, format_tps, parse_tps, format_example, parse_example, read_examples, comm_unique_id, pit, tensor_shapes,
    debug_switches, train_order,
)
from .player import Player  # noqa: E402,F401

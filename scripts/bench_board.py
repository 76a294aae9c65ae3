#!/usr/bin/env python3
"""Board-path micro-benchmark: 2^20 positions reached by random play, one fused
play → result → movegen-count → encode pass with inputs resident in HBM.  Reports achieved GB/s against the
HBM roofline using the algorithmic bytes per position (state in + state out + f32 planes).

The positions are produced by the engine's own batch operators (tg_movegen / tg_play through the C ABI); the bit-exact
comparison of this pass against the CPU checker lives in tests/test_gpu_fullsize.py."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=5)
ap.add_argument("--positions", type=int, default=1 << 20)
ap.add_argument("--distinct", type=int, default=1 << 15)
ap.add_argument("--plies", type=int, default=40, help="random plies played to reach the benchmark positions")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()

import tak_amd

n = args.board
eng = tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=4096)
rng = np.random.default_rng(1)
sb = tak_amd.state_bytes(n)
# start positions (Game::with_komi(2)): empty board, full reserves
stones, caps = {3: (10, 0), 4: (15, 0), 5: (21, 1), 6: (30, 1)}[n]
start = np.zeros(sb, np.uint8)
start[sb - 16 + 0] = n
start[sb - 16 + 4 : sb - 16 + 8] = [stones, caps, stones, caps]
start[sb - 16 + 8] = 4
states = np.tile(start, (args.distinct, 1))
target = rng.integers(2, args.plies + 1, args.distinct)  # a mix of game lengths
for ply in range(args.plies):
    res = eng.result(states)
    mv, cnt = eng.movegen(states)
    go = (res == 0) & (ply < target) & (cnt > 0)
    pick = (rng.random(len(states)) * np.maximum(cnt, 1)).astype(np.int64)
    moves = mv[np.arange(len(states)), pick]
    idx = np.nonzero(go)[0]
    if len(idx) == 0:
        break
    states[idx] = eng.play(states[idx], moves[idx])[0]
keep = eng.result(states) == 0
base = states[keep]
mv, cnt = eng.movegen(base)
pick = (rng.random(len(base)) * cnt).astype(np.int64)
moves = mv[np.arange(len(base)), pick]
reps_tile = (args.positions + len(base) - 1) // len(base)
states = np.tile(base, (reps_tile, 1))[: args.positions]
moves_all = np.tile(moves, reps_tile)[: args.positions]

ms, out_states, res, counts = eng.board_pass_bench(states, moves_all, reps=args.reps)
planes_bytes = tak_amd.input_channels(n) * n * n * 4
alg = 2 * sb + planes_bytes
gbs = args.positions * alg / (ms * 1e-3) / 1e9
# HBM bytes per launch by the memory-side counters: a separate `rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ` pass over this script,
# summarised by scripts/pmc_traffic.py (scripts/collect_evidence.sh) and committed under profiles/; null if that file is absent
traffic = None
pmc = os.path.join(ROOT, "profiles", f"pmc_board_pass_{n}x{n}.json")
if os.path.exists(pmc) and args.positions == 1 << 20:
    try:
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
    except Exception:
        traffic = None
print(json.dumps({
    "bench": "board_pass", "board": n, "positions": args.positions, "distinct_positions": int(len(base)), "avg_ms": ms,
    "algorithmic_bytes_per_position": alg, "positions_per_s": args.positions / (ms * 1e-3),
    "mean_legal_moves": float(cnt.mean()),
    "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": traffic,
                 "traffic_gbs": None if traffic is None else traffic / (ms * 1e-3) / 1e9,
                 "note": "planes are written as the reference tensor has them, C_in f32 channels per square (no padding): the store "
                         "bytes are the algorithmic ones; achievable HBM is ~6.3 TB/s"},
}))
eng.close()

#!/usr/bin/env python3
"""Soak run of the GPU self-play driver at full width with the hash evaluator (no network cost): many plies,
games finishing and recycling, examples drained — checks that no device error flag (arena / staging / queue
overflow, NaN, illegal move) is ever raised and prints game statistics."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import tak_amd

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=5)
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--rollouts", type=int, default=400)
ap.add_argument("--plies", type=int, default=200)
ap.add_argument("--arena", type=int, default=0, help="average node budget per game (one shared pool); 0 = sized from the free device memory")
ap.add_argument("--every", type=int, default=10, help="plies per progress line / drain")
ap.add_argument("--evaluator", default="hash", choices=["hash", "dummy", "resnet"])
ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
ap.add_argument("--blocks", type=int, default=6)
ap.add_argument("--filters", type=int, default=64)
args = ap.parse_args()
ev = {"hash": tak_amd.EVAL_HASH, "dummy": tak_amd.EVAL_DUMMY, "resnet": tak_amd.EVAL_RESNET}[args.evaluator]
e = tak_amd.Engine(args.board, evaluator=ev, max_batch=args.games, res_blocks=args.blocks, filters=args.filters)
if args.evaluator == "resnet":  # random-init weights (tests/torch_ref.py builds the tch-layout tensors)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_ref

    if args.precision != "f32":
        e.set_precision(args.precision)
    e.load_state_dict(torch_ref.abi_tensors(torch_ref.make_net(args.board, args.blocks, args.filters, "fc5" if args.board == 5 else "conv", seed=0, randomize_bn=False)))
# the example ring holds two drain intervals of every game (a position per game and ply): nothing is dropped between drains
ring = max(1 << 18, 2 * args.games * args.every)
e.selfplay_create(args.games, arena_nodes=args.arena, seed=1, rollouts=args.rollouts, max_examples=ring)
t0 = time.time()
drained = 0
lens = []
for p in range(0, args.plies, args.every):
    e.selfplay_step(args.every)
    st = e.selfplay_stats()  # synchronises and raises on any device error flag
    hdr, states, moves, visits = e.selfplay_drain(ring)
    drained += len(hdr)
    pool = e.search_pool()
    print(json.dumps({"ply": p + args.every, "t": round(time.time() - t0, 1), **st, "drained": drained,
                      "pool_nodes_in_use": pool["in_use"], "pool_nodes_peak": pool["peak"], "pool_nodes_total": pool["total"]}), flush=True)
assert st["dropped_examples"] == 0, "the example ring was overrun between two drains"
assert drained == st["examples"]
print("soak ok")

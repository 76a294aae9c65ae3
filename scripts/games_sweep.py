#!/usr/bin/env python3
"""Where the engine leaves the latency-bound regime: config C2's network and search settings (5x5, 6 x 64, 400 sims/move) at
32 … 16 384 concurrent games, 2 plies timed after 1 warm-up ply each — expansions/s, time per lock-step iteration, and the network
forward's share of it (HIP events, every 8th forward).  One JSON line per width; `python scripts/games_sweep.py > profiles/r06_…`.
The reference runs 32 games (train/src/self_play.rs:94); the engine's unit of work is one leaf per game and iteration, so the
width is the batch size of every kernel in the loop."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref  # noqa: E402

import tak_amd  # noqa: E402

board, blocks, filters, head = 5, 6, 64, "fc5"
rollouts = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if len(sys.argv) > 2:
    board, blocks, filters, head = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
weights = torch_ref.abi_tensors(torch_ref.make_net(board, blocks, filters, head, seed=0, randomize_bn=False))
widths = [int(g) for g in os.environ["GAMES"].split(",")] if os.environ.get("GAMES") else (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)
for games in widths:
    e = tak_amd.Engine(board, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV,
                       evaluator=tak_amd.EVAL_RESNET, max_batch=games)
    if os.environ.get("PRECISION"):
        e.set_precision(os.environ["PRECISION"])
    e.load_state_dict(weights)
    e.selfplay_create(games, seed=0, rollouts=rollouts, max_examples=1 << 16)
    e.selfplay_step(1)
    e.sync()
    s0 = e.selfplay_stats()
    e.profile_enable(8)
    t0 = time.perf_counter()
    e.selfplay_step(2)
    e.sync()
    dt = time.perf_counter() - t0
    prof = e.profile_read()
    s1 = e.selfplay_stats()
    e.close()
    exp = s1["expansions"] - s0["expansions"]
    iters = 2 * (rollouts + 1)
    fwd = 1000.0 * prof["forward_ms"] / max(prof["forwards"], 1)
    print(json.dumps({"games": games, "expansions_per_s": exp / dt, "us_per_iteration": 1e6 * dt / iters, "forward_us": fwd,
                      "tree_and_launch_us": 1e6 * dt / iters - fwd, "expansions_per_s_per_game": exp / dt / games,
                      "network": f"{board}x{board} {blocks}x{filters} {head}", "sims_per_move": rollouts}), flush=True)

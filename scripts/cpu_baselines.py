#!/usr/bin/env python3
"""The CPU baselines BASELINE.md §4.2 plans, on the host cores of the box it runs on (core count stated): the CPU port of the
hot path (oracle/: AoS rules + scalar MCTS under OpenMP, network = PyTorch-CPU fp32) —
  C1 in full (5×5, 64 games, 100 sims/move, 80 games in all, random-init 6×64 net), and ≈ 60-second slices of C2 (5×5, 6×64) and
  C3 (6×6, 10×128) on 256 of their 4096 games; each also with the reference's DummyNet (board + MCTS alone).
The reference `train` binary itself cannot run here (no cargo / rustc, no libtorch 1.11): this is the "port" baseline.
    python scripts/cpu_baselines.py [--threads 16] [--slice-seconds 60] > profiles/rNN_cpu_baselines.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--slice-seconds", type=float, default=60.0)
args = ap.parse_args()

import torch

import torch_ref
from oracle import oracle as orc

torch.set_num_threads(min(args.threads, os.cpu_count() or 1))
threads = int(torch.get_num_threads())


def run(n, blocks, filters, head, games, rollouts, total_games, budget, dummy=False, max_plies=100000):
    net = torch_ref.make_net(n, blocks, filters, head, seed=0, randomize_bn=False)
    kw = dict(head=orc.HEAD_FC5 if head == "fc5" else orc.HEAD_CONV, seed=0, rollouts=rollouts, total_games=total_games)
    if dummy:
        sp = orc.SelfPlay(n, games, evaluator=orc.EVAL_DUMMY, **kw)
    else:
        sp = orc.SelfPlay(n, games, py_eval=lambda st: torch_ref.forward(net, orc.encode(n, st)), **kw)
    sp.set_threads(threads)
    t0 = time.perf_counter()
    plies = 0
    while True:
        sp.step(1)
        plies += 1
        dt = time.perf_counter() - t0
        alive = sp.states()[1].any()
        if not alive or dt > budget or plies >= max_plies:
            break
    st = sp.stats()
    return {"expansions_per_s": st["expansions"] / dt, "seconds": dt, "plies": plies, "expansions": st["expansions"],
            "games_finished": st["games_finished"], "ran_to_completion": not alive}


out = {"cores": threads, "kind": "port (oracle scalar MCTS under OpenMP + PyTorch-CPU fp32 network); the reference train binary is not runnable here",
       "C1_full": run(5, 6, 64, "fc5", 64, 100, 80, budget=900.0),
       "C1_full_board_and_mcts_only": run(5, 6, 64, "fc5", 64, 100, 80, budget=900.0, dummy=True),
       "C2_slice_256_of_4096_games": run(5, 6, 64, "fc5", 256, 400, 0, budget=args.slice_seconds),
       "C2_slice_board_and_mcts_only_4096_games": run(5, 6, 64, "fc5", 4096, 400, 0, budget=args.slice_seconds, dummy=True),
       "C3_slice_256_of_4096_games": run(6, 10, 128, "conv", 256, 400, 0, budget=args.slice_seconds),
       "C3_slice_board_and_mcts_only_4096_games": run(6, 10, 128, "conv", 4096, 400, 0, budget=args.slice_seconds, dummy=True)}
print(json.dumps(out, indent=1))

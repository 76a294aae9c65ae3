#!/usr/bin/env python3
"""Does the loop learn?  self-play → Network::train → commit, repeated, with the current network pitted against the
INITIAL (randomly initialised) one every few rounds.  Everything runs through the C ABI on one GPU; no checker involved.
The reference's schedule (LEARNING_RATE 1e-4, one optimiser step per 20 chunks) needs days of self-play; this
demonstration takes one step per chunk at a higher rate so that a few minutes show the trend.

    python scripts/learn_demo.py [--rounds 12 --games 2048 --rollouts 64 --examples 40000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--rollouts", type=int, default=64)
    ap.add_argument("--examples", type=int, default=40000)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--chunks-in-step", type=int, default=1)
    ap.add_argument("--pit-every", type=int, default=3)
    ap.add_argument("--pit-pairs", type=int, default=64)
    ap.add_argument("--pit-rollouts", type=int, default=12)
    ap.add_argument("--pit-batch", type=int, default=8)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--arena", type=int, default=1 << 16)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import tak_amd

    head = tak_amd.HEAD_FC5 if args.board == 5 else tak_amd.HEAD_CONV
    kw = dict(res_blocks=args.blocks, filters=args.filters, policy_head=head, evaluator=tak_amd.EVAL_RESNET)
    eng = tak_amd.Engine(args.board, max_batch=max(args.games, 2 * args.pit_pairs * args.pit_batch), **kw)
    first = tak_amd.Engine(args.board, max_batch=2 * args.pit_pairs * args.pit_batch, **kw)
    if args.precision != "f32":
        eng.set_precision(args.precision)
        first.set_precision(args.precision)
    eng.init_random(seed=args.seed)       # Network::default()
    first.init_random(seed=args.seed)     # the same tensors: this opponent never changes
    prev = tak_amd.Engine(args.board, max_batch=2 * args.pit_pairs * args.pit_batch, **kw)  # the network of the previous pit
    if args.precision != "f32":
        prev.set_precision(args.precision)
    prev.load_state_dict(first.state_dict())
    eng.train_create(chunk_size=500, chunks_in_step=args.chunks_in_step, learning_rate=args.lr)
    t_start = time.perf_counter()
    for rnd in range(args.rounds):
        eng.selfplay_create(args.games, arena_nodes=args.arena, seed=args.seed + rnd, rollouts=args.rollouts, max_examples=2 * args.examples)
        t0 = time.perf_counter()
        got = None
        while got is None or len(got[0]) < args.examples:
            eng.selfplay_step(8)
            eng.sync()
            part = eng.selfplay_drain(args.examples)
            got = part if got is None else [np.concatenate([a, b]) for a, b in zip(got, part)]
        st = eng.selfplay_stats()
        t_sp = time.perf_counter() - t0
        hdr, states, moves, visits = [a[: args.examples] for a in got]
        t0 = time.perf_counter()
        lp, lz, steps = eng.train(states, hdr["n_moves"], moves, visits, hdr["result"], seed=rnd)
        eng.train_commit()
        t_tr = time.perf_counter() - t0
        line = {"round": rnd, "selfplay_s": round(t_sp, 2), "games_finished": st["games_finished"], "plies": st["plies"],
                "white_wins": st["white_wins"], "black_wins": st["black_wins"], "draws": st["draws"], "expansions": st["expansions"],
                "train_s": round(t_tr, 2), "optimiser_steps": steps, "loss_p": round(lp, 4), "loss_z": round(lz, 4)}
        if (rnd + 1) % args.pit_every == 0 or rnd == args.rounds - 1:
            t0 = time.perf_counter()
            r = tak_amd.pit(eng, first, pairs=args.pit_pairs, rollouts=args.pit_rollouts, batch=args.pit_batch, idle_rollouts=1,
                            seed=1000 + rnd, max_plies=250, arena_nodes=1 << 15)
            line["pit_vs_initial"] = {k: r[k] for k in ("wins", "losses", "draws", "unfinished", "win_rate")}
            r = tak_amd.pit(eng, prev, pairs=args.pit_pairs, rollouts=args.pit_rollouts, batch=args.pit_batch, idle_rollouts=1,
                            seed=2000 + rnd, max_plies=250, arena_nodes=1 << 15)
            line["pit_vs_previous_pit"] = {k: r[k] for k in ("wins", "losses", "draws", "unfinished", "win_rate")}
            prev.load_state_dict(eng.state_dict())
            line["pit_s"] = round(time.perf_counter() - t0, 2)
        line["elapsed_s"] = round(time.perf_counter() - t_start, 1)
        print(json.dumps(line), flush=True)
    eng.close()
    first.close()
    prev.close()


if __name__ == "__main__":
    main()

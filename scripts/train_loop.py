#!/usr/bin/env python3
"""End-to-end rehearsal of config C5 on one GPU: self-play → examples → Network::train → commit → self-play again.
Everything runs through the C ABI (no CPU checker involved).  Prints timings of the training step.

    python scripts/train_loop.py [--blocks 10 --filters 128 --games 1024 --rollouts 32 --examples 4000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--rollouts", type=int, default=32)
    ap.add_argument("--examples", type=int, default=4000)
    ap.add_argument("--chunk", type=int, default=500)
    ap.add_argument("--chunks-in-step", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--pit-pairs", type=int, default=0, help="> 0: gate every round with tg_pit (train/src/main.rs:98-106)")
    ap.add_argument("--pit-rollouts", type=int, default=50, help="batches per move (pit.rs ROLLOUTS)")
    ap.add_argument("--pit-batch", type=int, default=16, help="virtual rollouts per batch (pit.rs BATCH_SIZE)")
    args = ap.parse_args()

    import tak_amd
    import torch_ref  # random-init weights in tch layout (PyTorch default init)

    head = "fc5" if args.board == 5 else "conv"
    net = torch_ref.make_net(args.board, args.blocks, args.filters, head, seed=0, randomize_bn=False)
    eng = tak_amd.Engine(args.board, res_blocks=args.blocks, filters=args.filters, evaluator=tak_amd.EVAL_RESNET, max_batch=args.games)
    tensors = torch_ref.abi_tensors(net)
    eng.load_state_dict(tensors)
    eng.train_create(chunk_size=args.chunk, chunks_in_step=args.chunks_in_step)
    old = None
    if args.pit_pairs:
        old = tak_amd.Engine(args.board, res_blocks=args.blocks, filters=args.filters, evaluator=tak_amd.EVAL_RESNET,
                             max_batch=2 * args.pit_pairs * args.pit_batch)
        old.load_state_dict(tensors)
    eng.selfplay_create(args.games, arena_nodes=1 << 13, seed=0, rollouts=args.rollouts, max_examples=4 * args.examples)
    report = []
    for rnd in range(args.rounds):
        t0 = time.perf_counter()
        got = [np.zeros((0,), tak_amd.engine.EXAMPLE_HEADER), np.zeros((0, eng.sb), np.uint8), np.zeros((0, 512), np.uint16), np.zeros((0, 512), np.uint32)]
        while len(got[0]) < args.examples:
            eng.selfplay_step(4)
            eng.sync()
            part = eng.selfplay_drain(args.examples)
            got = [np.concatenate([a, b]) for a, b in zip(got, part)]
        t_sp = time.perf_counter() - t0
        hdr, states, moves, visits = [a[: args.examples] for a in got]
        t0 = time.perf_counter()
        lp, lz, steps = eng.train(states, hdr["n_moves"], moves, visits, hdr["result"], seed=rnd)
        t_tr = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.train_commit()
        t_commit = time.perf_counter() - t0
        gate = None
        if old is not None:  # training_loop: keep the new network only if it beats the old one (WIN_RATE_THRESHOLD 0.55)
            t0 = time.perf_counter()
            gate = tak_amd.pit(eng, old, pairs=args.pit_pairs, rollouts=args.pit_rollouts, batch=args.pit_batch, idle_rollouts=1, seed=rnd, max_plies=200)
            gate["seconds"] = time.perf_counter() - t0
            new_tensors = {k: eng.train_get_tensor(k, v.shape) for k, v in tensors.items()}
            if gate["win_rate"] > 0.55:
                tensors = new_tensors
                old.load_state_dict(tensors)
            else:  # drop the trained copy: back to the old parameters, fresh trainer
                eng.load_state_dict(tensors)
                eng.train_create(chunk_size=args.chunk, chunks_in_step=args.chunks_in_step)
            gate["accepted"] = gate["win_rate"] > 0.55
            eng.selfplay_create(args.games, arena_nodes=1 << 13, seed=rnd + 1, rollouts=args.rollouts, max_examples=4 * args.examples)
        chunks = args.examples // args.chunk
        report.append({"round": rnd, "selfplay_s": t_sp, "examples": int(len(hdr)), "train_s": t_tr, "chunks": chunks, "steps": steps,
                       "ms_per_chunk": 1e3 * t_tr / max(chunks, 1), "positions_per_s": chunks * args.chunk * 8 / t_tr,
                       "loss_p": lp, "loss_z": lz, "commit_s": t_commit, "pit": gate, "stats": eng.selfplay_stats()})
        print(json.dumps(report[-1]), flush=True)
    eng.close()
    if old is not None:
        old.close()


if __name__ == "__main__":
    main()

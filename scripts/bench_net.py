#!/usr/bin/env python3
"""Conv-stack micro-benchmark of SURVEY.md §8(d): the network forward (tg_policy_eval_dev: packed states resident in HBM →
policy + eval in HBM) at B = 4096 and 16 384 positions, positions from random play through the engine's own rules kernels.
FLOPs per position are the algorithmic 2·MAC of SURVEY §8(d) (C2 29.24 M, C3 240.8 M, C5 net 161.7 M)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOPS = {(5, 6, 64): 29_235_200, (6, 10, 128): 240_795_648, (5, 10, 128): 161_689_600}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import tak_amd

    for (n, blocks, filters), flops in FLOPS.items():
        for B in (4096, 16384):
            e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, max_batch=B)
            if args.precision != "f32":
                e.set_precision(args.precision)
            e.init_random(seed=0)
            st = np.zeros((B, e.sb), np.uint8)
            hdr = e.sb - 16
            stones = 21 if n == 5 else 30
            st[:, hdr + 0] = n
            st[:, hdr + 4], st[:, hdr + 5], st[:, hdr + 6], st[:, hdr + 7] = stones, 1, stones, 1
            st[:, hdr + 8] = 4
            rng = np.random.default_rng(0)
            for ply in range(16):
                moves, counts = e.movegen(st)
                pick = (rng.random(B) * counts).astype(np.int64)
                st, status = e.play(st, moves[np.arange(B), pick])
                assert not status.any()
            d_states = torch.from_numpy(st).cuda()
            d_policy = torch.empty((B, e.psize), dtype=torch.float32, device="cuda")
            d_eval = torch.empty(B, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            for _ in range(3):
                e.policy_eval_dev(B, d_states.data_ptr(), d_policy.data_ptr(), d_eval.data_ptr())
            e.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                e.policy_eval_dev(B, d_states.data_ptr(), d_policy.data_ptr(), d_eval.data_ptr())
            e.sync()
            dt = (time.perf_counter() - t0) / args.reps
            assert abs(float(d_policy.sum(1).mean()) - 1.0) < 1e-4
            print(json.dumps({"board": n, "net": f"{blocks}x{filters}", "precision": args.precision, "positions": B, "ms_per_forward": round(dt * 1e3, 3),
                              "positions_per_s": round(B / dt), "tflops": round(B * flops / dt / 1e12, 1)}), flush=True)
            e.close()
            del d_states, d_policy, d_eval


if __name__ == "__main__":
    main()

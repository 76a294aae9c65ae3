#!/usr/bin/env python3
"""A/B timing of the network forward (tower + heads) on one BASELINE topology: `python scripts/ab_forward.py [c2|c3|c5]`.
Kernel variants are selected through environment variables read by the launchers (e.g. TG_TOWER_VARIANT); prints the
forward time and, from the in-library HIP-event profile, the tower's launch time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"c2": (5, 6, 64, 29_235_200), "c3": (6, 10, 128, 240_795_648), "c5": (5, 10, 128, 161_689_600)}


def main():
    import torch

    import tak_amd

    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    n, blocks, filters, flops = CFG[which]
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, max_batch=B)
    if os.environ.get("TG_PRECISION"):
        e.set_precision(os.environ["TG_PRECISION"])
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
    d_states = torch.from_numpy(st).cuda()
    d_policy = torch.empty((B, e.psize), dtype=torch.float32, device="cuda")
    d_eval = torch.empty(B, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        e.policy_eval_dev(B, d_states.data_ptr(), d_policy.data_ptr(), d_eval.data_ptr())
    e.sync()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        e.policy_eval_dev(B, d_states.data_ptr(), d_policy.data_ptr(), d_eval.data_ptr())
    e.sync()
    dt = (time.perf_counter() - t0) / reps
    e.profile_enable(1)
    for _ in range(20):
        e.policy_eval_dev(B, d_states.data_ptr(), d_policy.data_ptr(), d_eval.data_ptr())
    prof = e.profile_read()
    chk = float(d_policy[:64].double().sum()), float(d_eval[:64].double().sum())
    print(json.dumps({"cfg": which, "B": B, "env": {k: v for k, v in os.environ.items() if k.startswith("TG_")},
                      "forward_us": round(dt * 1e6, 1), "tflops": round(B * flops / dt / 1e12, 1),
                      "tower_us": round(prof["conv_ms"] / max(prof["conv_launches"], 1) * 1e3, 1),
                      "tower_frac_of_peak": round(prof["conv_flops"] / (prof["conv_ms"] / max(prof["conv_launches"], 1) * 1e-3) / 157.3e12, 4) if prof["conv_launches"] else None,
                      "tower_frac_executed": round(prof.get("conv_flops_executed", prof["conv_flops"]) / (prof["conv_ms"] / max(prof["conv_launches"], 1) * 1e-3) / 157.3e12, 4) if prof["conv_launches"] else None,
                      "checksum": chk}), flush=True)
    e.close()


if __name__ == "__main__":
    main()

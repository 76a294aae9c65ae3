#!/usr/bin/env python3
"""Narrow seam (`Network::policy_eval` from host memory): latency and PCIe-inclusive rate of
tg_policy_eval at the reference's batch size (32) and at a full batch.  Positions come from the engine's own rules
kernels (no checker involved)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import tak_amd

    out = []
    for n, blocks, filters in ((5, 6, 64), (6, 10, 128)):
        e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, max_batch=4096)
        e.init_random(seed=0)
        st = np.zeros((4096, e.sb), np.uint8)
        hdr = e.sb - 16
        stones, caps = (21, 1) if n == 5 else (30, 1)
        st[:, hdr + 0] = n
        st[:, hdr + 4], st[:, hdr + 5], st[:, hdr + 6], st[:, hdr + 7] = stones, caps, stones, caps
        st[:, hdr + 8] = 4
        rng = np.random.default_rng(0)
        for ply in range(12):  # a dozen random plies so that the batch is not 4096 copies of one position
            moves, counts = e.movegen(st)
            pick = (rng.random(len(st)) * counts).astype(np.int64)
            st, status = e.play(st, moves[np.arange(len(st)), pick])
            assert not status.any()
        for k in (1, 32, 256, 4096):
            e.policy_eval(st[:k])
            reps = 200 if k <= 256 else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                e.policy_eval(st[:k])
            dt = (time.perf_counter() - t0) / reps
            out.append({"board": n, "net": f"{blocks}x{filters}", "batch": k, "ms_per_call": round(dt * 1e3, 3), "positions_per_s": round(k / dt)})
            print(json.dumps(out[-1]), flush=True)
        e.close()


if __name__ == "__main__":
    main()

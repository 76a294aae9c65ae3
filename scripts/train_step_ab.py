#!/usr/bin/env python3
"""A/B timing of the training step on the C5 network (5x5, 10 blocks x 128 filters): `python scripts/train_step_ab.py [chunks]`.
Synthetic examples from random play through the engine's own rules kernels; prints ms per chunk (500 examples x 8 symmetries:
forward + backward) and the fraction of the f32 MFMA peak (3 x 161.69 MFLOP per position).  Kernel variants are selected
through environment variables read by the launchers (TG_WGRAD_PW, …)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import tak_amd

    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    driver = "--driver" in sys.argv  # through tg_train instead of one tg_train_chunk after the other
    n, blocks, filters, cs = 5, 10, 128, 500
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, max_batch=4096)
    e.init_random(seed=0)
    e.train_create(chunk_size=cs, chunks_in_step=20)
    count = cs
    st = np.zeros((count, e.sb), np.uint8)
    hdr = e.sb - 16
    st[:, hdr + 0] = n
    st[:, hdr + 4], st[:, hdr + 5], st[:, hdr + 6], st[:, hdr + 7] = 21, 1, 21, 1
    st[:, hdr + 8] = 4
    rng = np.random.default_rng(0)
    for ply in range(24):
        moves, counts = e.movegen(st)
        pick = (rng.random(count) * np.maximum(counts, 1)).astype(np.int64)
        nxt, status = e.play(st, moves[np.arange(count), pick])
        ok = (status == 0) & (e.result(nxt) == 0)
        st[ok] = nxt[ok]
    moves, counts = e.movegen(st)
    visits = np.zeros((count, 512), np.uint32)
    for i in range(count):
        visits[i, : counts[i]] = rng.integers(1, 50, counts[i])
    results = rng.choice([-1.0, 0.0, 1.0], count).astype(np.float32)
    for _ in range(2):
        e.train_chunk(st, counts.astype(np.int32), moves, visits, results)
    e.sync()
    if driver:
        rep = lambda a: np.concatenate([a] * chunks)
        big = (rep(st), rep(counts.astype(np.int32)), rep(moves), rep(visits), rep(results))
        e.train(*big, seed=1)
        t0 = time.perf_counter()
        lp, lz, _ = e.train(*big, seed=2)
    else:
        t0 = time.perf_counter()
        for _ in range(chunks):
            lp, lz, _ = e.train_chunk(st, counts.astype(np.int32), moves, visits, results)
    e.sync()
    dt = (time.perf_counter() - t0) / chunks
    pos = cs * 8
    print(json.dumps({"through": "tg_train" if driver else "tg_train_chunk", "env": {k: v for k, v in os.environ.items() if k.startswith("TG_")}, "ms_per_chunk": round(dt * 1e3, 3),
                      "positions_per_s": round(pos / dt), "frac_of_f32_mfma_peak": round(pos / dt * 3 * 161_689_600 / 157.3e12, 4),
                      "loss_p": lp, "loss_z": lz}), flush=True)
    e.close()


if __name__ == "__main__":
    main()

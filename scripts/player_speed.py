#!/usr/bin/env python3
"""Speed of the single-game `Player` mirror (one tree, `batch` virtual rollouts per network call) on the C2 network."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tak_amd

e = tak_amd.Engine(5, res_blocks=6, filters=64, max_batch=64)
e.init_random(seed=0)
start = np.zeros(256, np.uint8)
start[240:256] = [5, 0, 0, 0, 21, 1, 21, 1, 4, 0, 0, 0, 0, 0, 0, 0]
for batch in (1, 16, 32):
    p = tak_amd.Player(e, batch, False, start, arena_nodes=1 << 20, seed=1)
    for _ in range(20):
        p.rollout()
    e.sync()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        p.rollout()
    e.sync()
    dt = time.perf_counter() - t0
    print(f"Player batch {batch}: {n * batch / dt:.0f} virtual rollouts/s, {dt / n * 1e3:.3f} ms per rollout() call", flush=True)

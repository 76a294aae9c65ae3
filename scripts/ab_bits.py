#!/usr/bin/env python3
"""Checksum of the network forward (policy + eval bytes) for A/B runs of kernel variants: the digests of two runs with
different launcher environment variables must be equal (bit-identical outputs).  `python scripts/ab_bits.py c2 [B]`"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"c2": (5, 6, 64), "c3": (6, 10, 128), "c5": (5, 10, 128)}
import tak_amd

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n, blocks, filters = CFG[which]
e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, max_batch=B)
if os.environ.get("TG_PRECISION"):
    e.set_precision(os.environ["TG_PRECISION"])
e.init_random(seed=3)
st = np.zeros((B, e.sb), np.uint8)
hdr = e.sb - 16
stones = 21 if n == 5 else 30
st[:, hdr + 0] = n
st[:, hdr + 4], st[:, hdr + 5], st[:, hdr + 6], st[:, hdr + 7] = stones, 1, stones, 1
st[:, hdr + 8] = 4
rng = np.random.default_rng(0)
for ply in range(24):
    moves, counts = e.movegen(st)
    pick = (rng.random(B) * counts).astype(np.int64)
    st, status = e.play(st, moves[np.arange(B), pick])
p, v = e.policy_eval(st[: B - 3])  # a ragged last workgroup too
h = hashlib.sha256(p.tobytes() + v.tobytes()).hexdigest()
print(which, B, {k: v_ for k, v_ in os.environ.items() if k.startswith("TG_")}, h, float(p.sum()), float(v.sum()))

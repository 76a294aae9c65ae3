#!/usr/bin/env python3
"""Where k_backup_select spends its time: runs a few plies of C2 self-play on a library built with -DTG_TREE_STAMPS
(search_kernels.hip; TAKGPU_LIB points at it) and prints, for the waves of games 0..63 in the last iteration, the mean
s_memtime deltas between the phase boundaries.  Every stamp waits for the wave's outstanding memory operations, so the
deltas attribute latency to phases; their sum is somewhat longer than the undisturbed kernel.
    make -C tak_amd/csrc && hipcc … -DTG_TREE_STAMPS -c search_kernels.hip …; TAKGPU_LIB=… python scripts/probes/tree_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tak_amd
import torch_ref
from tak_amd import engine as E

precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
e = tak_amd.Engine(5, evaluator=tak_amd.EVAL_RESNET, max_batch=4096, res_blocks=6, filters=64)
if precision != "f32":
    e.set_precision(precision)
e.load_state_dict(torch_ref.abi_tensors(torch_ref.make_net(5, 6, 64, "fc5", seed=0, randomize_bn=False)))
e.selfplay_create(4096, arena_nodes=0, seed=1, rollouts=400, max_examples=1 << 16)
e.selfplay_step(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
e.sync()
buf = np.zeros((64, 32), np.uint64)
lib = E.load_library()
rc = lib.tg_debug_tree_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
names = {1: "leaf record + softmax stats", 2: "priors of the children", 3: "path backed up", 4: "root state + record"}
ok = (t[:, 0] > 0) & (t[:, 31] > t[:, 0])
t = t[ok]
print(f"{len(t)} waves; whole kernel body: mean {np.mean(t[:, 31] - t[:, 0]):.0f} cycles, max {np.max(t[:, 31] - t[:, 0])}")
# (slots 1 and 2, inside the backup, are overwritten by the stand-alone k_backup that ends a ply: only their sum is shown)
print(f"  {'backup (stats, priors, path)':32s} {np.mean(t[:, 3] - t[:, 0]):8.0f}")
print(f"  {names[4]:32s} {np.mean(t[:, 4] - t[:, 3]):8.0f}")
depths = []
scan = []
play = []
for w in range(len(t)):
    last = t[w, 4]
    d = 0
    for lvl in range(10):
        a, b = t[w, 5 + 2 * lvl], t[w, 6 + 2 * lvl]
        if a > last:
            scan.append(a - last)
            play.append(b - a)
            last = b
            d += 1
    depths.append(d)
    t[w, 23] = last
print(f"  descent: {np.mean(depths):.1f} levels (≤ 10 stamped) × (children scan {np.mean(scan):.0f} + play {np.mean(play):.0f})")
exp = t[:, 24] > 0
print(f"  waves that expand a leaf: {exp.sum()}")
te = t[exp]
for a, b, nm in ((23, 24, "last level → leaf"), (24, 25, "result"), (25, 26, "movegen"), (26, 27, "children created"), (27, 28, "virtual visits"), (28, 29, "path + leaf state stored"), (29, 31, "tail")):
    print(f"  {nm:32s} {np.mean(te[:, b] - te[:, a]):8.0f}")

#!/usr/bin/env python3
"""Generate tests/golden/net*.npz: the seed of the weights (tch layout), oracle-encoded inputs and the
PyTorch-CPU fp32 outputs (policy, eval) for a few small topologies.  The network arithmetic lives in
libtorch (not vendored in the reference), so PyTorch is the oracle here; the fixtures let the GPU box
check the HIP kernels without depending on its own torch build.  Run in the build container."""
import os
import sys

import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch_ref  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = [  # name, n, blocks, filters, head, positions
    ("net5_fc_2x32", 5, 2, 32, "fc5", 6),
    ("net5_fc_1x64", 5, 1, 64, "fc5", 5),
    ("net6_conv_1x32", 6, 1, 32, "conv", 4),
    ("net4_conv_1x32", 4, 1, 32, "conv", 5),
]

for name, n, blocks, filters, head, count in CASES:
    seed = zlib.crc32(name.encode()) % 1000
    net = torch_ref.make_net(n, blocks, filters, head, seed=seed)
    states = orc.random_positions(n, count, seed=42, max_plies=50, half_komi=4)
    planes = orc.encode(n, states)
    policy, ev = torch_ref.forward(net, planes)
    # weights are NOT stored (MBs): torch_ref.make_net(seed) regenerates them bit-identically on the
    # same PyTorch build; the test re-checks that by re-running the CPU forward against these outputs.
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}.npz"), states=states, policy=policy.astype(np.float32),
                        eval=ev.astype(np.float32), meta=np.array([n, blocks, filters, 0 if head == "fc5" else 1, seed]))
    print(name, policy.shape, float(policy.sum(1).mean()), ev)

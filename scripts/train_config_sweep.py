#!/usr/bin/env python3
"""The training step on topologies the test suite does not name (no residual block, 256 filters, odd chunk sizes, chunks around the
full-batch kernels' threshold of 1024 positions): tests/test_gpu_train.py's gradient gate — fp64 under the engine's ReLU decisions,
2e-5 per tensor — run as a sweep.  `python scripts/train_config_sweep.py`; prints one line per configuration, exits 1 on a failure."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = [(5, 0, 64, "fc5", 16), (6, 0, 128, "conv", 8), (4, 2, 256, "conv", 12), (5, 1, 256, "fc5", 20), (3, 0, 32, "conv", 7),
         (5, 3, 64, "fc5", 127), (5, 3, 64, "fc5", 129), (6, 2, 128, "conv", 29), (5, 2, 128, "conv", 130), (4, 1, 128, "conv", 70)]
bad = 0
for cfg in CASES:
    try:
        T.test_chunk_gradients_vs_autograd(orc, *cfg)
        print("ok ", cfg, flush=True)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("BAD", cfg, repr(ex)[:300], flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Host-to-host latency of tg_policy_eval (the narrow seam: Network::policy_eval from a host-side MCTS, alpha-tak/src/model/network.rs:26-35)
at small batches — the reference calls it with 32 leaves (train/src/self_play.rs:94) — on its own shipped topologies and the BASELINE ones.
`TG_NO_SPLIT_TOWER=1 python scripts/policy_eval_latency.py` times the one-workgroup-per-position towers of rounds 1 – 5 beside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref  # noqa: E402

import tak_amd  # noqa: E402

e0 = tak_amd.Engine(5, evaluator=tak_amd.EVAL_DUMMY, max_batch=1024)
for (n, blocks, filters, head) in [(6, 16, 128, "conv"), (5, 8, 128, "fc5"), (5, 10, 128, "fc5"), (6, 10, 128, "conv"), (5, 6, 64, "fc5")]:
    net = torch_ref.make_net(n, blocks, filters, head, seed=1)
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV,
                       evaluator=tak_amd.EVAL_RESNET, max_batch=512)
    e.load_state_dict(torch_ref.abi_tensors(net))
    g = tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=512)
    import numpy as np
    sts = np.zeros((512, e.sb), np.uint8)  # the start position: latency does not depend on the position
    hdr = e.sb - 16
    stones, caps = {5: (21, 1), 6: (30, 1)}[n]
    sts[:, hdr + 0] = n
    sts[:, hdr + 4], sts[:, hdr + 5], sts[:, hdr + 6], sts[:, hdr + 7] = stones, caps, stones, caps
    sts[:, hdr + 8] = 4
    g.close()
    for B in (1, 8, 32, 64, 128, 256, 512):
        e.policy_eval(sts[:B])
        t0 = time.perf_counter()
        for _ in range(50):
            e.policy_eval(sts[:B])
        dt = (time.perf_counter() - t0) / 50
        print(f"{n}x{n} {blocks}x{filters} {head} B={B}: tg_policy_eval {dt * 1e6:.0f} us host to host", flush=True)
    e.close()
e0.close()

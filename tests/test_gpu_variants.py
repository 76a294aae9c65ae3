"""Kernel variants selected by the launchers' A/B switches must return the same BITS: the halo-image towers against the
plain-image ones (TG_NO_HALO_TOWER), fragment-major against row-major FC input (TG_NO_FRAG_OUT), in exact f32 and on the
split-bf16 path; the ring FC against the small-batch FC is covered by test_gpu_net (batch independence), the FC's gather epilogue
against the logits-row epilogue by test_fc_gather_epilogue_builds_the_same_trees below.  The switches are read once per process, so every variant runs scripts/ab_bits.py (sha256 of
policy + eval of 4093 positions of one BASELINE topology, a ragged last workgroup included) in its own process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(cfg, batch, **env):
    e = dict(os.environ)
    for k in ("TG_NO_SPLIT_TOWER", "TG_SPLIT_AGENT_FENCES", "TG_NO_HALO_TOWER", "TG_NO_FC_GATHER", "TG_S3_NO_FC_RING", "TG_FC_PERMUTED_SRC", "TG_NO_FRAG_OUT", "TG_PRECISION", "TG_NO_CONST_BIAS", "TG_NO_FC_STATS"):
        e.pop(k, None)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_bits.py"), cfg, str(batch)], env=e, check=True,
                         capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    return out.split("} ")[1].split()[0]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,batch", [("c2", 4096), ("c3", 1024), ("c5", 2048)])
def test_launcher_variants_return_identical_bits(cfg, batch):
    base = _digest(cfg, batch)
    assert len(base) == 64
    assert _digest(cfg, batch, TG_NO_HALO_TOWER="1") == base
    if cfg != "c3":  # FC policy head
        assert _digest(cfg, batch, TG_NO_FRAG_OUT="1") == base  # row-major tower output into the ring FC
        assert _digest(cfg, batch, TG_FC_PERMUTED_SRC="1") == base  # the ring's LDS-DMA from the [chunk][column][q] weight layout
    # layer 0 with every input plane as data (the order of round 2) is another summation order: other low bits, and ITS
    # halo / plain variants agree with each other
    dense = _digest(cfg, batch, TG_NO_CONST_BIAS="1")
    assert dense != base
    assert _digest(cfg, batch, TG_NO_CONST_BIAS="1", TG_NO_HALO_TOWER="1") == dense
    s3 = _digest(cfg, batch, TG_PRECISION="bf16x3")
    assert s3 != base
    assert _digest(cfg, batch, TG_PRECISION="bf16x3", TG_NO_HALO_TOWER="1") == s3
    if cfg != "c3":  # the split-bf16 FC's ring kernel (full batches) against k_fc_s3b + k_fc_stats
        assert _digest(cfg, batch, TG_PRECISION="bf16x3", TG_S3_NO_FC_RING="1") == s3


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,batch", [("c3", 32), ("c5", 24), ("c5", 61), ("c3", 128)])
def test_split_tower_returns_the_bits_of_the_one_workgroup_tower(cfg, batch):
    """Small batches of the 128-filter networks: k_tower_split (a position over 8 / 4 / 2 workgroups, slices exchanged between layers) against
    k_tower (TG_NO_SPLIT_TOWER=1), and its exchange with agent-scope fences (TG_SPLIT_AGENT_FENCES=1) against the same-L2 fast path"""
    base = _digest(cfg, batch)
    assert _digest(cfg, batch, TG_NO_SPLIT_TOWER="1") == base
    assert _digest(cfg, batch, TG_SPLIT_AGENT_FENCES="1") == base


TREE_DIGEST = r"""
import hashlib, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import tak_amd, torch_ref
from oracle import oracle as orc
G, BATCH = 2560 // {batch}, {batch}
net = torch_ref.make_net(5, 2, 64, "fc5", seed=5)
e = tak_amd.Engine(5, res_blocks=2, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=G * BATCH)
if {precision!r} != "f32":
    e.set_precision({precision!r})
e.load_state_dict(torch_ref.abi_tensors(net))
base = orc.random_positions(5, 3000, seed=9, max_plies=60, half_komi=4)
base = base[orc.result(5, base) == 0]
sts = np.tile(base, (G // len(base) + 1, 1))[:G]
e.search_create(G, arena_nodes=1 << 12, batch=BATCH)   # BATCH virtual rollouts per tree and iteration (Player's batching): G * BATCH leaves
e.search_reset(sts)
e.search_run(40 // BATCH)
h = hashlib.sha256()
for g in range(0, G, 7):
    d = e.search_dump(g)
    for f in d.dtype.names:
        h.update(np.ascontiguousarray(d[f]).tobytes())
r = e.search_root()
for k in sorted(r):
    h.update(np.ascontiguousarray(r[k]).tobytes())
print("DIGEST", h.hexdigest())
"""


@pytest.mark.gpu
@pytest.mark.parametrize("precision,batch", [("f32", 1), ("bf16x3", 1), ("f32", 2)])
def test_fc_gather_epilogue_builds_the_same_trees(precision, batch):
    """Search iterations at ≥ 2049 leaves (round 6; ≥ 513 before) run the policy FC with its gather epilogue (no logits rows: the children's logits of every
    leaf and the statistics record with the value pre-activation); TG_NO_FC_GATHER=1 writes the logits rows and lets the backup
    gather — the round-3 data flow.  Same logits, same statistics, so the same trees, bit for bit: 366 whole trees and all roots
    of 2560 games after 40 iterations (and 1280 games with two virtual rollouts per iteration: leaf slot = game · batch + pass).  On the
    split-bf16 path the ring FC with its gather epilogue is also compared with the
    small-workgroup FC + statistics kernel + logits rows (TG_S3_NO_FC_RING)."""
    def digest(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("TG_")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", TREE_DIGEST.format(root=ROOT, precision=precision, batch=batch)], env=e, check=True, capture_output=True, text=True,
                             timeout=600).stdout
        return [l for l in out.splitlines() if l.startswith("DIGEST")][-1].split()[1]

    base = digest()
    assert base == digest(TG_NO_FC_GATHER="1")
    if precision == "bf16x3":
        assert base == digest(TG_S3_NO_FC_RING="1")

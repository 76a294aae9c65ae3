"""Kernel variants selected by the launchers' A/B switches must return the same BITS: the halo-image towers against the
plain-image ones (TG_NO_HALO_TOWER), the barrier-free policy FC against the barrier version (TG_FC_BARRIER), fragment-major
against row-major FC input (TG_NO_FRAG_OUT), in exact f32 and on the split-bf16 path.  The switches are read once per process, so every variant runs scripts/ab_bits.py (sha256 of
policy + eval of 4093 positions of one BASELINE topology, a ragged last workgroup included) in its own process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(cfg, batch, **env):
    e = dict(os.environ)
    for k in ("TG_NO_HALO_TOWER", "TG_FC_BARRIER", "TG_NO_FRAG_OUT", "TG_PRECISION", "TG_NO_CONST_BIAS", "TG_NO_FC_STATS"):
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
        assert _digest(cfg, batch, TG_FC_BARRIER="1") == base
        assert _digest(cfg, batch, TG_NO_FRAG_OUT="1") == base  # row-major tower output into the ring FC
        assert _digest(cfg, batch, TG_NO_FRAG_OUT="1", TG_FC_BARRIER="1") == base
    # layer 0 with every input plane as data (the order of round 2) is another summation order: other low bits, and ITS
    # halo / plain variants agree with each other
    dense = _digest(cfg, batch, TG_NO_CONST_BIAS="1")
    assert dense != base
    assert _digest(cfg, batch, TG_NO_CONST_BIAS="1", TG_NO_HALO_TOWER="1") == dense
    s3 = _digest(cfg, batch, TG_PRECISION="bf16x3")
    assert s3 != base
    assert _digest(cfg, batch, TG_PRECISION="bf16x3", TG_NO_HALO_TOWER="1") == s3

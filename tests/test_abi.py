"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/takgpu.h declares,
and refuses to compute without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import tak_amd

    if not os.path.exists(tak_amd.LIB_PATH):
        tak_amd.build_library()
    return tak_amd.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "takgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tg_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"libtakgpu.so does not export {name}"
    from tak_amd.engine import ABI_SYMBOLS

    assert declared == set(ABI_SYMBOLS)


def test_the_library_exports_the_c_abi_and_nothing_else(lib):
    """-fvisibility=hidden + the linker's version script (tak_amd/csrc/exports.map): `nm -D` shows the header's entry points, no tg::
    C++ symbol, no kernel handle, no std:: instantiation"""
    import tak_amd
    from tak_amd.engine import ABI_SYMBOLS

    out = subprocess.run(["nm", "-D", "--defined-only", tak_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert defined == set(ABI_SYMBOLS), sorted(defined ^ set(ABI_SYMBOLS))[:10]


def _csrc_sources():
    d = os.path.join(ROOT, "tak_amd", "csrc")
    return {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h", ".cuh"))}


def test_every_switch_goes_through_the_one_reader_and_is_in_the_table(lib, monkeypatch):
    """A/B switches: nothing in csrc calls getenv but env_on / env_int (engine.hip), every call site names an entry of the table
    tg_debug_switches walks, and `=0` / an empty value leave a switch off"""
    import tak_amd

    src = _csrc_sources()
    table = set(re.findall(r'"(TG_[A-Z0-9_]+)"', src["engine.hip"].split("k_switch_names[] = {")[1].split("};")[0]))
    used = set()
    for f, text in src.items():
        code = re.sub(r"//[^\n]*", "", text)
        if f != "engine.hip":
            assert "getenv" not in code, f"{f} reads the environment itself"
        used |= set(re.findall(r'env_(?:on|int)\("(TG_[A-Z0-9_]+)"\)', code))
    assert len(table) >= 20 and used == table, sorted(used ^ table)
    for name in table:
        monkeypatch.delenv(name, raising=False)
    assert tak_amd.debug_switches() == []
    monkeypatch.setenv("TG_NO_HALO_TOWER", "0")
    monkeypatch.setenv("TG_NO_FC_GATHER", "")
    monkeypatch.setenv("TG_WGRAD_PW", "0")
    assert tak_amd.debug_switches() == []
    monkeypatch.setenv("TG_NO_HALO_TOWER", "1")
    monkeypatch.setenv("TG_WGRAD_PW", "2")
    assert sorted(tak_amd.debug_switches()) == ["TG_NO_HALO_TOWER=1", "TG_WGRAD_PW=2"]
    # DESIGN.md documents every switch of the table
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert not [n for n in table if n not in design], [n for n in table if n not in design]


def test_train_order_is_a_permutation_and_stable():
    import numpy as np

    import tak_amd

    a, b = tak_amd.train_order(7, 1000), tak_amd.train_order(7, 1000)
    assert np.array_equal(a, b) and np.array_equal(np.sort(a), np.arange(1000))
    assert not np.array_equal(a, tak_amd.train_order(8, 1000))
    assert tak_amd.train_order(1, 0).size == 0


def test_sizes(lib):
    import tak_amd

    assert lib.tg_state_bytes(5) == 256 and lib.tg_state_bytes(6) == 384 and lib.tg_state_bytes(3) == 256
    assert tak_amd.input_channels(5) == 72 and tak_amd.input_channels(6) == 92
    assert tak_amd.policy_size(5, tak_amd.HEAD_FC5) == 1575
    assert tak_amd.policy_size(6, tak_amd.HEAD_CONV) == 9036
    assert tak_amd.policy_size(3, tak_amd.HEAD_CONV) == 243  # output_size(3), search/tests.rs DummyNet


def test_no_cpu_fallback(lib):
    import torch

    import tak_amd

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(tak_amd.TgError) as ei:
        tak_amd.Engine(5, evaluator=tak_amd.EVAL_DUMMY)
    assert ei.value.code == -2


def test_product_does_not_touch_oracle():
    # the product path must never import, link or call the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tak_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower().replace("no oracle", ""), os.path.join(dirpath, f)


def _build_c_host(tmp_path):
    exe = str(tmp_path / "selfplay_host")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "selfplay_host.c"), "-L" + os.path.join(ROOT, "tak_amd"), "-ltakgpu",
           "-Wl,-rpath," + os.path.join(ROOT, "tak_amd"), "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_header_is_plain_c99():
    """the boundary is a C ABI: takgpu.h must compile as C (not only as C++), with no other include"""
    hdr = os.path.join(ROOT, "include", "takgpu.h")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)


def test_c_host_links_and_fails_loudly_without_gpu(lib, tmp_path):
    """examples/selfplay_host.c (C99, no Python / torch) builds against the header and the library alone; without a
    GPU the engine refuses to start instead of falling back to anything"""
    import torch

    exe = _build_c_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (tests/test_gpu_net.py runs the host)")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr

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

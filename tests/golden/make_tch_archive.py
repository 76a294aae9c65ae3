#!/usr/bin/env python3
"""Generates tests/golden/tch_*.model: network checkpoints in the container format the reference binary writes.

`Network::save` (alpha-tak/src/model/net5.rs:95-104) is tch's VarStore::save → Tensor::save_multi → torch-sys
`at_save_multi`: OutputArchive::write(name, tensor, is_buffer=false) per variable, then OutputArchive::save_to.
tests/golden/tch_archive_writer.cpp makes exactly those libtorch calls; this script compiles it against the libtorch inside
this image's PyTorch wheel (g++, ≈ 20 s), feeds it seeded weights under tch's variable names (tak_amd.checkpoint.tch_names)
and commits the archives plus the expected network outputs on seeded positions (PyTorch-CPU fp32).  The variable NAMES are
still the recalled tch 0.7 convention (tch is not vendored in the reference) — what these fixtures pin is the container.

    python tests/golden/make_tch_archive.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = [("tch_net4_conv_1x32", 4, 1, 32, "conv", 11), ("tch_net6_conv_1x32", 6, 1, 32, "conv", 12)]


def build_writer(tmp):
    import torch

    t = os.path.dirname(torch.__file__)
    exe = os.path.join(tmp, "tch_archive_writer")
    subprocess.run(["g++", "-O1", "-std=c++17", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                    os.path.join(HERE, "tch_archive_writer.cpp"), f"-I{t}/include", f"-I{t}/include/torch/csrc/api/include",
                    f"-L{t}/lib", "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{t}/lib", "-o", exe], check=True)
    return exe


def write_blob(path, named):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(named)))
        for name, arr in named:
            arr = np.ascontiguousarray(arr, np.float32)
            f.write(struct.pack("<I", len(name)) + name.encode())
            f.write(struct.pack("<I", arr.ndim) + struct.pack(f"<{arr.ndim}q", *arr.shape))
            f.write(arr.tobytes())


def main():
    import torch_ref
    from oracle import oracle as orc
    from tak_amd import checkpoint

    with tempfile.TemporaryDirectory() as tmp:
        exe = build_writer(tmp)
        for stem, n, blocks, filters, head, seed in CONFIGS:
            net = torch_ref.make_net(n, blocks, filters, head, seed=seed)  # randomised BatchNorm statistics
            tensors = torch_ref.abi_tensors(net)
            blob = os.path.join(tmp, stem + ".blob")
            write_blob(blob, [(tch, tensors[abi]) for abi, tch in checkpoint.tch_names(blocks)])
            out = os.path.join(HERE, stem + ".model")
            subprocess.run([exe, blob, out], check=True)
            sts = orc.random_positions(n, 64, seed=seed, max_plies=30, half_komi=4)
            sts = sts[orc.result(n, sts) == 0][:6]
            p, v = torch_ref.forward(net, orc.encode(n, sts))
            # per tensor a checksum instead of the values: the archive itself holds them
            sums = {"sum_" + k: np.float64(a.astype(np.float64).sum()) for k, a in tensors.items()}
            np.savez_compressed(os.path.join(HERE, stem + ".expected.npz"), states=sts, policy=p, value=v, **sums)
            print(f"{stem}.model: {os.path.getsize(out)} bytes, {len(tensors)} variables")


if __name__ == "__main__":
    main()

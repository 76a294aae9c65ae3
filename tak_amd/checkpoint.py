"""tch `.model` checkpoints (reference `Network::save` / `load`, alpha-tak/src/model/net5.rs:95-104) ⇄ the named tensors of
include/takgpu.h.

tch-rs 0.7 saves a `VarStore` with `Tensor::save_multi`: a TorchScript archive whose attributes are the variables, keyed by
their VarStore path.  The reference builds every layer directly under `vs.root()`, so the paths are the bare leaf names
`weight`, `bias`, `running_mean`, `running_var`; when a name is taken tch appends `__{k}` with k = the number of variables
created so far (tch `nn::var_store::Path::add`).  The suffix therefore encodes the CREATION INDEX of a variable, and the
reference creates its layers in a fixed order (net5.rs:29-62 / net6.rs:29-57):

    conv0, bn0, [conv1, conv2, bn1, bn2] × RES_BLOCKS, policy, value

This module does not rely on the order in which tch creates the tensors INSIDE a layer (bias before weight in `nn::conv2d`,
the position of the running statistics in `nn::batch_norm2d` — tch's source is not vendored in the reference): variables are
placed on the creation axis by their suffix, un-suffixed names fill the free slots of the first layers, and each layer's
tensors are told apart by name and rank.  libtorch itself is not needed: PyTorch's `torch.jit` reads and writes the same
archive.  NOT verified against a file written by the reference binary (no Rust toolchain / tch here) — the round trip and
the robustness to intra-layer order are what `tests/test_checkpoint.py` pins.
"""
import re

import numpy as np


def _layers(res_blocks):
    out = [("conv", "conv0"), ("bn", "bn0")]
    for i in range(res_blocks):
        out += [("conv", f"res{i}.conv1"), ("conv", f"res{i}.conv2"), ("bn", f"res{i}.bn1"), ("bn", f"res{i}.bn2")]
    out += [("head", "policy"), ("linear", "value")]
    return out


_LEAVES = {"conv": ("weight", "bias"), "linear": ("weight", "bias"), "head": ("weight", "bias"),
           "bn": ("weight", "bias", "running_mean", "running_var")}


def tch_names(res_blocks, conv_order=("bias", "weight"), bn_order=("running_mean", "running_var", "weight", "bias")):
    """[(abi_name, tch_variable_name)] for a network of `res_blocks` blocks, emitting the intra-layer creation order given.
    Default = tch 0.7's constructors as recalled from its source (not vendored in the reference, so unconfirmed here):
    nn::conv2d / nn::linear create `bias` then `weight`; nn::batch_norm2d creates `running_mean`, `running_var`, then
    `weight`, `bias`.  The suffix is the creation index, so with this order the first BatchNorm of a network is
    running_mean, running_var, weight__4, bias__5 — the names VarStore::load of the reference binary looks up.  The READER
    below does not depend on this choice; the WRITER does, and tests/test_checkpoint.py pins the exact list."""
    names, taken, out = [], set(), []
    for kind, prefix in _layers(res_blocks):
        for leaf in (bn_order if kind == "bn" else conv_order):
            k = len(names)
            name = leaf if leaf not in taken else f"{leaf}__{k}"
            taken.add(name)
            names.append(name)
            out.append((f"{prefix}.{leaf}", name))
    return out


def save_tch_varstore(path, tensors, res_blocks, **order):
    """Write {abi_name: array} as a tch VarStore archive (`Network::save`)."""
    import torch

    class _Store(torch.nn.Module):
        def __init__(self):
            super().__init__()

    # tch's `Tensor::save_multi` goes through torch::serialize::OutputArchive::write(name, tensor, /*is_buffer=*/false): every
    # variable — the BatchNorm running statistics too — is a PARAMETER of the archived module, and tch's loader walks
    # module.named_parameters() only.  So parameters it is (requires_grad off: they are plain data here).
    m = _Store()
    for abi, name in tch_names(res_blocks, **order):
        t = torch.from_numpy(np.ascontiguousarray(tensors[abi], np.float32)).clone()
        m.register_parameter(name, torch.nn.Parameter(t, requires_grad=False))
    torch.jit.save(torch.jit.script(m), path)


def load_tch_varstore(path, res_blocks):
    """Read a tch VarStore archive (`Network::load`) → {abi_name: float32 array} with the names of include/takgpu.h."""
    import torch

    mod = torch.jit.load(path, map_location="cpu")
    found = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in list(mod.named_parameters()) + list(mod.named_buffers())}
    layers = _layers(res_blocks)
    total = sum(len(_LEAVES[k]) for k, _ in layers)
    if len(found) != total:
        raise ValueError(f"{path}: {len(found)} variables, expected {total} for {res_blocks} residual blocks")
    # creation index of every variable: the __k suffix, or a free slot of the first layer that creates that leaf name
    slot = {}
    plain = []
    for name in found:
        mt = re.fullmatch(r"(weight|bias|running_mean|running_var)(?:__(\d+))?", name)
        if not mt:
            raise ValueError(f"{path}: unexpected variable name {name!r}")
        if mt.group(2) is None:
            plain.append(name)
        else:
            k = int(mt.group(2))
            if k in slot or k >= total:
                raise ValueError(f"{path}: creation index {k} of {name!r} is out of range or duplicated")
            slot[k] = name
    # layer boundaries on the creation axis
    bounds, pos = [], 0
    for kind, prefix in layers:
        bounds.append((pos, pos + len(_LEAVES[kind]), kind, prefix))
        pos += len(_LEAVES[kind])
    for name in sorted(plain):  # un-suffixed = the first variable of that leaf name: conv0 for weight/bias, bn0 for the statistics
        lo, hi = (bounds[0][0], bounds[0][1]) if name in ("weight", "bias") else (bounds[1][0], bounds[1][1])
        free = [k for k in range(lo, hi) if k not in slot]
        if not free:
            raise ValueError(f"{path}: no creation slot left for {name!r}")
        slot[free[0]] = name
    out = {}
    for lo, hi, kind, prefix in bounds:
        group = {}
        for k in range(lo, hi):
            if k not in slot:
                raise ValueError(f"{path}: no variable with creation index {k} ({prefix})")
            leaf = slot[k].split("__")[0]
            if leaf in group:
                raise ValueError(f"{path}: two {leaf!r} variables in layer {prefix}")
            group[leaf] = found[slot[k]]
        if set(group) != set(_LEAVES[kind]):
            raise ValueError(f"{path}: layer {prefix} holds {sorted(group)}, expected {sorted(_LEAVES[kind])}")
        want_rank = {"conv": 4, "linear": 2, "bn": 1}.get(kind)
        if want_rank is not None and group["weight"].ndim != want_rank:
            raise ValueError(f"{path}: {prefix}.weight has rank {group['weight'].ndim}, expected {want_rank}")
        for leaf, arr in group.items():
            out[f"{prefix}.{leaf}"] = arr
    return out

"""tch VarStore archives (Network::save / load, net5.rs:95-104) ⇄ the ABI tensor names: round trip through the TorchScript
archive, robustness to the order in which tch creates the variables inside a layer, loud failure on a mismatching file.
Host logic only (no GPU).  Not a parity test against the reference binary — no file written by tch exists here."""
import itertools
import os

import numpy as np
import pytest

import torch_ref


@pytest.mark.parametrize("n,blocks,filters,head", [(5, 2, 32, "fc5"), (6, 1, 32, "conv")])
def test_round_trip_and_order_robustness(tmp_path, n, blocks, filters, head):
    from tak_amd import checkpoint

    net = torch_ref.make_net(n, blocks, filters, head, seed=2)
    tensors = torch_ref.abi_tensors(net)
    names = checkpoint.tch_names(blocks)
    # the writer's names, pinned: tch's suffix is the creation index; conv / linear create bias then weight, batch_norm2d its
    # running statistics first (tch 0.7) — the names the reference's VarStore::load looks up
    assert names[:16] == [
        ("conv0.bias", "bias"), ("conv0.weight", "weight"),
        ("bn0.running_mean", "running_mean"), ("bn0.running_var", "running_var"), ("bn0.weight", "weight__4"), ("bn0.bias", "bias__5"),
        ("res0.conv1.bias", "bias__6"), ("res0.conv1.weight", "weight__7"), ("res0.conv2.bias", "bias__8"), ("res0.conv2.weight", "weight__9"),
        ("res0.bn1.running_mean", "running_mean__10"), ("res0.bn1.running_var", "running_var__11"),
        ("res0.bn1.weight", "weight__12"), ("res0.bn1.bias", "bias__13"),
        ("res0.bn2.running_mean", "running_mean__14"), ("res0.bn2.running_var", "running_var__15"),
    ]
    assert names[-4:] == [("policy.bias", f"bias__{len(names) - 4}"), ("policy.weight", f"weight__{len(names) - 3}"),
                          ("value.bias", f"bias__{len(names) - 2}"), ("value.weight", f"weight__{len(names) - 1}")]
    assert len(names) == len(tensors)
    bn_orders = [("weight", "bias", "running_mean", "running_var"), ("running_mean", "running_var", "weight", "bias"),
                 ("bias", "running_var", "weight", "running_mean")]
    for i, (conv_order, bn_order) in enumerate(itertools.product([("bias", "weight"), ("weight", "bias")], bn_orders)):
        path = os.path.join(tmp_path, f"m{i}.model")
        checkpoint.save_tch_varstore(path, tensors, blocks, conv_order=conv_order, bn_order=bn_order)
        back = checkpoint.load_tch_varstore(path, blocks)
        assert set(back) == set(tensors)
        if i == 0:  # tch's loader walks module.named_parameters() only: every variable must be one, the BN statistics too
            import torch

            params = dict(torch.jit.load(path).named_parameters())
            want = {t for _, t in checkpoint.tch_names(blocks, conv_order=conv_order, bn_order=bn_order)}
            assert set(params) == want and not any(p.requires_grad for p in params.values())
        for k in tensors:
            assert back[k].shape == tensors[k].shape and np.array_equal(back[k], tensors[k]), (k, conv_order, bn_order)


def test_mismatching_archives_fail_loudly(tmp_path):
    from tak_amd import checkpoint

    net = torch_ref.make_net(5, 2, 32, "fc5", seed=3)
    tensors = torch_ref.abi_tensors(net)
    path = os.path.join(tmp_path, "m.model")
    checkpoint.save_tch_varstore(path, tensors, 2)
    with pytest.raises(ValueError):
        checkpoint.load_tch_varstore(path, 3)  # a network of another depth
    shallow = torch_ref.abi_tensors(torch_ref.make_net(5, 1, 32, "fc5", seed=3))
    checkpoint.save_tch_varstore(path, shallow, 1)
    with pytest.raises(ValueError):
        checkpoint.load_tch_varstore(path, 2)


ARCHIVES = [("tch_net4_conv_1x32", 4, 1, 32, "conv", 11), ("tch_net6_conv_1x32", 6, 1, 32, "conv", 12)]


@pytest.mark.parametrize("stem,n,blocks,filters,head,seed", ARCHIVES)
def test_reads_an_archive_written_the_way_tch_writes_it(stem, n, blocks, filters, head, seed):
    """tests/golden/tch_*.model were written by libtorch's torch::serialize::OutputArchive — write(name, tensor, false) per
    variable + save_to, the calls behind tch's VarStore::save (tests/golden/tch_archive_writer.cpp, make_tch_archive.py) —
    not by torch.jit.script as the round-trip tests above: the container the reference binary produces."""
    import zipfile

    from tak_amd import checkpoint

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(golden, stem + ".model")
    names = zipfile.ZipFile(path).namelist()
    assert any(nm.endswith("data.pkl") for nm in names) and sum("/data/" in nm for nm in names) == 8 + 14 * blocks
    back = checkpoint.load_tch_varstore(path, blocks)
    want = torch_ref.abi_tensors(torch_ref.make_net(n, blocks, filters, head, seed=seed))
    exp = np.load(os.path.join(golden, stem + ".expected.npz"))
    assert set(back) == set(want)
    for k in want:
        assert back[k].shape == want[k].shape and back[k].dtype == np.float32
        assert np.isclose(back[k].astype(np.float64).sum(), float(exp["sum_" + k]), rtol=0, atol=1e-9), k
        assert np.array_equal(back[k], want[k]), k  # same image, same seed: the generator's weights
    # and the network they describe returns the recorded outputs (PyTorch-CPU fp32)
    from oracle import oracle as orc

    net = torch_ref.load_abi_tensors(torch_ref.make_net(n, blocks, filters, head, seed=0), back)
    p, v = torch_ref.forward(net, orc.encode(n, exp["states"]))
    assert np.array_equal(p, exp["policy"]) and np.array_equal(v, exp["value"])

"""The reference side of the training-step gates (tests/torch_ref.py) checked on the CPU: fp64 gradients under GIVEN ReLU decisions —
the yardstick test_gpu_train / test_gpu_c5_realsize hold the engine's gradients to — must (a) reproduce plain fp64 autograd when
given the fp64 network's own decisions, (b) agree with PyTorch-f32 autograd to rounding when given PyTorch-f32's decisions, whatever
the number of decisions f32 takes differently from fp64, and (c) actually depend on the decisions."""
import copy

import numpy as np
import pytest
import torch

import torch_ref


def _batch(n, head, count, seed, orc):
    sts = orc.random_positions(n, count * 3, seed=seed, max_plies=40, half_komi=4)
    sts = sts[orc.result(n, sts) == 0][:count]
    mv, cnt = orc.movegen(n, sts)
    rng = np.random.default_rng(seed)
    visits = np.zeros((len(sts), 512), np.uint32)
    for i in range(len(sts)):
        visits[i, : cnt[i]] = rng.integers(1, 30, cnt[i])
    a_states, pi = orc.augment(n, orc.HEAD_FC5 if head == "fc5" else orc.HEAD_CONV, sts, cnt.astype(np.int32), mv, visits)
    return orc.encode(n, a_states), pi, np.repeat(rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), len(sts)), 8)


def _f32_decisions(net, planes):
    n32 = copy.deepcopy(net).train()
    masks = []
    with torch.no_grad():
        s = torch.relu(n32.bn0(n32.conv0(torch.from_numpy(planes))))
        masks.append(s > 0)
        for blk in n32.res:
            y = torch.relu(blk.bn1(blk.conv1(s)))
            masks.append(y > 0)
            s = torch.relu(blk.bn2(blk.conv2(y)) + s)
            masks.append(s > 0)
    return masks


@pytest.mark.parametrize("n,blocks,filters,head", [(5, 2, 32, "fc5"), (6, 1, 32, "conv")])
def test_fp64_gradients_under_given_relu_decisions(orc, n, blocks, filters, head):
    net = torch_ref.make_net(n, blocks, filters, head, seed=3)
    planes, pi, z = _batch(n, head, 6, seed=5, orc=orc)
    (own,), pres = torch_ref.fp64_gradients(net, planes, pi, z, [None])
    assert len(pres) == 1 + 2 * blocks
    decisions = [p > 0 for p in pres]
    (given,), _ = torch_ref.fp64_gradients(net, planes, pi, z, [decisions])
    assert all(np.array_equal(own[k], given[k]) for k in own)                       # (a)
    # (b) PyTorch f32 autograd against fp64 under PyTorch f32's decisions: rounding only
    lp, lz = torch_ref.train_chunk(net, planes, pi, z)
    g32 = torch_ref.named_grads(net)
    (g64,), _ = torch_ref.fp64_gradients(net, planes, pi, z, [_f32_decisions(net, planes)])
    for k in g64:
        if k.endswith(".bias") and "conv" in k and not k.startswith("policy"):
            continue  # zero true gradient in front of a BatchNorm
        assert np.linalg.norm(g32[k] - g64[k]) <= 5e-6 * np.linalg.norm(g64[k]), k
    # (c) one decision taken the other way moves the gradients of its layer and of every layer before it
    flipped = [d.clone() for d in decisions]
    idx = tuple(int(v) for v in torch.nonzero(flipped[-1])[0])
    flipped[-1][idx] = False
    (moved,), _ = torch_ref.fp64_gradients(net, planes, pi, z, [flipped])
    assert not np.array_equal(moved["conv0.weight"], own["conv0.weight"])
    assert np.array_equal(moved["policy.weight"], own["policy.weight"])               # behind the last ReLU: untouched

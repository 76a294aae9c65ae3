"""Where does the engine's gradient error at the C5 real size come from?  (diagnostic behind tests/test_gpu_c5_realsize.py)

For the LAST BatchNorm (res9.bn2) the bias gradient is Σ_rows g with g = dy·[y > 0]: per channel, the difference between the
engine and fp64 is either spread over all channels (a systematic error) or sits in a few channels and equals single dy elements of
rows whose pre-activation y is within rounding of zero (ReLU decisions).  Prints both views.

    python scripts/probes/dbg_grad_c5.py            (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import tak_amd  # noqa: E402
import test_gpu_c5_realsize as T  # noqa: E402
import torch_ref  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n, blocks, filters, count = 5, 10, 128, 500
net = torch_ref.make_net(n, blocks, filters, "fc5", seed=17)
shapes = {torch_ref.abi_name(k): tuple(v.shape) for k, v in net.named_parameters()}
e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
e.load_state_dict(torch_ref.abi_tensors(net))
e.train_create(chunk_size=count, chunks_in_step=1000)
ex = T._examples(orc, n, count, seed=41)
sts, cnt, mv, visits, results = ex
a_states, pi = orc.augment(n, orc.HEAD_FC5, sts, cnt, mv, visits)
planes, z = orc.encode(n, a_states), np.repeat(results, 8)
e.train_chunk(*ex)
g_eng = {k: e.train_get_grad(k, shapes[k]) for k in ("res9.bn2.bias", "res9.bn2.weight", "res9.conv2.weight", "res9.bn1.bias")}
logp_eng, v_eng = e.train_forward(a_states)

import copy  # noqa: E402

n64 = copy.deepcopy(net).double().train()
x = torch.from_numpy(planes.astype(np.float64))
s = torch.relu(n64.bn0(n64.conv0(x)))
for i, blk in enumerate(n64.res):
    y1 = torch.relu(blk.bn1(blk.conv1(s)))
    pre = blk.bn2(blk.conv2(y1)) + s
    if i == blocks - 1:
        pre.retain_grad()
        last_pre = pre
    s = torch.relu(pre)
s.retain_grad()
flat = s.reshape(s.shape[0], -1)
logp = torch.log_softmax(n64.policy(flat), dim=1)
v = torch.tanh(n64.value(flat))
b = x.shape[0]
loss = -(torch.from_numpy(pi.astype(np.float64)) * logp).sum() / b + (torch.from_numpy(np.asarray(z, np.float64))[:, None] - v).square().sum() / b
loss.backward()
print("forward: max |logp - logp64|", float(np.abs(logp_eng - logp.detach().numpy()).max()), " max |v - v64|", float(np.abs(v_eng - v.detach().numpy()[:, 0]).max()))
ypre = last_pre.detach().numpy()          # [B, C, 5, 5] pre-activation of the last ReLU
dy = s.grad.numpy()                       # gradient w.r.t. the last activation
for w in (1e-7, 1e-6, 1e-5, 1e-4):
    print(f"elements of the last pre-activation with |y| < {w:g}: {int((np.abs(ypre) < w).sum())} of {ypre.size}")
g64 = n64.res[blocks - 1].bn2.bias.grad.numpy()
d = g_eng["res9.bn2.bias"].astype(np.float64) - g64
print("res9.bn2.bias: ‖Δ‖/‖g‖ =", np.linalg.norm(d) / np.linalg.norm(g64))
order = np.argsort(-np.abs(d))
print("largest per-channel |Δ| :", [(int(c), float(d[c])) for c in order[:8]])
print("median per-channel |Δ|  :", float(np.median(np.abs(d))), " rms:", float(np.sqrt((d ** 2).mean())))
for c in order[:5]:
    # is Δ_c a single dy element of a row with y ≈ 0 in channel c?
    yc, dc = ypre[:, c].ravel(), dy[:, c].ravel()
    near = np.argsort(np.abs(yc))[:20]
    best = near[np.argmin(np.abs(np.abs(dc[near]) - abs(d[c])))]
    print(f"  channel {int(c)}: Δ = {d[c]:+.4e}; closest near-zero element: y = {yc[best]:+.3e}, dy = {dc[best]:+.4e} (|y| rank {int(np.where(near == best)[0][0])})")
e.close()

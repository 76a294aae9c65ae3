"""Config C5's training step at its REAL size — the 10-block × 128-filter network on one reference chunk of 500 examples ×
8 symmetries = 4000 positions = 100 000 rows (alpha-tak/src/model/network.rs:58-97: train_inner on `chunks_exact(500)`,
Example::to_tensors' 8-fold augmentation) — against PyTorch-CPU autograd of the same network: the kernels `bench.py`'s
`extra.train_c5` times (k_conv_halo<…,37>, k_wgrad_halo<5,2>, BatchNorm's sums from the conv accumulators, the ring FC).

What is compared, and how tightly (round 5):
  losses           relative 1e-5 against PyTorch f32
  gradients        the backward pass is the exact derivative of a piecewise-linear function ONCE the ReLU decisions are fixed, and
                   every decision is taken by the forward pass.  So the fp64 reference is differentiated with the ENGINE's decisions
                   (the masks y > 0 of all 21 layers, read back through tg_train_debug_read): every gradient tensor then has to agree
                   with no allowance at all — ‖g − g64‖₂ ≤ 2e-5·‖g64‖₂ (measured: ≤ 1e-6).  A wrong product, a wrong BatchNorm
                   moment, a dropped partial row or a wrong reduction anywhere in the backward pass fails this by orders of
                   magnitude; nothing is hidden behind a tolerance the builder chose.
  decisions        the engine's masks differ from the fp64 network's own at a few dozen of 2.7·10⁸ elements; each of those has a
                   pre-activation within the forward pass's rounding error of zero (|y64| ≤ 5e-5 at unit scale is asserted; measured
                   ≤ 2e-5), and their number is held against PyTorch-f32's own count on the same chunk (≤ 2 × + 10).  Since round 5
                   the training convolutions sum every tap in a chain of its own (conv_mainloop.cuh, SPLIT), the engine's forward
                   error per layer equals ATen's (1.8e-7 σ; one chain over all 1152 products: 4.5e-7 σ and 3 – 4 × the flips).
  plain distance   per tensor ‖g − g64(own decisions)‖₂ for the engine and for PyTorch f32 is still written to gpurun_out/ — a record
                   of what the differing decisions move (one flip moves a tensor near the loss by 2e-5 … 4e-4 of its norm, whoever
                   computes it: ATen's own f32 gradients miss a plain 2e-4 on 38 of 62 tensors; round 4 measured the ambiguity A of
                   the elements within ±3e-6 of zero, profiles/r04_a_c5_realsize_gradient_parity.json) — and held loosely against
                   PyTorch's on the same chunk (≤ 3 × + 2e-4); it is not the gate.
  Adam             given identical gradients, parameters within 2e-7 after a step"""
import json
import os

import numpy as np
import pytest

import torch_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU = 3e-6  # |pre-activation| below which a ReLU decision is counted as near zero (report only)


def _examples(orc, n, count, seed):
    sts = orc.random_positions(n, count * 3, seed=seed, max_plies=60, half_komi=4)
    sts = sts[orc.result(n, sts) == 0][:count]
    assert len(sts) == count
    mv, cnt = orc.movegen(n, sts)
    rng = np.random.default_rng(seed)
    visits = np.zeros((count, 512), np.uint32)
    for i in range(count):
        visits[i, : cnt[i]] = rng.integers(0, 50, cnt[i])
        visits[i, rng.integers(cnt[i])] += 1
    results = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), count)
    return sts, cnt.astype(np.int32), mv, visits, results


def _f32_masks(net, planes):
    """the ReLU decisions PyTorch f32 takes on this chunk (a no-grad forward in training mode: the arithmetic of its autograd run)"""
    import copy

    import torch

    n32 = copy.deepcopy(net).train()
    masks = []
    with torch.no_grad():
        x = torch.from_numpy(planes.astype(np.float32))
        s = torch.relu(n32.bn0(n32.conv0(x)))
        masks.append(s > 0)
        for blk in n32.res:
            y = torch.relu(blk.bn1(blk.conv1(s)))
            masks.append(y > 0)
            s = torch.relu(blk.bn2(blk.conv2(y)) + s)
            masks.append(s > 0)
    return masks


CFG = (5, 10, 128, "fc5", 500)
CAP = 20  # res9.conv2: conv2 of a block — its data gradient is handed down without the skip path's gradient added


@pytest.fixture(scope="module")
def c5(orc):
    """ONE chunk of the C5 step on the engine, with the backward pass's tensors of layer CAP kept (tg_train_debug_capture)"""
    import tak_amd

    n, blocks, filters, head, count = CFG
    net = torch_ref.make_net(n, blocks, filters, head, seed=17)
    shapes = {torch_ref.abi_name(k): tuple(v.shape) for k, v in net.named_parameters()}
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    e.load_state_dict(torch_ref.abi_tensors(net))
    lr, wd = 1e-3, 1e-2
    e.train_create(learning_rate=lr, weight_decay=wd, chunk_size=count, chunks_in_step=1000)
    ex = _examples(orc, n, count, seed=41)
    e.train_debug_capture(CAP)
    lp, lz, stepped = e.train_chunk(*ex)
    assert not stepped
    rows = count * 8 * n * n
    run = dict(e=e, net=net, shapes=shapes, ex=ex, lp=lp, lz=lz, lr=lr, wd=wd, rows=rows,
               g_eng={k: e.train_get_grad(k, shapes[k]) for k in shapes})
    for what in ("dy", "dz", "dx", "z", "y", "mean", "invstd"):
        run[what] = e.train_debug_read(what, CAP, (filters,) if what in ("mean", "invstd") else (rows, filters))
    run["x"] = e.train_debug_read("y", CAP - 1, (rows, filters))
    e.train_debug_capture(-1)
    yield run
    e.close()


def test_one_layer_of_the_backward_pass_without_a_relu_decision(c5):
    """Mask-free layerwise check at the real size (round 5): ONE 128 → 128 layer of the backward pass — BatchNorm backward, weight
    gradient (k_wgrad_halo + split-K reduction), data gradient (k_conv_halo with the flipped taps) — fed the engine's own saved
    input activation x, its ReLU mask and its upstream gradient dy, against fp64 autograd of conv → batch_norm(training) on the
    same operands.  No ReLU decision is taken on the reference side, so the plain bound applies: ≤ 2e-5 relative per tensor.
    The forward convolution of the same layer is held to 2.5e-7 σ rms against the fp64 convolution of the same input — what
    ATen's f32 convolution achieves (1.8e-7 σ) and a single accumulation chain over all 1152 products does not (4.5e-7 σ).
    Reference: alpha-tak/src/model/res_block.rs:13-24 (conv2 → bn2), network.rs:58-97 (backward through it)."""
    import torch
    import torch.nn.functional as F

    n, blocks, filters, head, count = CFG
    net, rows = c5["net"], c5["rows"]
    blk = net.res[(CAP - 2) // 2]
    conv, bn = blk.conv2, blk.bn2
    B = rows // (n * n)

    def nchw(a):
        return torch.from_numpy(np.ascontiguousarray(a)).double().reshape(B, n, n, filters).permute(0, 3, 1, 2).contiguous()

    def nhwc(t):
        return t.permute(0, 2, 3, 1).reshape(-1, filters).numpy()

    x = nchw(c5["x"]).requires_grad_(True)
    w = conv.weight.detach().double().requires_grad_(True)
    b = conv.bias.detach().double().requires_grad_(True)
    gamma = bn.weight.detach().double().requires_grad_(True)
    beta = bn.bias.detach().double().requires_grad_(True)
    z = F.conv2d(x, w, b, padding=1)
    z.retain_grad()
    out = F.batch_norm(z, None, None, gamma, beta, True, 0.0, 1e-5)
    g = nchw(c5["dy"] * (c5["y"] > 0))  # the engine's mask on the engine's upstream gradient
    out.backward(g)

    rep = {}

    def rel(name, ours, ref):
        ref = np.asarray(ref, np.float64)
        rep[name] = float(np.linalg.norm(np.asarray(ours, np.float64) - ref) / np.linalg.norm(ref))

    z64 = nhwc(z.detach())
    sigma = z64.std(0)
    rep["forward z: rms error / sigma"] = float(np.sqrt((((c5["z"] - z64) / sigma) ** 2).mean()))
    rel("BatchNorm mean (of sigma)", c5["mean"] / sigma, z64.mean(0) / sigma)
    rel("BatchNorm invstd", c5["invstd"], 1.0 / np.sqrt(z64.var(0) + 1e-5))
    rel("dz (BatchNorm backward)", c5["dz"], nhwc(z.grad))
    rel("dx (data gradient)", c5["dx"], nhwc(x.grad))
    name = f"res{(CAP - 2) // 2}"
    rel("dW (weight gradient)", c5["g_eng"][f"{name}.conv2.weight"], w.grad.numpy())
    rel("dgamma", c5["g_eng"][f"{name}.bn2.weight"], gamma.grad.numpy())
    rel("dbeta", c5["g_eng"][f"{name}.bn2.bias"], beta.grad.numpy())
    print("layerwise check, layer", CAP, json.dumps(rep, indent=1))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "c5_realsize_layerwise_backward.json"), "w") as f:
            json.dump(dict(layer=CAP, tensors=rep), f, indent=1)
    except OSError:
        pass
    assert rep["forward z: rms error / sigma"] <= 2.5e-7, rep
    assert rep["BatchNorm invstd"] <= 2e-7 and rep["BatchNorm mean (of sigma)"] <= 2e-6, rep
    for k in ("dz (BatchNorm backward)", "dx (data gradient)", "dW (weight gradient)", "dgamma", "dbeta"):
        assert rep[k] <= 2e-5, (k, rep)
    # the conv bias gradient (true value: zero, BatchNorm removes the mean) against the scale of dz's column sums' terms
    gb = c5["g_eng"][f"{name}.conv2.bias"]
    assert np.abs(gb).max() <= 1e-6 * np.abs(c5["dz"]).sum(0).max(), float(np.abs(gb).max())


def test_c5_chunk_gradients_and_adam_at_the_reference_chunk_size(orc, c5):
    import torch

    n, blocks, filters, head, count = CFG
    e, net, shapes, ex, lp, lz, lr, wd = (c5[k] for k in ("e", "net", "shapes", "ex", "lp", "lz", "lr", "wd"))
    sts, cnt, mv, visits, results = ex
    a_states, pi = orc.augment(n, orc.HEAD_FC5, sts, cnt, mv, visits)
    planes, z = orc.encode(n, a_states), np.repeat(results, 8)
    B = planes.shape[0]
    assert B == 4000
    g_eng = c5["g_eng"]
    L = 1 + 2 * blocks
    # the engine's ReLU decisions, layer by layer ([rows][F] NHWC → [B, F, n, n])
    m_eng = torch_ref.engine_relu_decisions(e, L, B, n, filters)

    lp_ref, lz_ref = torch_ref.train_chunk(net, planes, pi, z)  # PyTorch f32 autograd
    g32 = torch_ref.named_grads(net)
    assert abs(lp - lp_ref) <= 1e-5 * abs(lp_ref) and abs(lz - lz_ref) <= 1e-5 * max(abs(lz_ref), 1e-3), (lp, lp_ref, lz, lz_ref)
    m_t32 = _f32_masks(net, planes)
    # The gate needs ONE fp64 backward pass (the engine's decisions).  TG_C5_FULL_AUDIT=1 (scripts/collect_evidence.sh; the record in
    # profiles/) adds two more — the fp64 network's own decisions and PyTorch f32's — for the plain distances and ATen's own error
    # under fixed decisions; each pass costs a minute of CPU time.
    full = os.environ.get("TG_C5_FULL_AUDIT", "0") != "0"
    print(f"PyTorch f32 done; fp64 forward and {3 if full else 1} backward pass(es) …", flush=True)
    gs, pres = torch_ref.fp64_gradients(net, planes, pi, z, [m_eng, None, m_t32] if full else [m_eng], verbose=True)
    g64e = gs[0]
    g64, g64t = (gs[1], gs[2]) if full else (None, None)

    # ---- the decisions themselves ----
    flips = []
    for l in range(L):
        own = pres[l] > 0
        de, dt = m_eng[l] != own, m_t32[l] != own
        flips.append(dict(layer=l, engine=int(de.sum()), torch_f32=int(dt.sum()), near_zero=int((pres[l].abs() < TAU).sum()),
                          engine_max_abs_pre=float(pres[l][de].abs().max()) if de.any() else 0.0,
                          torch_f32_max_abs_pre=float(pres[l][dt].abs().max()) if dt.any() else 0.0))
    n_eng, n_t32 = sum(f["engine"] for f in flips), sum(f["torch_f32"] for f in flips)
    print(f"ReLU decisions that differ from the fp64 network's, of {L * B * filters * n * n}: engine {n_eng}, PyTorch f32 {n_t32};"
          f" largest |pre-activation| among them: engine {max(f['engine_max_abs_pre'] for f in flips):.2e}, PyTorch f32 {max(f['torch_f32_max_abs_pre'] for f in flips):.2e}")

    scale = np.sqrt(sum(float((g ** 2).sum()) for g in g64e.values()) / sum(g.size for g in g64e.values()))
    rows = []
    for name in shapes:
        d = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b))  # noqa: E731
        r = dict(tensor=name, norm=float(np.linalg.norm(g64e[name])), engine_vs_fp64_same_decisions=d(g_eng[name], g64e[name]),
                 engine_vs_torch_f32=d(g_eng[name], g32[name].astype(np.float64)))
        if full:
            r.update(torch_f32_vs_fp64_same_decisions=d(g32[name], g64t[name]), engine_vs_fp64=d(g_eng[name], g64[name]),
                     torch_f32_vs_fp64=d(g32[name], g64[name]))
        rows.append(r)
    report = dict(config="C5 network 5x5 10x128, one chunk of 500 examples x 8 symmetries (100000 rows)", tau=TAU,
                  loss_p=[lp, lp_ref], loss_z=[lz, lz_ref], relu_decisions=flips, tensors=rows)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "c5_realsize_gradient_parity.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass

    # every decision the engine takes differently lies within the forward pass's rounding error of zero, and there are not many
    assert max(f["engine_max_abs_pre"] for f in flips) <= 5e-5, flips
    assert n_eng <= 2 * n_t32 + 10, (n_eng, n_t32)
    worst_same = worst_plain = ("", 0.0)
    for r in rows:
        name = r["tensor"]
        if name.endswith(".bias") and "conv" in name and not name.startswith("policy"):  # true gradient exactly zero: rounding noise
            assert np.abs(g_eng[name]).max() <= 1e-3 * scale * np.sqrt(4000 * n * n), name
            continue
        # THE gate: with the same ReLU decisions, no allowance
        same = r["engine_vs_fp64_same_decisions"] / r["norm"]
        assert same <= 2e-5, ("engine, same decisions", r)
        worst_same = max(worst_same, (name, same), key=lambda t: t[1])
        if full:
            # the plain distance (own decisions on the reference side) is what the differing decisions move — on both sides: the
            # engine's against PyTorch f32's on the same chunk (each a handful of random single elements: a loose factor)
            assert r["engine_vs_fp64"] <= 3.0 * r["torch_f32_vs_fp64"] + 2e-4 * r["norm"], ("engine, own decisions", r)
            worst_plain = max(worst_plain, (name, r["engine_vs_fp64"] / r["norm"]), key=lambda t: t[1])
    print("worst tensor with the engine's ReLU decisions on the fp64 side (relative):", worst_same)
    if full:
        print("worst tensor against the fp64 network's own decisions (relative):", worst_plain)
    gb = g_eng["policy.bias"]  # every row of dLogits sums to zero → so does the policy bias gradient
    assert abs(float(gb.astype(np.float64).sum())) <= 1e-5

    # one Adam step on identical gradients (network.rs:40-45, 92-95): parameters within 2e-7
    opt = torch_ref.make_adam(net, lr=lr, wd=wd)
    for k, p in net.named_parameters():
        p.grad = torch.from_numpy(g_eng[torch_ref.abi_name(k)])
    opt.step()
    e.train_step()
    for k, p in net.named_parameters():
        name = torch_ref.abi_name(k)
        d = np.abs(e.train_get_tensor(name, shapes[name]) - p.detach().numpy())
        # lr·m̂/(√v̂ + eps): float rounding only, except where g + wd·p cancels down to the order of eps = 1e-8
        assert np.quantile(d, 0.999) <= 2e-7 and d.max() <= 1.001 * lr, (name, float(d.max()))
        assert np.abs(e.train_get_grad(name, shapes[name])).max() == 0.0  # zero_grad


GRAD_DUMP = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import tak_amd, torch_ref
from oracle import oracle as orc
import test_gpu_c5_realsize as T
n, blocks, filters, head, count = T.CFG
net = torch_ref.make_net(n, blocks, filters, head, seed=17)
e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
e.load_state_dict(torch_ref.abi_tensors(net))
e.train_create(chunk_size=count, chunks_in_step=1000)
lp, lz, _ = e.train_chunk(*T._examples(orc, n, count, seed=41))
out = dict(loss=np.float32([lp, lz]))
for k, v in net.named_parameters():
    out[torch_ref.abi_name(k)] = e.train_get_grad(torch_ref.abi_name(k), tuple(v.shape))
np.savez({out!r}, **out)
"""


def test_fused_backward_sums_and_two_streams_by_value_at_the_real_size(tmp_path):
    """The default path (BatchNorm-backward sums from the data-gradient convolution's epilogue, weight gradients on their own stream)
    against the plain path (TG_NO_BWD_SUMS_FUSION=1: a pass over dy, y, z; TG_TRAIN_ONE_STREAM=1) at B = 4000, where the partial-row
    counts differ from the small configurations the bit-exact A/B digests run at.  The forward pass — and with it every ReLU mask —
    is the same in both, so apart from the order of two double-precision sums per channel the gradients must agree: ≤ 1e-5
    relative per tensor, no allowance.  A dropped or mis-indexed partial row would cost O(1e-3)."""
    import subprocess
    import sys

    def run(tag, **env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("TG_")}
        e.update(env)
        path = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", GRAD_DUMP.format(root=ROOT, out=path)], env=e, check=True, timeout=600)
        return dict(np.load(path))

    a, b = run("default"), run("plain", TG_NO_BWD_SUMS_FUSION="1", TG_TRAIN_ONE_STREAM="1")
    assert np.array_equal(a["loss"], b["loss"])
    worst = ("", 0.0)
    for k in a:
        if k == "loss" or (k.endswith(".bias") and "conv" in k and not k.startswith("policy")):
            continue  # (conv biases in front of a BatchNorm: zero true gradient, rounding noise on both sides)
        r = float(np.linalg.norm(a[k].astype(np.float64) - b[k].astype(np.float64)) / np.linalg.norm(b[k].astype(np.float64)))
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r <= 1e-5, (k, r)
    print("default vs plain backward at B = 4000: worst tensor", worst)


def test_tg_train_at_the_real_size_equals_tg_train_chunk(orc):
    """`extra.train_c5` of the bench line times tg_train — issue-ahead, copy stream — on the 10 × 128 network with chunks of 500
    examples (B = 4000 positions); its bit-equality with tg_train_chunk was only ever tested on a 2 × 64 network with 128-example
    chunks.  Here at the real size: 3 chunks of 500 examples, a step after every chunk, and again with a step after two chunks and
    one left over — parameters, BatchNorm running statistics, left-over gradients and losses, bit for bit
    (alpha-tak/src/model/network.rs:37-56, 89-96)."""
    import test_gpu_train as T

    n, blocks, filters, head, count = CFG
    ex = T._examples(orc, n, 3 * count + 3, seed=43)
    for per_step in (1, 2):
        (lp, lz), compared = T.issue_ahead_against_chunk_by_chunk(orc, (n, blocks, filters, head, count, 3, per_step), seed=9, ex=ex)
        print(f"C5 real size, {per_step} chunk(s) per step: tg_train == tg_train_chunk on {compared} tensors; mean losses {lp:.6f} / {lz:.6f}")

"""Per-layer error budget of the training step's FORWARD pass at config C5's real size (10 × 128 network, one chunk of 500 examples ×
8 symmetries = 100 000 rows): the engine (HIP) beside PyTorch-CPU f32, both against an fp64 run of the same network.

For every conv layer l (0 = conv0, 1 + 2i = res{i}.conv1, 2 + 2i = res{i}.conv2), relative to the fp64 network:
  z      conv output (before BatchNorm): rms and max error in units of the layer's σ(z)                  — accumulated error
  zloc   the same with the layer's OWN f32 input fed to an fp64 convolution                               — this layer's convolution alone
  mean   batch mean, in units of σ(z);  istd   relative error of 1/√(var + eps)                           — BatchNorm's statistics
  y      activation after BatchNorm / skip / ReLU: rms and max absolute error
  flips  ReLU decisions that differ from the fp64 network's (of F · rows)

    python scripts/train_error_budget.py [out.json]          (GPU box; needs the oracle for the examples)
"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import tak_amd  # noqa: E402
import test_gpu_c5_realsize as T  # noqa: E402
import torch_ref  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def layers_of(net):
    out = [("conv0", net.conv0, net.bn0, None)]
    for i, blk in enumerate(net.res):
        out.append((f"res{i}.conv1", blk.conv1, blk.bn1, None))
        out.append((f"res{i}.conv2", blk.conv2, blk.bn2, "skip"))
    return out


def nhwc(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def nchw(a, n, dtype):
    b = a.shape[0] // (n * n)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).reshape(b, n, n, a.shape[1]).permute(0, 3, 1, 2).contiguous()


def main():
    n, blocks, filters, count = 5, 10, 128, 500
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    net = torch_ref.make_net(n, blocks, filters, "fc5", seed=17)
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    e.load_state_dict(torch_ref.abi_tensors(net))
    e.train_create(chunk_size=count, chunks_in_step=1000)
    ex = T._examples(orc, n, count, seed=41)
    sts, cnt, mv, visits, results = ex
    a_states, pi = orc.augment(n, orc.HEAD_FC5, sts, cnt, mv, visits)
    planes = orc.encode(n, a_states)
    e.train_chunk(*ex)
    rows = planes.shape[0] * n * n

    net64 = copy.deepcopy(net).double().train()
    net32 = copy.deepcopy(net).train()
    x64 = torch.from_numpy(planes.astype(np.float64))
    x32 = torch.from_numpy(planes.astype(np.float32))
    eps = 1e-5
    report = []
    s64, s32 = x64, x32            # inputs of the current layer
    skip64 = skip32 = None         # block inputs
    y_eng_prev = None
    with torch.no_grad():
        for l, ((name, c64, b64, sk), (_, c32, b32, _)) in enumerate(zip(layers_of(net64), layers_of(net32))):
            z64 = c64(s64)
            z32 = c32(s32)
            z_e = e.train_debug_read("z", l, (rows, filters)).astype(np.float64)
            mean_e = e.train_debug_read("mean", l, (filters,)).astype(np.float64)
            istd_e = e.train_debug_read("invstd", l, (filters,)).astype(np.float64)
            y_e = e.train_debug_read("y", l, (rows, filters))
            z64n = nhwc(z64).numpy()
            z32n = nhwc(z32).numpy().astype(np.float64)
            mu64 = z64n.mean(0)
            var64 = z64n.var(0)
            sig = np.sqrt(var64)
            istd64 = 1.0 / np.sqrt(var64 + eps)
            # PyTorch f32's own batch statistics: what batch_norm(training=True) normalises with (recomputed the way ATen reports them:
            # save_mean / save_invstd are not exposed, so take them from a functional call on z32)
            y32_bn = F.batch_norm(z32, None, None, b32.weight, b32.bias, True, 0.0, eps)
            # solve mean / invstd per channel from two rows of (z, y) is ill-conditioned; use ATen's native op instead
            _, save_mean, save_istd = torch.native_batch_norm(z32, b32.weight, b32.bias, None, None, True, 0.0, eps)
            mu32, istd32 = save_mean.numpy().astype(np.float64), save_istd.numpy().astype(np.float64)
            # this layer's convolution alone: fp64 convolution of the implementation's own f32 input
            if l == 0:
                in_e = x64
            else:
                in_e = nchw(y_eng_prev, n, torch.float64)
            zloc_e = nhwc(c64(in_e)).numpy()
            zloc_32 = nhwc(c64(s32.double())).numpy()
            # activations
            pre64 = b64(z64) if sk is None else b64(z64) + skip64
            pre32 = y32_bn if sk is None else y32_bn + skip32
            y64 = torch.relu(pre64)
            y32 = torch.relu(pre32)
            pre64n, y64n = nhwc(pre64).numpy(), nhwc(y64).numpy()
            y32n = nhwc(y32).numpy()
            pre32n = nhwc(pre32).numpy()

            def stats(d, unit):
                d = d / unit
                return float(np.sqrt((d ** 2).mean())), float(np.abs(d).max())

            rec = dict(layer=l, name=name, sigma_z=float(sig.mean()), abs_mean_over_sigma=float(np.abs(mu64 / sig).mean()))
            rec["z_engine"] = stats(z_e - z64n, sig)
            rec["z_torch32"] = stats(z32n - z64n, sig)
            rec["zloc_engine"] = stats(z_e - zloc_e, sig)
            rec["zloc_torch32"] = stats(z32n - zloc_32, sig)
            rec["mean_engine"] = stats(mean_e - mu64, sig)
            rec["mean_torch32"] = stats(mu32 - mu64, sig)
            rec["istd_engine"] = stats(istd_e / istd64 - 1.0, 1.0)
            rec["istd_torch32"] = stats(istd32 / istd64 - 1.0, 1.0)
            # mean / invstd against the implementation's OWN z (fp64 moments of its f32 z): the statistics step alone
            rec["meanloc_engine"] = stats(mean_e - z_e.mean(0), sig)
            rec["meanloc_torch32"] = stats(mu32 - z32n.mean(0), sig)
            rec["istdloc_engine"] = stats(istd_e * np.sqrt(z_e.var(0) + eps) - 1.0, 1.0)
            rec["istdloc_torch32"] = stats(istd32 * np.sqrt(z32n.var(0) + eps) - 1.0, 1.0)
            rec["y_engine"] = stats(y_e.astype(np.float64) - y64n, 1.0)
            rec["y_torch32"] = stats(y32n.astype(np.float64) - y64n, 1.0)
            rec["flips_engine"] = int(((y_e > 0) != (pre64n > 0)).sum())
            rec["flips_torch32"] = int(((pre32n > 0) != (pre64n > 0)).sum())
            rec["near_zero_3e-6"] = int((np.abs(pre64n) < 3e-6).sum())
            rec["abs_pre_rms"] = float(np.sqrt((pre64n ** 2).mean()))
            report.append(rec)
            print(f"{l:2d} {name:12s} σz {rec['sigma_z']:.3f} |μ|/σ {rec['abs_mean_over_sigma']:.2f} | z rms eng {rec['z_engine'][0]:.2e} t32 {rec['z_torch32'][0]:.2e}"
                  f" | zloc rms eng {rec['zloc_engine'][0]:.2e} t32 {rec['zloc_torch32'][0]:.2e} | mean eng {rec['mean_engine'][0]:.1e} t32 {rec['mean_torch32'][0]:.1e}"
                  f" (loc {rec['meanloc_engine'][0]:.1e} / {rec['meanloc_torch32'][0]:.1e}) | istd eng {rec['istd_engine'][0]:.1e} t32 {rec['istd_torch32'][0]:.1e}"
                  f" (loc {rec['istdloc_engine'][0]:.1e} / {rec['istdloc_torch32'][0]:.1e}) | y rms eng {rec['y_engine'][0]:.2e} max {rec['y_engine'][1]:.1e}"
                  f" t32 {rec['y_torch32'][0]:.2e} max {rec['y_torch32'][1]:.1e} | flips eng {rec['flips_engine']} t32 {rec['flips_torch32']} (|pre|<3e-6: {rec['near_zero_3e-6']})",
                  flush=True)
            # next layer
            if sk is None and l >= 1:      # conv1 of a block: the block input stays the skip
                pass
            if l == 0 or sk == "skip":      # output of conv0 / of a block = next block's input
                skip64, skip32 = y64, y32
            s64, s32 = y64, y32
            y_eng_prev = y_e
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "train_error_budget.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(dict(config="C5 network 5x5 10x128, one chunk of 500 examples x 8 symmetries (100000 rows), forward in training mode",
                       units="z, zloc, mean: sigma(z) of the layer; istd: relative; y: absolute; (rms, max)", layers=report), f, indent=1)
    e.close()


if __name__ == "__main__":
    main()

"""Plain PyTorch fp32 statement of the reference's networks (alpha-tak/src/model/{net5,net6,res_block}.rs),
used as the arithmetic oracle for the HIP network kernels: libtorch's conv2d / batch_norm(eval) / relu /
linear / softmax / tanh are the same ATen ops tch-rs calls.  Test infrastructure only."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def input_channels(n):
    stones, caps = {3: (10, 0), 4: (15, 0), 5: (21, 1), 6: (30, 1)}[n]
    return (n + 2 + 6) * 2 + 2 + 2 * stones + 2 * caps


class ResBlock(nn.Module):  # res_block.rs:13-24
    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(f)
        self.bn2 = nn.BatchNorm2d(f)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class TakNet(nn.Module):
    """head 'fc5' = Net5 (net5.rs:19-131), head 'conv' = Net6 (net6.rs:19-139); blocks/filters runtime."""

    def __init__(self, n, res_blocks, filters, head):
        super().__init__()
        self.n, self.head, self.f = n, head, filters
        self.conv0 = nn.Conv2d(input_channels(n), filters, 3, padding=1)
        self.bn0 = nn.BatchNorm2d(filters)
        self.res = nn.ModuleList([ResBlock(filters) for _ in range(res_blocks)])
        if head == "fc5":
            assert n == 5
            self.policy = nn.Linear(filters * 25, 1575)
        else:
            self.policy = nn.Conv2d(filters, 3 + 4 * (2 ** n - 2), 3, padding=1)
        self.value = nn.Linear(filters * n * n, 1)

    def forward(self, x):  # forward_mcts (eval mode BN)
        s = F.relu(self.bn0(self.conv0(x)))
        for blk in self.res:
            s = blk(s)
        if self.head == "fc5":
            p = self.policy(s.reshape(s.shape[0], -1))
        else:
            p = self.policy(s).reshape(s.shape[0], -1)
        p = torch.softmax(p, dim=1)
        v = torch.tanh(self.value(s.reshape(s.shape[0], -1)))
        return p, v[:, 0]


    def forward_training(self, x):  # forward_training (net5.rs:113-118 / net6.rs:111-122): log_softmax, BN per self.training
        s = F.relu(self.bn0(self.conv0(x)))
        for blk in self.res:
            s = blk(s)
        if self.head == "fc5":
            p = self.policy(s.reshape(s.shape[0], -1))
        else:
            p = self.policy(s).reshape(s.shape[0], -1)
        return torch.log_softmax(p, dim=1), torch.tanh(self.value(s.reshape(s.shape[0], -1)))


def make_net(n, res_blocks, filters, head, seed=0, randomize_bn=True):
    torch.manual_seed(seed)
    net = TakNet(n, res_blocks, filters, head)
    if randomize_bn:  # exercise the BN fold: non-trivial affine + running statistics
        g = torch.Generator().manual_seed(seed + 1)
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.data = torch.rand(m.num_features, generator=g) * 1.0 + 0.5
                m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
                m.running_mean = torch.randn(m.num_features, generator=g) * 0.1
                m.running_var = torch.rand(m.num_features, generator=g) * 1.0 + 0.5
    return net.eval()


def abi_tensors(net):
    """state_dict → {ABI tensor name: float32 array} (names of include/takgpu.h)."""
    out = {}
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        parts = k.split(".")
        if parts[0] == "res":
            name = f"res{parts[1]}." + ".".join(parts[2:])
        else:
            name = k
        out[name] = v.detach().cpu().numpy().astype(np.float32)
    return out


def load_abi_tensors(net, tensors):
    """inverse of abi_tensors: {ABI tensor name: array} → the module's parameters and buffers"""
    sd = net.state_dict()
    for k in list(sd.keys()):
        if k.endswith("num_batches_tracked"):
            continue
        sd[k] = torch.from_numpy(np.ascontiguousarray(tensors[abi_name(k)], np.float32)).reshape(sd[k].shape).clone()
    net.load_state_dict(sd)
    return net.eval()


@torch.no_grad()
def forward(net, planes):
    p, v = net(torch.from_numpy(np.ascontiguousarray(planes, np.float32)))
    return p.numpy(), v.numpy()


def abi_name(k):
    parts = k.split(".")
    return f"res{parts[1]}." + ".".join(parts[2:]) if parts[0] == "res" else k


def train_chunk(net, planes, pi, z):
    """train_inner (alpha-tak/src/model/network.rs:58-91): BN in training mode, loss_p = −Σπ·logp / B,
    loss_z = Σ(z − v)² / B, backward (gradients accumulate in .grad).  Returns (loss_p, loss_z)."""
    net.train()
    x = torch.from_numpy(np.ascontiguousarray(planes, np.float32))
    p = torch.from_numpy(np.ascontiguousarray(pi, np.float32))
    zt = torch.from_numpy(np.ascontiguousarray(z, np.float32))[:, None]
    logp, v = net.forward_training(x)
    b = x.shape[0]
    loss_p = -(p * logp).sum() / b
    loss_z = (zt - v).square().sum() / b
    (loss_z + loss_p).backward()
    return float(loss_p.detach()), float(loss_z.detach())


def make_adam(net, lr=1e-4, wd=1e-4):
    """Adam { wd, ..Default::default() }.build(vs, lr) (network.rs:40-45): torch::optim::Adam with L2 weight decay"""
    return torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)


def named_grads(net):
    return {abi_name(k): v.grad.detach().numpy().copy() for k, v in net.named_parameters()}


def fp64_gradients(net, planes, pi, z, mask_sets, verbose=False):
    """fp64 forward of train_inner's loss (network.rs:58-91) on a copy of `net`, then one backward pass per entry of mask_sets:
    None = the fp64 network's own ReLU decisions, a float t = the decisions taken at y > t, a list of 1 + 2·blocks bool tensors
    [B, F, n, n] = THOSE decisions (layer order conv0, res0.conv1, res0.conv2, …).  The backward pass of a ReLU network is the exact
    derivative of a piecewise-linear function once the decisions are fixed, and the forward pass takes them: with an
    implementation's own decisions on the reference side its gradients have to agree to rounding, with no allowance for "flips".
    → ([{ABI name: gradient}], [pre-activation of every ReLU, float64 tensors])"""
    import copy

    class Relu(torch.autograd.Function):
        mode = None

        @staticmethod
        def forward(ctx, x, layer):
            ctx.save_for_backward(x)
            ctx.layer = layer
            return x.clamp_min(0.0)

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            m = Relu.mode
            if m is None:
                return g * (x > 0.0), None
            if isinstance(m, float):
                return g * (x > m), None
            return g * m[ctx.layer], None

    n64 = copy.deepcopy(net).double().train()
    for p in n64.parameters():
        p.grad = None
    x = torch.from_numpy(np.ascontiguousarray(planes, np.float64))
    pres = []

    def relu(t):
        pres.append(t.detach())
        return Relu.apply(t, len(pres) - 1)

    s = relu(n64.bn0(n64.conv0(x)))
    for blk in n64.res:  # res_block.rs:13-24
        y = relu(blk.bn1(blk.conv1(s)))
        s = relu(blk.bn2(blk.conv2(y)) + s)
    flat = s.reshape(s.shape[0], -1)
    logits = n64.policy(flat) if n64.head == "fc5" else n64.policy(s).reshape(s.shape[0], -1)
    logp = torch.log_softmax(logits, dim=1)
    v = torch.tanh(n64.value(flat))
    b = x.shape[0]
    loss = -(torch.from_numpy(np.ascontiguousarray(pi, np.float64)) * logp).sum() / b + (torch.from_numpy(np.asarray(z, np.float64))[:, None] - v).square().sum() / b
    params = list(n64.named_parameters())
    out = []
    for i, m in enumerate(mask_sets):
        Relu.mode = m
        gs = torch.autograd.grad(loss, [p for _, p in params], retain_graph=i + 1 < len(mask_sets))
        out.append({abi_name(k): g.numpy().copy() for (k, _), g in zip(params, gs)})
        if verbose:
            print(f"  fp64 backward pass {i + 1} of {len(mask_sets)} done", flush=True)
    return out, pres


def engine_relu_decisions(engine, layers, positions, n, filters):
    """the ReLU decisions an engine took in its last training chunk / forward — y > 0 of every conv layer, read back through
    tg_train_debug_read ([rows][F] NHWC) — as bool tensors [B, F, n, n] for fp64_gradients"""
    rows = positions * n * n
    return [torch.from_numpy(engine.train_debug_read("y", l, (rows, filters)) > 0).reshape(positions, n, n, filters).permute(0, 3, 1, 2)
            for l in range(layers)]

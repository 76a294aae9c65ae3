import sys, copy
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import torch_ref, tak_amd
from oracle import oracle as orc
import test_gpu_train as T
n, blocks, filters, head = 5, 1, 128, "fc5"
count = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = torch_ref.make_net(n, blocks, filters, head, seed=10 + n)
e = T._engine(n, blocks, filters, head)
e.load_state_dict(torch_ref.abi_tensors(net))
e.train_create(chunk_size=count, chunks_in_step=1000)
net64 = copy.deepcopy(net).double()
ex = T._examples(orc, n, count, seed=20)
planes, pi, z, _ = T._targets(orc, n, head, ex)
net64.train()
logp64, v64 = net64.forward_training(torch.from_numpy(planes.astype(np.float64)))
loss64 = -(torch.from_numpy(pi.astype(np.float64)) * logp64).sum() / len(planes) + (torch.from_numpy(np.asarray(z, np.float64))[:, None] - v64).square().sum() / len(planes)
loss64.backward()
torch_ref.train_chunk(net, planes, pi, z)
e.train_chunk(*ex)
shapes = T._shapes(net)
g = e.train_get_grad("conv0.weight", shapes["conv0.weight"]).astype(np.float64)
g64 = net64.conv0.weight.grad.numpy()
g32 = net.conv0.weight.grad.numpy().astype(np.float64)
err = np.sqrt(((g-g64)**2).sum(axis=(0,2,3))); e32 = np.sqrt(((g32-g64)**2).sum(axis=(0,2,3))); nr = np.sqrt((g64**2).sum(axis=(0,2,3)))
print("count", count)
for name in ("res0.conv1.weight","res0.conv2.weight","policy.weight","conv0.bias","bn0.weight"):
    gg = e.train_get_grad(name, shapes[name]).astype(np.float64)
    k=[k for k,_ in net.named_parameters() if torch_ref.abi_name(k)==name][0]
    a=dict(net64.named_parameters())[k].grad.numpy(); b=dict(net.named_parameters())[k].grad.numpy().astype(np.float64)
    print(name, np.linalg.norm(gg-a), np.linalg.norm(b-a), np.linalg.norm(a))

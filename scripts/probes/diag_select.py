import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, tak_amd, torch_ref
net = torch_ref.make_net(5, 6, 64, "fc5", seed=0, randomize_bn=False)
e = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=4096)
e.load_state_dict(torch_ref.abi_tensors(net))
e.selfplay_create(4096, arena_nodes=1 << 17, seed=0, rollouts=400)
for plies in (1, 8):
    e.selfplay_step(plies)
    print(e.search_counters())
e.close()

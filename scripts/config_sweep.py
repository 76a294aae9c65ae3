import sys, os, itertools
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, torch_ref, tak_amd
from oracle import oracle as orc
bad = 0
cfgs = [(3,0,32,"conv"),(3,2,64,"conv"),(4,1,64,"conv"),(4,0,128,"conv"),(4,3,96,"conv"),(5,1,96,"fc5"),(5,3,32,"fc5"),(5,1,160,"conv"),(5,0,64,"fc5"),(5,0,64,"conv"),
        (6,2,64,"conv"),(6,1,32,"conv"),(6,0,64,"conv"),(6,3,96,"conv"),(5,2,256,"fc5"),(6,1,256,"conv"),(5,1,128,"conv"),(6,5,128,"conv")]
for (n, blocks, filters, head) in cfgs:
    for prec in ("f32", "bf16x3"):
        if prec == "bf16x3" and not ((n == 5 and filters in (64, 128)) or (n == 6 and filters == 128)):
            continue
        try:
            net = torch_ref.make_net(n, blocks, filters, head, seed=n * 10 + blocks, randomize_bn=True)
            e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV, evaluator=tak_amd.EVAL_RESNET, max_batch=300)
            if prec != "f32": e.set_precision(prec)
            e.load_state_dict(torch_ref.abi_tensors(net))
            sts = orc.random_positions(n, 400, seed=3, max_plies=60 if n >= 5 else 10, half_komi=4)[:300]
            p, v = e.policy_eval(sts)
            p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts[:64]))
            dp, dv = np.abs(p[:64] - p_ref).max(), np.abs(v[:64] - v_ref).max()
            ok = dp <= 1e-4 and dv <= 1e-4
            for k in (1, 31, 129, 260):
                pk, vk = e.policy_eval(sts[:k])
                if not (np.array_equal(pk, p[:k]) and np.array_equal(vk, v[:k])):
                    ok = False; print("  batch", k, "differs")
            print(("ok  " if ok else "BAD ") + f"{n}x{n} {blocks}x{filters} {head} {prec}: dp {dp:.1e} dv {dv:.1e}", flush=True)
            bad += not ok
            e.close()
        except Exception as ex:
            print(f"EXC {n}x{n} {blocks}x{filters} {head} {prec}: {ex!r}", flush=True); bad += 1
print("bad:", bad)

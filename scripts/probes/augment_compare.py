import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, tak_amd, torch_ref
import test_gpu_train as T
from oracle import oracle as orc
for n, head in ((5, "conv"), (5, "fc5"), (6, "conv"), (4, "conv")):
    e = T._engine(n, 1, 64, head)
    ex = T._examples(orc, n, 40, seed=3)
    sts, cnt, mv, visits, results = ex
    a_states, pi = orc.augment(n, orc.HEAD_FC5 if head == "fc5" else orc.HEAD_CONV, sts, cnt, mv, visits)
    g_states, g_pi = e.augment_examples(sts, cnt, mv, visits)
    print(n, head, "states equal", np.array_equal(a_states, g_states), "pi equal", np.array_equal(pi, g_pi), "max|dpi|", float(np.abs(pi - g_pi).max()),
          "sums", float(pi.sum(1).min()), float(g_pi.sum(1).min()), float(g_pi.sum(1).max()))
    if not np.array_equal(pi, g_pi):
        bad = np.argwhere(pi != g_pi)
        print("  first mismatches (row, index):", bad[:6].tolist(), "row", bad[0][0], "symmetry", bad[0][0] % 8)
        r = bad[0][0]
        print("  oracle nonzero idx", np.nonzero(pi[r])[0][:12], "gpu nonzero idx", np.nonzero(g_pi[r])[0][:12])
    e.close()

import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_train as T
from oracle import oracle as orc
cfg = eval(sys.argv[1])
try:
    T.test_chunk_gradients_vs_autograd(orc, *cfg); print("ok", cfg, dict((k,v) for k,v in os.environ.items() if k.startswith("TG_")))
except Exception as ex:
    print("BAD", cfg, dict((k,v) for k,v in os.environ.items() if k.startswith("TG_")), repr(ex)[:200])

import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
from oracle import letkf_oracle as O
mia.build()
eng = mia.LetkfEngine("cuda:0")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
def T(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print("%-28s %.2f s" % (label, time.perf_counter() - t0), flush=True); return r
for k, stride, c, m in ((96, 1, 20.0, 2), (80, 1, 16.5, 1)):
    case = T("synthetic_case", lambda: O.synthetic_case(203, k, stride, seed=k + m, m=m))
    nb = T("localize", lambda: eng.localize(case["grid_x"], case["obs_x"], [c]))
    X, yb, d = dev(case["state"]), dev(case["yb"]), dev(case["d"])
    for sp in (1, 1, 0, 0):
        _cabi.set_option("tile_split", sp)
        T("analysis split=%d" % sp, lambda: eng.analysis(X, yb, d, nb, 1.1, return_flags=True, method="matfun"))
    T("oracle", lambda: O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1))
    _cabi.set_option("tile", 0)
    T("analysis tile=0", lambda: eng.analysis(X, yb, d, nb, 1.1, return_flags=True, method="matfun"))
    _cabi.set_option("tile", 1)

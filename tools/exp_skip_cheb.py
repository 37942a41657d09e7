import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
os.environ.setdefault("MIA_BUILD_FLAGS", "-DMIA_EXPERIMENTS")   # the hooks this script drives exist in experiment builds only
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
nb = eng.localize(gx, ox, [10.0])
rec = eng.pack_obs(Yb, d, torch.float32)
for skip in (0, 1, 4, 8, 15, 32):
    os.environ["MIA_EXPERIMENT_SKIP"] = str(skip)
    ts = []
    for b in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng.analysis(X, None, None, nb, 1.1, rec=rec, method="matfun", defer_retry=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 100)
    print(f"skip mask {skip:2d} (1 gather, 2 gram, 4 recurrence->deg 3, 8 output): min {min(ts[1:]):.3f} ms")

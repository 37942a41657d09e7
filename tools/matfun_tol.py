#!/usr/bin/env python3
"""matfun route: accuracy / time versus the truncation target (MIA_CHEB_LOGTOL), experiments."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
os.environ.setdefault("MIA_BUILD_FLAGS", "-DMIA_EXPERIMENTS")   # the hooks this script drives exist in experiment builds only
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g7_synthetic_configs.npz"))
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name, c, gamma, k, stride in (("c2", 10.0, None, 40, 2), ("c4", 16.5, None, 80, 1), ("c5", 10.0, 0.5, 40, 2)):
    X = g[f"{name}_state"]
    nb = eng.localize(g[f"{name}_grid_x"], g[f"{name}_obs_x"], [c])
    Xd = torch.as_tensor(X, dtype=torch.float32, device=dev)
    yb = torch.as_tensor(g[f"{name}_yb"], dtype=torch.float32, device=dev)
    d = torch.as_tensor(g[f"{name}_d"], dtype=torch.float32, device=dev)
    ref = g[f"{name}_1p1_analysis"]; xm = X.mean(axis=1, keepdims=True)
    Xb, gx, ox, Ybb, db = bench.make_case(100000, k, stride, dev)
    nbb = eng.localize(gx, ox, [c]); recb = eng.pack_obs(Ybb, db, torch.float32)
    for lt in ("15", "13", "12", "11", "10", "9"):
        os.environ["MIA_CHEB_LOGTOL"] = lt
        xa, fl = eng.analysis(Xd, yb, d, nb, 1.1, rbf_gamma=gamma, return_flags=True, method="matfun")
        f = fl.cpu().numpy()
        for _ in range(3): eng.analysis(Xb, None, None, nbb, 1.1, rec=recb, rbf_gamma=gamma, method="matfun")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng.analysis(Xb, None, None, nbb, 1.1, rec=recb, rbf_gamma=gamma, method="matfun")
        torch.cuda.synchronize()
        print(f"{name} logtol {lt:5s} err {rel(xa.cpu().numpy(), ref):.2e} inc {rel(xa.cpu().numpy()-xm, ref-xm):.2e} deg {((f>>8)&0xff).mean():.1f} | {(time.perf_counter()-t0)*100:.3f} ms")
os.environ.pop("MIA_CHEB_LOGTOL")

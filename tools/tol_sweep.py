#!/usr/bin/env python3
"""Accuracy / time of the analysis kernel versus the Jacobi stopping tolerance (experiments)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch_assimilate_amd as mia  # noqa: E402
os.environ.setdefault("MIA_BUILD_FLAGS", "-DMIA_EXPERIMENTS")   # the hooks this script drives exist in experiment builds only
mia.build()

dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g7_synthetic_configs.npz"))


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


for name, c, gamma, k, stride in (("c2", 10.0, None, 40, 2), ("c4", 16.5, None, 80, 1), ("c5", 10.0, 0.5, 40, 2)):
    X = g[f"{name}_state"]
    nb = eng.localize(g[f"{name}_grid_x"], g[f"{name}_obs_x"], [c])
    Xd = torch.as_tensor(X, dtype=torch.float32, device=dev)
    yb = torch.as_tensor(g[f"{name}_yb"], dtype=torch.float32, device=dev)
    d = torch.as_tensor(g[f"{name}_d"], dtype=torch.float32, device=dev)
    Xb, gx, ox, Ybb, db = bench.make_case(100000, k, stride, dev)
    nbb = eng.localize(gx, ox, [c])
    recb = eng.pack_obs(Ybb, db, torch.float32)
    for tol in ("1e-3", "5e-4", "2.4e-4", "1e-4", "3e-5", "1e-5", "2.4e-7"):
        os.environ["MIA_JACOBI_STOP_TOL"] = tol
        xa, fl = eng.analysis(Xd, yb, d, nb, 1.1, rbf_gamma=gamma, return_flags=True)
        ref = g[f"{name}_1p1_analysis"]
        xm = X.mean(axis=1, keepdims=True)
        e_full, e_inc = rel(xa.cpu().numpy(), ref), rel(xa.cpu().numpy() - xm, ref - xm)
        f = fl.cpu().numpy()
        for _ in range(2):
            eng.analysis(Xb, None, None, nbb, 1.1, rec=recb, rbf_gamma=gamma)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.analysis(Xb, None, None, nbb, 1.1, rec=recb, rbf_gamma=gamma)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name} stop_tol={tol:8s} err {e_full:.2e} inc {e_inc:.2e} sweeps mean {((f >> 8) & 0xff).mean():.2f} "
              f"rot rounds {(f >> 16).mean():.1f}  |  1e5 pts: {ms:.3f} ms")
os.environ.pop("MIA_JACOBI_STOP_TOL")

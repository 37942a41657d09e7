#!/usr/bin/env python3
"""Cost model of the analysis kernel: time versus forced number of Jacobi sweeps (experiments)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
os.environ.setdefault("MIA_BUILD_FLAGS", "-DMIA_EXPERIMENTS")   # the hooks this script drives exist in experiment builds only
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
cfgs = {"c2": (40, 2, 10.0, None), "c4": (80, 1, 16.5, None), "c5": (40, 2, 10.0, 0.5)}
for name in sys.argv[1:] or ["c2"]:
    k, stride, c, gamma = cfgs[name]
    X, gx, ox, Yb, d = bench.make_case(100000, k, stride, dev)
    nb = eng.localize(gx, ox, [c])
    rec = eng.pack_obs(Yb, d, torch.float32)
    for ms in ("0", "1", "2", "3", "4", "16"):
        for ppb in ("7", "24"):
            os.environ["MIA_MAX_SWEEPS"] = ms
            os.environ["MIA_PTS_PER_BLOCK"] = ppb
            for _ in range(3):
                eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=gamma)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=gamma)
            torch.cuda.synchronize()
            print(f"{name} max_sweeps={ms:3s} pts/block={ppb:3s}: {(time.perf_counter()-t0)*100:.3f} ms")

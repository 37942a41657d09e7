import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
nb = eng.localize(gx, ox, [10.0])
rec = eng.pack_obs(Yb, d, torch.float32)
for nov in (False, True):
    if nov: os.environ["MIA_EXPERIMENT_NO_V"] = "1"
    for ms in ("0", "2", "4"):
        os.environ["MIA_MAX_SWEEPS"] = ms
        for _ in range(3): eng.analysis(X, None, None, nb, 1.1, rec=rec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng.analysis(X, None, None, nb, 1.1, rec=rec)
        torch.cuda.synchronize()
        print(f"noV={nov} max_sweeps={ms}: {(time.perf_counter()-t0)*100:.3f} ms")

"""Debug helper: all (k, p) blocks of golden g3 through the matfun route in one process."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import torch_assimilate_amd as mia
from oracle import letkf_oracle as O
g = np.load("tests/golden/g3_g4_core_blocks.npz")
eng = mia.LetkfEngine("cuda:0")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
order = [int(a) for a in sys.argv[1:]] or range(len(g["cases"]))
for ci in order:
    k, p = g["cases"][ci]
    if k > 128 or min(k, p) > 64:
        continue
    yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
    X = np.random.RandomState(ci).normal(size=(2, k, 1))
    cap = max(p, 1)
    nb = eng.localize_from_dist(np.zeros((1, 1, cap)), np.tile(np.arange(cap, dtype=np.int32), (1, 1)), [1.0])
    print("case", ci, k, p, "p_max", nb.p_max, flush=True)
    xa, fl = eng.analysis(dev(X), dev(yb), dev(d), nb, 1.1, return_flags=True, method="matfun")
    torch.cuda.synchronize()
    ref = O.apply_weights(X, g[f"etkf_{ci}_1p1"][None])
    print("  ok flags", int(fl.cpu()[0]), "err", float(np.linalg.norm(xa.cpu().numpy() - ref) / np.linalg.norm(ref)), flush=True)

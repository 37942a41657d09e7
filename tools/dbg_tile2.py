import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np, torch
import torch_assimilate_amd as mia
from oracle import letkf_oracle as O
mia.build()
np.set_printoptions(linewidth=200, precision=2)
eng = mia.LetkfEngine("cuda:0")
def dev(a, dtype=torch.float32): return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")
for (k,stride,c,m) in [(33,2,3.0,2)]:
    case = O.synthetic_case(203, k, stride, seed=k+m, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    P = case["yb"].shape[1]
    xa, fl, retry = eng.analysis_tiles(dev(case["state"]), rec, P, tiles, 1.1)
    xa = xa.cpu().numpy()
    ref = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1)[0]
    err = np.linalg.norm(xa-ref, axis=(1,2))/np.linalg.norm(ref, axis=(1,2))
    print(k,stride,c,m,'ut',tiles.ut,'per-row err', err, 'p_max', nb.p_max)
    e3 = np.abs(xa-ref).max(axis=1)     # (m, G)
    for r in range(m):
        bad = np.flatnonzero(e3[r] > 1e-4)
        print('  row', r, 'bad points', len(bad), bad[:40])

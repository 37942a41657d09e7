"""Weights on the tile route: accuracy over the test shapes + C2 timing (MIA_BUILD_FLAGS=-DMIA_W_TRIM=n selects the trim)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
from oracle import letkf_oracle as O
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
def D(a): return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
worst = 0.0
for (k, stride, c, m) in [(40, 2, 10.0, 1), (10, 1, 1.6, 1), (24, 2, 6.5, 2), (64, 2, 12.0, 1), (33, 2, 3.0, 1), (20, 4, 18.0, 1), (96, 3, 16.0, 1), (16, 2, 7.0, 1)]:
    case = O.synthetic_case(203, k, stride, seed=3 * k + m, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    if tiles.stats.tolist()[1]:
        tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
    for strength in (1.0, 3.0):
        yb, d = case["yb"] * strength, case["d"] * strength
        rec = eng.pack_split(D(yb), D(d))
        for inf in (1.0, 1.1):
            xa, W, fl, retry = eng.weights_tiles(D(case["state"]), rec, yb.shape[1], tiles, inf)
            if int(retry.item()):
                eng.weights_retry(D(case["state"]), D(yb), D(d), nb, inf, xa, W, fl)
            ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, c, inf)
            Wn = W.cpu().numpy()
            e = np.linalg.norm(Wn - ref_w) / np.linalg.norm(ref_w)
            ei = np.linalg.norm(Wn - ref_w) / np.linalg.norm(ref_w - np.eye(k))
            worst = max(worst, e)
            print("k=%2d p=%2d strength=%.0f inf=%.1f: W %.2e (W - I %.2e) retry %d meandeg %.1f" % (k, nb.p_max, strength, inf, e, ei, int(retry.item()), float((fl >> 8).float().mean())))
print("worst", worst)
G = 100000
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
tl = eng.localize_tiles(gx, ox, [10.0], 20)
srec = eng.pack_split(Yb, d)
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); res = eng.weights_tiles(X, srec, Yb.shape[1], tl, 1.1); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("C2 weights + analysis: %.4f ms" % float(np.median(ts[1:])))

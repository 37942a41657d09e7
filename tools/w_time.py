"""C2 timing of the weights route (mia_letkf_weights_tiles_f32) + error against the oracle at 48 points."""
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
G = 100000
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
tl = eng.localize_tiles(gx, ox, [10.0], 20)
srec = eng.pack_split(Yb, d)
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); res = eng.weights_tiles(X, srec, Yb.shape[1], tl, 1.1); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
pts = np.random.RandomState(3).choice(G, 48, replace=False)
yb_h, d_h = Yb.double().cpu().numpy(), d.double().cpu().numpy()
gxh, oxh = gx.cpu().numpy(), ox.cpu().numpy()
refw = np.stack([O.localized_weights(O.abs_distance_1d(gxh[g], oxh), yb_h, d_h, [10.0], 1.1) for g in pts])
gotw = res[1][torch.as_tensor(pts, device=dev)].double().cpu().numpy()
print("C2 weights + analysis: %.4f ms; error vs oracle %.2e; declined %d" % (float(np.median(ts[1:])), np.linalg.norm(gotw - refw) / np.linalg.norm(refw), int(res[3].item())))

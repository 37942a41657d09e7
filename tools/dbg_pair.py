import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
from oracle import letkf_oracle as O
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G = 3000
case = O.synthetic_case(G, 40, 2)
X = torch.as_tensor(case["state"], dtype=torch.float32, device=dev)
Yb = torch.as_tensor(case["yb"], dtype=torch.float32, device=dev); d = torch.as_tensor(case["d"], dtype=torch.float32, device=dev)
gx, ox = case["grid_x"], case["obs_x"]
rec = eng.pack_split(Yb, d)
outs = {}
for extra in (0, 1, 2):
    tiles = eng.localize_tiles(gx, ox, [10.0], 20, extra_blocks=extra)
    xa, fl, retry = eng.analysis_tiles(X, rec, Yb.shape[1], tiles, 1.1)
    outs[extra] = xa.clone()
    hdr, idx, D = tiles.unpack()[:3]
    print("extra", extra, "kernel", _cabi.last_analysis_kernel(), "stats", tiles.stats.tolist(), "max U", int(hdr[:, 0].max()))
    outs[("D", extra)] = (hdr, idx, D)
for extra in (1, 2):
    a, b = outs[0], outs[extra]
    ne = a != b
    print("extra", extra, "vs 0: differing", int(ne.sum()), "max abs", float((a - b).abs().max()))
h0, i0, D0 = outs[("D", 0)]
for extra in (1, 2):
    h, i, D = outs[("D", extra)]
    print("extra", extra, "hdr equal", bool((h[:, :3] == h0[:, :3]).all()), "idx first 32 equal", bool((i[:, :32] == i0).all()),
          "D first two blocks equal", bool((D[:, :2] == D0).all()), "max |D diff|", float(np.abs(D[:, :2] - D0).max()), "rest zero", bool((D[:, 2:] == 0).all()))

"""Fused localisation (option tile_fused, letkf_tile2f.hip) against lists in memory, through the step driver: results compared
element by element, serial step time."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
import bench
mia.build()
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
case = bench.make_case(G, 40, 2, dev)
outs = {}
for fused in (0, 1):
    _cabi.set_option("tile_fused", fused)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(5):
        out = r.assimilate(*case)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        out = r.assimilate(*case)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    outs[fused] = out.clone()
    print("tile_fused=%d: kernel %s, serial step %.4f ms, flags ok %s" % (fused, r.dominant_kernel_name, ms, r.last_flags_ok()))
a, b = outs[0], outs[1]
ne = (a != b)
print("elements that differ: %d of %d; relative Frobenius difference %.3e; largest |diff| %.3e" %
      (int(ne.sum()), a.numel(), float(torch.linalg.norm(a - b) / torch.linalg.norm(a)), float((a - b).abs().max())))
if int(ne.sum()):
    cols = ne.any(dim=0).any(dim=0).nonzero().flatten()
    print("grid points that differ: %d, first %s" % (cols.numel(), cols[:12].tolist()))

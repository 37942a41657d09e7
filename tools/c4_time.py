"""Config 4 (k = 80, ~63 local obs) on the tile route: kernel time and error against the oracle at 32 points."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(100000, 80, 1, dev, seed=43)
rec = bench.tile_route_case(eng, X, gx, ox, Yb, d, 16.5, 1.1, n_check=32)
print({k: v for k, v in rec.items()})

"""Fused kernel (letkf_tile2f_kernel) duration against the number of tiles: is the launch bound by rounds of resident wavefronts
(5 per SIMD = 5120 tiles per round) or by the work?  Kernel time by the dispatch's own start / stop events, serial steps."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
for G in [int(a) for a in sys.argv[1:]] or [40960, 81920, 90000, 98304, 100000, 110000, 131072, 163840, 200000]:
    case = bench.make_case(G, 40, 2, dev)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, fuse_tile_lists=True)
    for _ in range(5):
        r.assimilate(*case)
    r.kernel_timings.clear()
    for _ in range(40):
        r.time_next_step()
        r.assimilate(*case)
    ts = sorted(a.elapsed_time(b) for a, b in r.kernel_timings)
    print("G %7d  tiles %6d  rounds of 5120: %.2f  kernel %s  median %.4f ms  min %.4f  per 1e5 points %.4f ms" %
          (G, (G + 15) // 16, (G + 15) // 16 / 5120.0, r.dominant_kernel_name, ts[len(ts) // 2], ts[0], ts[len(ts) // 2] * 1e5 / G))
    r.close()

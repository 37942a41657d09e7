"""A/B of the two-waves-per-tile kernel (letkf_tile2p.hip, option tile_pair) against letkf_tile2_kernel on unions of more than
32 slots: config 4 and the 316 x 316 mesh; kernel time by HIP events around 20 launches, results compared."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
import bench
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
cases = {"c4": (bench.make_case(100000, 80, 1, dev, seed=43), 16.5, 0),
         "mesh_2d": (bench.make_case_2d(316, 316, 40, 2, dev, seed=44), 2.5, 2),
         "c4_m4": ((lambda c: (c[0].repeat(4, 1, 1).contiguous(),) + c[1:])(bench.make_case(100000, 80, 1, dev, seed=43)), 16.5, 0)}
for name, ((X, g, o, Yb, d), rad, extra) in cases.items():
    nb = eng.localize(g, o, [rad])
    tiles = eng.localize_tiles(g, o, [rad], nb.p_max, extra_blocks=extra)
    assert int(tiles.stats[1].item()) == 0
    srec = eng.pack_split(Yb, d)
    P = int(Yb.shape[1])
    out = {}
    for pair in (0, 1):
        _cabi.set_option("tile_pair", pair)
        out_buf = torch.empty((X.shape[0], X.shape[1], X.shape[2]), dtype=torch.float32, device=dev)
        xa, fl, retry = eng.analysis_tiles(X, srec, P, tiles, 1.1, out=out_buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.analysis_tiles(X, srec, P, tiles, 1.1, out=out_buf)
        e1.record()
        torch.cuda.synchronize()
        out[pair] = (e0.elapsed_time(e1) / 20, xa.clone())
    diff = float(torch.linalg.norm(out[0][1] - out[1][1]) / torch.linalg.norm(out[0][1]))
    print("%-8s one wave per tile %.4f ms, two %.4f ms (ratio %.2f); results differ by %.1e" % (name, out[0][0], out[1][0], out[1][0] / out[0][0], diff))

"""Experiment builds: the fused kernel's waves of one SIMD started apart (MIA_TILE2_STAGGER, units of 64 cycles per wave slot),
serial steps at config 2.  MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS python tools/fused_stagger.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
case = bench.make_case(100000, 40, 2, dev)
r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, fuse_tile_lists=True)
for _ in range(10):
    ref = r.assimilate(*case)
ref = ref.clone()
for st in (0, 4, 8, 16, 32, 64, 128, 0):
    os.environ["MIA_TILE2_STAGGER"] = str(st)
    for _ in range(5):
        r.assimilate(*case)
    r.kernel_timings.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        if i % 4 == 0:
            r.time_next_step()
        out = r.assimilate(*case)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print("stagger %3d: serial step %.4f ms, kernel %.4f ms, same result %s" % (st, ms, r.kernel_ms() or -1, bool(torch.equal(out, ref))), flush=True)

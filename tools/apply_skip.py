"""Experiment builds: csrc/apply_local.hip with parts switched off (MIA_APPLY_SKIP bits: 1 products, 2 loads of x, 4 stores of xa,
8 loads of W), one process.  MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS python tools/apply_skip.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G, k = 100000, 40
gen = torch.Generator(device=dev); gen.manual_seed(1)
W = torch.randn((G, k, k), generator=gen, device=dev) / k ** 0.5
for m in (16, 64):
    X = torch.randn((m, k, G), generator=gen, device=dev)
    for use16 in ("0",):
        for skip in (0, 1, 2, 4, 8, 3, 7, 15, 0):
            os.environ["MIA_APPLY_SKIP"] = str(skip)
            eng.apply_local_weights(X, W); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = eng.apply_local_weights(X, W)
            e1.record(); torch.cuda.synchronize()
            print("m = %3d  skip %2d: %.3f ms" % (m, skip, e0.elapsed_time(e1) / 5), flush=True)

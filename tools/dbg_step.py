import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
mia.build()
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
for bucket in (1, 0):
    _cabi.set_option("bucket_index", bucket)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=6)
    calls = {"n": 0}
    orig = r._assimilate_native
    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    r._assimilate_native = counted
    for _ in range(5):
        r.assimilate(X, gx, ox, Yb, d)
    torch.cuda.synchronize(); calls["n"] = 0
    t0 = time.perf_counter()
    for _ in range(200):
        r.assimilate(X, gx, ox, Yb, d)
    torch.cuda.synchronize()
    print("bucket", bucket, "serial ms/step %.4f" % ((time.perf_counter() - t0) / 200 * 1e3), "native steps", r.native_steps, "assimilate_native calls", calls["n"],
          "scan", r._scan_index, "extra", r._tile_extra, "no_tl", r._no_tile_lists)
    import collections
    pend = collections.deque()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(600):
        pend.append(r.submit(X, gx, ox, Yb, d))
        if len(pend) == 6: pend.popleft().result()
    while pend: pend.popleft().result()
    torch.cuda.synchronize()
    print("   pipelined ms/step %.4f" % ((time.perf_counter() - t0) / 600 * 1e3), "calls", calls["n"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        r.submit(X, gx, ox, Yb, d).result()
    torch.cuda.synchronize()
    print("   submit().result() one at a time ms/step %.4f" % ((time.perf_counter() - t0) / 200 * 1e3))

#!/usr/bin/env python3
"""HIP-event timing of the row-f3 kernels on the C2 geometry: mia_lienks_update_f32 (transform, tau 1 and 0.8; bundle)
and mia_apply_local_weights_f32, 1e5 grid points, k = 40, <= 20 local observations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch_assimilate_amd as mia  # noqa: E402

dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
nb = eng.localize(gx, ox, [10.0])
rec = eng.pack_obs(Yb, d, torch.float32)
W0 = torch.eye(40, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


t1, W = timed(lambda: eng.ienks_update(W0, None, None, nb, 1.0, None, rec=rec))
t2, W2 = timed(lambda: eng.ienks_update(W, None, None, nb, 0.8, None, rec=rec))
rec_b = eng.pack_obs(Yb * 1e-3, d, torch.float32)      # bundle variant: the perturbations arrive scaled by epsilon
t3, W3 = timed(lambda: eng.ienks_update(W0, None, None, nb, 1.0, 1e-3, rec=rec_b))
t3b, _ = timed(lambda: eng.ienks_update(W3, None, None, nb, 1.0, 1e-3, rec=rec_b))
t1e, _ = timed(lambda: eng.ienks_update(W0, None, None, nb, 1.0, None, rec=rec, method="eig"))
t4, xa = timed(lambda: eng.apply_local_weights(X, W))
wbytes = G * 40 * 40 * 4
print("grid points %d, k = 40, local obs <= %d" % (G, nb.p_max))
print("lienks_update transform tau=1.0 (shared prior weights in):  %.3f ms  (%.2e updates/s)" % (t1, G / t1 * 1e3))
print("lienks_update transform tau=0.8 (per-point weights in):     %.3f ms  (%.2e updates/s)" % (t2, G / t2 * 1e3))
print("lienks_update bundle    tau=1.0 eps=1e-3, first iteration:  %.3f ms  (%.2e updates/s)" % (t3, G / t3 * 1e3))
print("lienks_update bundle    tau=1.0 eps=1e-3, second iteration: %.3f ms  (%.2e updates/s)" % (t3b, G / t3b * 1e3))
print("lienks_update transform tau=1.0 through the general kernel: %.3f ms  (%.2e updates/s)" % (t1e, G / t1e * 1e3))
print("apply_local_weights (m=1):                                  %.3f ms  (W stream %.0f GB/s of 8000 peak)" % (t4, wbytes / t4 / 1e6))

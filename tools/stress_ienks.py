#!/usr/bin/env python3
"""Randomised check of the localised IEnKS update and the kernel-expression LKETKF against the CPU oracle
(test infrastructure): ensemble sizes incl. odd ones, sparse / dense local lists, tau, both variants, two chained
iterations; random kernel compositions.  float64 <= 1e-8, float32 <= 2e-4 on weights."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia  # noqa: E402
from torch_assimilate_amd import kernels as K  # noqa: E402
from oracle import letkf_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
worst = {}
for case in range(n_cases):
    k = int(rs.choice([5, 7, 10, 24, 40, 64]))
    G = 12
    P = int(rs.choice([6, 40, 300]))
    c = float(rs.choice([0.05, 0.3]))
    tau = float(rs.choice([1.0, 1.0, 0.6]))
    eps = None if rs.rand() < 0.5 else 1e-3
    grid, obs = rs.uniform(0, 1, size=G), rs.uniform(0, 1, size=P)
    hx = rs.normal(size=(k, P)) * 0.5
    yb, d = hx - hx.mean(axis=0), rs.normal(size=P) * 0.5
    scale = 1.0 if eps is None else eps
    nb = eng.localize(grid, obs, [c])
    tag = "k%d P%d pmax%d c%.2f tau%.1f eps%s" % (k, P, nb.p_max, c, tau, eps)
    for dtype, name in ((torch.float64, "ienks f64"), (torch.float32, "ienks f32")):
        w_ref = np.eye(k)
        w = torch.eye(k, dtype=dtype, device=dev)
        try:
            for it in range(2):
                w_ref = O.lienks_weights(w_ref, grid, obs, yb * scale, d, c, tau, eps)
                w = eng.ienks_update(w, torch.as_tensor(yb * scale, dtype=dtype, device=dev), torch.as_tensor(d, dtype=dtype, device=dev),
                                     nb, tau, eps)
        except mia.MiaError as err:
            print("skipped (unsupported shape):", tag, str(err)[:50])
            continue
        e = rel(w.cpu().numpy(), w_ref)
        if e > worst.get(name, (0, ""))[0]:
            worst[name] = (e, tag)
    # a random kernel composition through the expression route
    kerns = [(K.PolyKernel(2.0, 1.0), lambda x, y: O.poly_kernel(x, y, 2.0, 1.0)),
             (K.OrnsteinUhlenbeckKernel(5.0), lambda x, y: O.orn_uhl_kernel(x, y, 5.0)),
             (K.RationalKernel(2.0, 1.5), lambda x, y: O.rational_kernel(x, y, 2.0, 1.5)),
             (K.TanhKernel(0.05, 0.1), lambda x, y: O.tanh_kernel(x, y, 0.05, 0.1))]
    (ka, oa), (kb, ob) = kerns[rs.randint(4)], kerns[rs.randint(4)]
    comp = rs.randint(3)
    kern = [ka + kb, ka * kb, ka][comp]
    ofun = [lambda x, y: oa(x, y) + ob(x, y), lambda x, y: oa(x, y) * ob(x, y), oa][comp]
    state = rs.normal(size=(1, k, G))
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, c, 1.1, core=lambda a, b, inf: O.ketkf_weights(a, b, ofun, inf))
    try:
        a = mia.LKETKF(kern, localization=mia.GaspariCohn(c, mia.AbsoluteDistance()), inf_factor=1.1, dtype=torch.float64, engine=eng)
        xa = a.analyse_arrays(state, yb, d, grid_coords=grid, obs_coords=obs)
        e = rel(xa.cpu().numpy(), ref)
        if e > worst.get("lketkf f64", (0, ""))[0]:
            worst["lketkf f64"] = (e, tag + " " + str(kern))
    except mia.MiaError as err:
        print("skipped (unsupported shape):", tag, str(err)[:50])
bad = False
lim = {"ienks f64": 1e-8, "ienks f32": 2e-4, "lketkf f64": 1e-8}
for name, (e, tag) in sorted(worst.items()):
    print("%-11s worst %.2e  at %s" % (name, e, tag))
    bad |= e > lim[name]
sys.exit(1 if bad else 0)

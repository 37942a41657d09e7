#!/usr/bin/env python3
"""Randomised cross-check of every float32 analysis route against the float64 kernel (itself pinned to the oracle by the
parity tests): random ensemble sizes (odd ones, > 64), observation densities from empty to over-full lists (dual and
primal routes), 1-D / 2-D geometry, 1..40 state rows, inflation, with and without weights, linear and RBF cores.
Prints the worst relative Frobenius error per route; exits non-zero above 1.5e-5 on an analysis or 2e-4 on raw weights
(float32 weights of a k = 7 ensemble under 1500 accurate local observations were the worst case seen: 1e-4)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia  # noqa: E402

dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = {}


def rel(a, b):
    return float(torch.linalg.norm(a.double() - b.double()) / max(float(torch.linalg.norm(b.double())), 1e-300))


for case in range(n_cases):
    k = int(rs.choice([5, 7, 10, 12, 20, 24, 33, 40, 48, 64, 65, 80, 96]))
    G = int(rs.choice([37, 200, 513]))
    nc = int(rs.choice([1, 2]))
    m = int(rs.choice([1, 2, 3, 8, 17, 40]))
    P = int(rs.choice([0, 3, 50, 400, 1500]))
    inf = float(rs.choice([1.0, 1.1, 1.5]))
    gamma = float(rs.choice([0.0, 0.0, 0.5, 2.0]))
    grid = rs.uniform(0, 1, size=(G, nc))
    obs = rs.uniform(0, 1, size=(P, nc))
    c = float(rs.choice([0.02, 0.08, 0.3]))
    scale = float(rs.choice([0.2, 1.0, 3.0]))
    X = rs.normal(size=(m, k, G))
    hx = rs.normal(size=(k, P)) * scale
    yb = hx - (hx.mean(axis=0) if P else 0.0)
    d = rs.normal(size=P) * scale
    nb = eng.localize(grid, obs, [c])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    kw = dict(rbf_gamma=gamma if gamma > 0 else None)
    try:
        ref_x, ref_w = eng.analysis(t(X, torch.float64), t(yb, torch.float64), t(d, torch.float64), nb, inf, return_weights=True, **kw)
    except mia.MiaError as err:        # local block beyond a workgroup's LDS (p_max * k too large): a documented limit
        skipped = globals().get("skipped", 0) + 1
        print("skipped (unsupported shape): k%d pmax%d: %s" % (k, nb.p_max, str(err)[:60]))
        continue
    tag = "k%d G%d nc%d m%d P%d pmax%d inf%.1f g%.1f sc%.1f" % (k, G, nc, m, P, nb.p_max, inf, gamma, scale)
    runs = {}
    runs["eig"] = eng.analysis(t(X, torch.float32), t(yb, torch.float32), t(d, torch.float32), nb, inf, method="eig", **kw)
    runs["auto"] = eng.analysis(t(X, torch.float32), t(yb, torch.float32), t(d, torch.float32), nb, inf, **kw)
    xw, W = eng.analysis(t(X, torch.float32), t(yb, torch.float32), t(d, torch.float32), nb, inf, return_weights=True, **kw)
    runs["weights.xa"] = xw
    for name, out in runs.items():
        e = rel(out, ref_x)
        if e > worst.get(name, (0, ""))[0]:
            worst[name] = (e, tag)
    e = rel(W, ref_w)
    if e > worst.get("weights.W", (0, ""))[0]:
        worst["weights.W"] = (e, tag)
    if not torch.isfinite(runs["auto"]).all():
        print("NON-FINITE", tag)
        sys.exit(2)
bad = False
for name, (e, tag) in sorted(worst.items()):
    print("%-11s worst %.2e  at %s" % (name, e, tag))
    bad |= e > (2e-4 if name == 'weights.W' else 1.5e-5)
sys.exit(1 if bad else 0)

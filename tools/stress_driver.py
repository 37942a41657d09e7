#!/usr/bin/env python3
"""Randomised check of the one-call step driver: ShardedLetkf (serial call, steps in flight) against the entry-by-entry
engine route on random geometries -- ensemble sizes, 1-D / 2-D meshes, sparse to dense networks (dual and primal
routes), 1..20 state rows, RBF core, changing inputs between steps.  Any difference above 2e-6 (both run the same
kernels; the driver sizes its lists from the previous step) or a flagged grid point fails."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia  # noqa: E402

dev = torch.device("cuda:0")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = (0.0, "")
for case in range(n_cases):
    k = int(rs.choice([6, 10, 24, 40, 64]))
    G = int(rs.choice([100, 1000, 4001]))
    nc = int(rs.choice([1, 2]))
    m = int(rs.choice([1, 3, 9, 20]))
    P = int(rs.choice([0, 20, 500, 3000]))
    c = float(rs.choice([0.02, 0.1]))
    gamma = None if rs.rand() < 0.7 else 0.5
    inf = float(rs.choice([1.0, 1.2]))
    grid = rs.uniform(0, 1, size=(G, nc))
    obs = rs.uniform(0, 1, size=(P, nc))
    tag = "k%d G%d nc%d m%d P%d c%.2f g%s" % (k, G, nc, m, P, c, gamma)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    runner = mia.ShardedLetkf(dev, 0, 1, radii=[c], inf_factor=inf, rbf_gamma=gamma, max_in_flight=3)
    ref_runner = mia.ShardedLetkf(dev, 0, 1, radii=[c], inf_factor=inf, rbf_gamma=gamma, native_step=False)
    inputs = []
    for step in range(5):
        X = rs.normal(size=(m, k, G))
        hx = rs.normal(size=(k, P)) * 0.5
        inputs.append((t(X), t(grid, torch.float64), t(obs, torch.float64), t(hx - (hx.mean(axis=0) if P else 0.0)), t(rs.normal(size=P) * 0.5)))
    refs = [ref_runner.assimilate(*a).clone() for a in inputs]
    outs = [runner.assimilate(*inputs[0]), runner.assimilate(*inputs[1])]
    pend = [runner.submit(*a) for a in inputs[2:]]
    outs += [h.result() for h in pend]
    if not runner.last_flags_ok():
        print("FLAGGED", tag)
        sys.exit(2)
    for o, r in zip(outs, refs):
        e = float(torch.linalg.norm(o.double() - r.double()) / max(float(torch.linalg.norm(r.double())), 1e-300))
        if e > worst[0]:
            worst = (e, tag)
print("driver vs engine: worst %.2e at %s (%d cases, native steps in the last case: %d)" % (worst[0], worst[1], n_cases, runner.native_steps))
sys.exit(1 if worst[0] > 2e-6 else 0)

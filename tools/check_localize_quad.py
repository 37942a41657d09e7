#!/usr/bin/env python3
"""localize_quad_kernel against localize_kernel: identical lists (counts, indices, weights bit for bit), 1-D / 2-D / 3-D,
one and two radii, ragged grid sizes; then their times alone on the GPU."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
mia.build()
eng = mia.LetkfEngine("cuda:0")
rs = np.random.RandomState(5)
cases = [(np.arange(100003.0)[:, None], np.arange(0, 100003, 2.0)[:, None], [10.0], [0]),
         (rs.uniform(0, 1, size=(5001, 2)), rs.uniform(0, 1, size=(3000, 2)), [0.05], [0, 0]),
         (rs.uniform(0, 1, size=(1500, 3)), rs.uniform(0, 1, size=(4000, 3)), [0.12, 0.25], [0, 0, 1]),
         (np.arange(7.0)[:, None], np.array([[3.0], [40.0]]), [2.0], [0])]
for grid, obs, radii, groups in cases:
    out = {}
    for qd in (1, 0):
        _cabi.set_option("localize_quad", qd)
        nb = eng.localize(grid, obs, radii, coord_group=groups)
        out[qd] = (nb.cnt.cpu().numpy(), nb.idx.cpu().numpy()[:, :nb.p_max], nb.w.cpu().numpy()[:, :nb.p_max], nb.p_max)
    same = all(np.array_equal(a, b) for a, b in zip(out[1][:3], out[0][:3])) and out[1][3] == out[0][3]
    if not same:
        (c1, i1, w1, _), (c0, i0, w0, _) = out[1], out[0]
        print("  cnt equal", np.array_equal(c1, c0), "idx equal", np.array_equal(i1, i0), "w equal", np.array_equal(w1, w0),
              "max |dw|/w", float(np.abs(w1 - w0).max()), "n differing w", int((w1 != w0).sum()), "of", w1.size, "shapes", i1.shape, i0.shape)
    print("grid", grid.shape, "obs", obs.shape, "radii", radii, "p_max", out[1][3], "identical" if same else "DIFFERENT")
    assert same
grid, obs, radii, groups = cases[0]
for qd in (1, 0, 1, 0):
    _cabi.set_option("localize_quad", qd)
    for _ in range(3): eng.localize(grid, obs, radii, coord_group=groups)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.localize(grid, obs, radii, coord_group=groups)
    torch.cuda.synchronize()
    print("localize_quad = %d: %.1f us per call (index + lists, host included)" % (qd, (time.perf_counter() - t0) / 20 * 1e6))

"""A pool of CPU worker processes that evaluate the float64 oracle (oracle/letkf_oracle.py) at MANY grid points of a full-size
1-D benchmark case, so that the GPU tests can compare EVERY grid point of configuration 2 (1e5) and of the other configurations
with the reference's per-point algorithm instead of a sample (interface/test_letkf.py:106-157 checks every grid point too).

The workers are forked once, at session start, BEFORE the test process initialises the GPU (tests/conftest.py): a process that holds
a HIP context must neither fork workers that outlive its threads' locks nor exec.  Inputs travel as .npy files that the workers
memory-map (one copy in the page cache for all of them); tasks carry index ranges only.  Test infrastructure: nothing in the
product imports this."""
import multiprocessing as mp
import os
import shutil
import tempfile

import numpy as np

_POOL = None
_NPROC = 0
_CASES = {}        # per worker: directory -> memory-mapped arrays


def _cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def _init():
    import torch
    torch.set_num_threads(1)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"


def start():
    """Fork the workers (idempotent).  Call before anything touches the GPU."""
    global _POOL, _NPROC
    if _POOL is None:
        _NPROC = _cores()
        _POOL = mp.get_context("fork").Pool(_NPROC, initializer=_init)
    return _POOL


def started():
    return _POOL is not None


def stop():
    global _POOL
    if _POOL is not None:
        _POOL.terminate()
        _POOL.join()
        _POOL = None


def _load(path):
    if path not in _CASES:
        _CASES.clear()            # (one case at a time per worker)
        _CASES[path] = {n: np.load(os.path.join(path, n + ".npy"), mmap_mode="r") for n in ("state", "gx", "ox", "yb", "d")}
    return _CASES[path]


def _work(task):
    """Oracle analysis of the grid points `pts` of the 1-D case stored under `path`: the reference's per-point path --
    localize_obs on the |x - y| distance, mask / sqrt(rho), ETKF (or RBF-KETKF) weights in float64, _apply_weights -- over the
    observations within `reach` coordinate units (the taper's support is 2 c: the rest weighs exactly zero)."""
    path, pts, c, inf, gamma, reach = task
    from oracle import letkf_oracle as O
    a = _load(path)
    st, gx, ox, yb, d = a["state"], a["gx"], a["ox"], a["yb"], a["d"]
    core = O.etkf_weights if gamma is None else (lambda p, q, i, g_=gamma: O.ketkf_weights(p, q, lambda x, y: O.rbf_kernel(x, y, g_), i))
    out = np.empty((st.shape[0], st.shape[1], len(pts)), dtype=np.float64)
    for n, g in enumerate(pts):
        lo, hi = np.searchsorted(ox, [gx[g] - reach, gx[g] + reach])
        w = O.localized_weights(O.abs_distance_1d(gx[g], np.asarray(ox[lo:hi])), np.asarray(yb[:, lo:hi]), np.asarray(d[lo:hi]), [c], inf, core=core)
        out[:, :, n] = O.apply_weights(np.asarray(st[:, :, [g]]), w[None])[:, :, 0]
    return out


def oracle_analysis(state, gx, ox, yb, d, c, inf, pts, gamma=None, chunk=250):
    """(m, k, len(pts)) float64 oracle analysis at the grid points `pts` (sorted observation coordinates `ox` required)."""
    assert _POOL is not None, "oracle pool not started (tests/conftest.py starts it for -m gpu sessions)"
    ox = np.asarray(ox, dtype=np.float64)
    assert np.all(np.diff(ox) >= 0), "observation coordinates must be sorted"
    tmp = tempfile.mkdtemp(prefix="mia_oracle_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for n, arr in (("state", state), ("gx", gx), ("ox", ox), ("yb", yb), ("d", d)):
            np.save(os.path.join(tmp, n + ".npy"), np.ascontiguousarray(arr, dtype=np.float64))
        pts = np.asarray(pts, dtype=np.int64)
        tasks = [(tmp, pts[i:i + chunk], float(c), float(inf), gamma, 2.0 * float(c) + 1.0) for i in range(0, len(pts), chunk)]
        parts = _POOL.map(_work, tasks, chunksize=1)
        return np.concatenate(parts, axis=-1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def per_point_errors(got, ref):
    """Relative error of every grid point's (m, k) block and the relative Frobenius error of the whole."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    num = np.sqrt(((got - ref) ** 2).sum(axis=(0, 1)))
    den = np.sqrt((ref ** 2).sum(axis=(0, 1)))
    return num / np.maximum(den, 1e-300), float(np.linalg.norm(got - ref) / np.linalg.norm(ref))

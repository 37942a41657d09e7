"""Weight persistence (``weight_save_path``): mirror of BaseAssimilation.store_weights / load_weights
(pytassim/interface/base.py:280-325) and of what they call, save_netcdf / load_netcdf
(pytassim/utilities/xarray.py:36-173), for the weights of this path.

On-disk format: what ``xarray.DataArray.to_netcdf`` writes for the reference's weights when it falls back to its
scipy engine -- a netCDF-3 (64-bit offset) file with one data variable ``__xarray_dataarray_variable__`` of dims
``(grid, ensemble, ensemble_new)`` (global filters: ``(ensemble, ensemble_new)``), float64, plus one coordinate
variable per dimension; a multi-level grid index is flattened to ``arange`` with the level values as extra
coordinate variables and the attribute ``multidim_levels = "name1;name2"`` on the index (encode_multidim,
utilities/xarray.py:66-103).  ``xarray.open_dataarray`` + ``decode_multidim`` read it back.

Data movement: W (G, k, k) lives in HBM (1.28 GB at 1e5 points, k = 40, float64).  It is streamed device -> host in
chunks of grid points through two pinned staging buffers on a side stream, so the copy of chunk c+1 overlaps the
(byte-swapping) write of chunk c; loading streams the other way.  No floating-point work happens here.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["store_weights", "load_weights", "DATA_VARIABLE"]

DATA_VARIABLE = "__xarray_dataarray_variable__"


def _nc3(values) -> Tuple[str, np.ndarray]:
    """netCDF-3 type code and values of a coordinate: the classic format has no int64 (xarray's scipy engine stores
    such indexes as int32), no bool and no unicode."""
    a = np.asarray(values)
    if a.dtype.kind in "iub":
        if a.size and (a.min() < -2 ** 31 or a.max() >= 2 ** 31):
            raise ValueError("integer coordinate does not fit netCDF-3's int32")
        return "i4", a.astype(np.int32)
    if a.dtype.kind == "f":
        return ("f4", a) if a.dtype.itemsize == 4 else ("f8", a.astype(np.float64))
    raise ValueError("coordinate values must be numeric (got dtype %s)" % a.dtype)


def _chunks(n: int, step: int):
    for c0 in range(0, n, step):
        yield c0, min(n, c0 + step)


def store_weights(path: str, weights: torch.Tensor, grid_index=None, ensemble: Optional[Sequence] = None,
                  grid_levels: Optional[Dict[str, Sequence]] = None, chunk_points: int = 16384) -> None:
    """Write weights (G, k, k) [or (k, k)] to ``path``.  ``grid_index``: values of the ``grid`` coordinate (default
    arange); ``grid_levels``: {level name: values} of a multi-level grid index (then ``grid`` is stored as arange
    with ``multidim_levels``); ``ensemble``: member labels (default arange), also used for ``ensemble_new``
    (interface/letkf.py:146)."""
    from scipy.io import netcdf_file
    w = weights.detach()
    if w.dim() not in (2, 3) or w.shape[-1] != w.shape[-2]:
        raise ValueError("weights must be (grid, ensemble, ensemble_new) or (ensemble, ensemble_new)")
    k = w.shape[-1]
    G = w.shape[0] if w.dim() == 3 else None
    ens = np.arange(k) if ensemble is None else np.asarray(ensemble)
    if ens.shape != (k,):
        raise ValueError("one ensemble label per member")
    f = netcdf_file(path, "w", version=2)
    try:
        dims: Tuple[str, ...] = ("ensemble", "ensemble_new")
        f.createDimension("ensemble", k)
        f.createDimension("ensemble_new", k)
        if G is not None:
            f.createDimension("grid", G)
            dims = ("grid",) + dims
            code, vals = _nc3(np.arange(G) if (grid_levels or grid_index is None) else grid_index)
            if vals.shape != (G,):
                raise ValueError("one grid index value per grid point")
            gv = f.createVariable("grid", code, ("grid",))
            gv[:] = vals
            if grid_levels:
                gv.multidim_levels = ";".join(grid_levels)
                for name, lvals in grid_levels.items():
                    code, lvals = _nc3(lvals)
                    if lvals.shape != (G,):
                        raise ValueError("grid level %r needs one value per grid point" % name)
                    lv = f.createVariable(name, code, ("grid",))
                    lv[:] = lvals
        code, evals = _nc3(ens)
        for name in ("ensemble", "ensemble_new"):
            ev = f.createVariable(name, code, (name,))
            ev[:] = evals
        var = f.createVariable(DATA_VARIABLE, "f8", dims)
        if grid_levels:
            var.coordinates = " ".join(grid_levels)
        if G is None or not w.is_cuda:
            var[:] = w.cpu().numpy().astype(np.float64)
        else:
            _stream_out(w, var, chunk_points)
    finally:
        f.close()


def _stream_out(w: torch.Tensor, var, chunk_points: int) -> None:
    G, k, _ = w.shape
    step = max(1, min(int(chunk_points), G))
    side = torch.cuda.Stream(device=w.device)
    side.wait_stream(torch.cuda.current_stream(w.device))         # the weights are complete before the first copy
    bufs = [torch.empty((step, k, k), dtype=w.dtype, pin_memory=True) for _ in range(2)]
    evs = [torch.cuda.Event() for _ in range(2)]
    spans = list(_chunks(G, step))
    with torch.cuda.stream(side):
        bufs[0][:spans[0][1] - spans[0][0]].copy_(w[spans[0][0]:spans[0][1]], non_blocking=True)
        evs[0].record(side)
    for i, (c0, c1) in enumerate(spans):
        if i + 1 < len(spans):                                     # next chunk in flight while this one is written
            n0, n1 = spans[i + 1]
            with torch.cuda.stream(side):
                bufs[(i + 1) & 1][:n1 - n0].copy_(w[n0:n1], non_blocking=True)
                evs[(i + 1) & 1].record(side)
        evs[i & 1].synchronize()
        var[c0:c1] = bufs[i & 1][:c1 - c0].numpy()                # float -> big-endian float64 of the file
    torch.cuda.current_stream(w.device).wait_stream(side)


def load_weights(path: str, device=None, dtype: torch.dtype = torch.float64, chunk_points: int = 16384):
    """Read weights stored by :func:`store_weights` (or by the reference's store_weights through xarray's scipy
    engine).  Returns (weights tensor on ``device``, coords dict with 'grid', 'ensemble', 'ensemble_new' and, for a
    multi-level grid, one entry per level -- decode_multidim, utilities/xarray.py:139-173)."""
    from scipy.io import netcdf_file
    f = netcdf_file(path, "r", mmap=False)
    try:
        if DATA_VARIABLE in f.variables:
            var = f.variables[DATA_VARIABLE]
        else:
            cands = [v for n, v in f.variables.items() if n not in f.dimensions and len(v.dimensions) >= 2]
            if len(cands) != 1:
                raise ValueError("%s holds no unique weights variable" % path)
            var = cands[0]
        if tuple(var.dimensions[-2:]) != ("ensemble", "ensemble_new"):
            raise ValueError("weights variable must end in (ensemble, ensemble_new), found %r" % (var.dimensions,))
        coords = {}
        for name in var.dimensions:
            if name in f.variables:
                cv = f.variables[name]
                coords[name] = np.array(cv[:])
                levels = getattr(cv, "multidim_levels", None)
                if levels:
                    levels = levels.decode() if isinstance(levels, bytes) else levels
                    for lname in levels.split(";"):
                        coords[lname] = np.array(f.variables[lname][:])
                    coords["multidim_levels"] = levels.split(";")
        data = var.data
        dev = torch.device(device) if device is not None else torch.device("cpu")
        if dev.type != "cuda" or data.ndim == 2:
            out = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(device=dev, dtype=dtype)
        else:
            out = _stream_in(data, dev, dtype, chunk_points)
    finally:
        f.close()
    return out, coords


def _stream_in(data: np.ndarray, dev: torch.device, dtype: torch.dtype, chunk_points: int) -> torch.Tensor:
    G, k, _ = data.shape
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    out = torch.empty((G, k, k), dtype=dtype, device=dev)
    step = max(1, min(int(chunk_points), G))
    side = torch.cuda.Stream(device=dev)
    # `out` comes from the caching allocator on the current stream: work enqueued there that still uses the block's
    # previous contents must finish before the side stream writes into it
    side.wait_stream(torch.cuda.current_stream(dev))
    out.record_stream(side)
    bufs = [torch.empty((step, k, k), dtype=dtype, pin_memory=True) for _ in range(2)]
    evs = [None, None]
    for i, (c0, c1) in enumerate(_chunks(G, step)):
        b = bufs[i & 1]
        if evs[i & 1] is not None:
            evs[i & 1].synchronize()                               # the buffer's previous upload has left
        b[:c1 - c0].numpy()[...] = data[c0:c1].astype(np_dtype, copy=False)   # big-endian file -> native staging
        with torch.cuda.stream(side):
            out[c0:c1].copy_(b[:c1 - c0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
            evs[i & 1] = ev
    torch.cuda.current_stream(dev).wait_stream(side)
    return out

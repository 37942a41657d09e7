"""xarray-in / xarray-out adapter: pytassim's ``assimilate()`` flow around the gfx950 engine.

Follows BaseAssimilation.assimilate (pytassim/interface/base.py:419-512) and
FilterAssimilation.update_state (interface/filter.py:96-165); what stays xarray in the reference
stays xarray here, only estimate_weights + _apply_weights are replaced by the fused GPU call.
xarray (and pandas) are imported lazily: they are absent from the build and GPU images.
"""
from __future__ import annotations

import logging
import time
import warnings

import numpy as np

logger = logging.getLogger(__name__)


def _total_seconds(index):
    """pytassim.utilities.pandas.dtindex_to_total_seconds: seconds since 1970-01-01."""
    import pandas as pd
    return np.asarray((pd.DatetimeIndex(index) - pd.Timestamp(1970, 1, 1)).total_seconds(), dtype=np.float64)


def _index_to_array(index):
    """pytassim.utilities.pandas.index_to_array (pandas.py:70-102): MultiIndex -> (n, levels)."""
    import pandas as pd
    if isinstance(index, pd.MultiIndex):
        return np.array(index.tolist(), dtype=np.float64)
    vals = np.asarray(index)
    if vals.dtype == object:
        vals = np.array([np.atleast_1d(v) for v in vals], dtype=np.float64)
    vals = vals.astype(np.float64)
    return vals.reshape(len(vals), -1)


def _rcinv_normalise(ds, value):
    """Observation.mul_rcinv (observation.py:241-295) for one value array with an obs_grid_1 axis."""
    cov = ds["covariance"]
    if "obs_grid_2" in cov.dims:                          # correlated: right-multiply by inv(chol(R).T)
        import xarray as xr
        if "time" in cov.dims:
            parts = []
            for t in range(cov.sizes["time"]):
                ci = np.linalg.inv(np.linalg.cholesky(cov.isel(time=t).values).T)
                parts.append(cov.isel(time=t).copy(data=ci))
            cinv = xr.concat(parts, dim="time")
        else:
            cinv = cov.copy(data=np.linalg.inv(np.linalg.cholesky(cov.values).T))
        out = xr.dot(value, cinv, dims="obs_grid_1").rename({"obs_grid_2": "obs_grid_1"})
        return out.assign_coords(obs_grid_1=value["obs_grid_1"])
    return value / np.sqrt(cov)                            # uncorrelated: 1 / sqrt(var)


def obs_space_variables(ens_obs, observations):
    """_get_obs_space_variables + _stack_obs (base.py:359-379, 223-241) -> Yb (k, P), d (P,),
    obs coordinate table (P, 1 + n_coord) with column 0 = time in seconds."""
    ybs, ds_, coords = [], [], []
    for ens, obs in zip(ens_obs, observations):
        mean = ens.mean("ensemble")
        perts = ens - mean
        innov = _rcinv_normalise(obs, obs["observations"] - mean).transpose("time", "obs_grid_1")
        perts = _rcinv_normalise(obs, perts).transpose("ensemble", "time", "obs_grid_1")
        k = perts.sizes["ensemble"]
        ybs.append(np.asarray(perts.values, dtype=np.float64).reshape(k, -1))
        ds_.append(np.asarray(innov.values, dtype=np.float64).reshape(-1))
        t = _total_seconds(innov.indexes["time"])
        g = _index_to_array(innov.indexes["obs_grid_1"])
        coords.append(np.hstack([np.repeat(t, len(g))[:, None], np.tile(g, (len(t), 1))]))
    return np.concatenate(ybs, axis=1), np.concatenate(ds_), np.concatenate(coords, axis=0)


def assimilate(algo, state, observations, pseudo_state=None, analysis_time=None):
    import xarray as xr
    start = time.time()
    if not isinstance(state, xr.DataArray):
        raise TypeError("*** Given state is not a valid {0} ***\n{1:s}".format(type(xr.DataArray), str(state)))
    if isinstance(observations, xr.Dataset):
        observations = (observations,)
    if not observations:
        warnings.warn("No observation is given, I will return the background state!", UserWarning)
        return state
    for p in (algo.pre_transform or ()):
        state, observations, pseudo_state = p.pre(state, observations, pseudo_state)
    if analysis_time is None:
        analysis_time = state.time[-1].values              # base.py:154-179 (latest state time)
    pseudo = state if pseudo_state is None else pseudo_state
    back = state if algo.smoother else state.sel(time=[analysis_time])
    pseudo = pseudo if algo.smoother else pseudo  # the obs operator sees the full pseudo state (filter.py:140-150)
    ens_obs, used = [], []
    for obs in observations:                               # base.py:181-220
        try:
            ens_obs.append(obs.obs.operator(obs, pseudo))
            used.append(obs)
        except (AttributeError, NotImplementedError) as err:
            raise NotImplementedError("observation subset without a usable `.obs.operator`") from err
    yb, d, obs_tab = obs_space_variables(ens_obs, used)
    grid = _index_to_array(back.indexes["grid"])
    t0 = _total_seconds(back.indexes["time"][:1])[0]
    grid_info = np.hstack([np.full((len(grid), 1), t0), grid])     # mixin_local.py:55-58
    st = np.asarray(back.transpose("var_name", "time", "ensemble", "grid").values)
    xa = algo.analyse_arrays(st, yb, d, grid_coords=grid, obs_coords=obs_tab[:, 1:],
                             grid_info=grid_info, obs_info=obs_tab)
    analysis = back.transpose("var_name", "time", "ensemble", "grid").copy(
        data=xa.cpu().numpy().astype(st.dtype)).transpose(*back.dims)
    for p in (algo.post_transform or ()):
        analysis = p.post(analysis, state, observations, pseudo_state)
    logger.info("Finished assimilation after {0:.2f} s".format(time.time() - start))
    return analysis

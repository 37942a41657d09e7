"""xarray-in / xarray-out shim over :mod:`assim_flow`: converts pytassim's state ``DataArray`` and observation
``Dataset`` objects to the array-level data model, runs ``assimilate_arrays`` (which carries ALL the semantics of
interface/base.py:419-512 and interface/filter.py:39-165) and wraps the analysis back.  xarray is imported lazily:
it is absent from the build and GPU images, so everything with behaviour lives in ``assim_flow`` where it is tested.
"""
from __future__ import annotations

import numpy as np

from . import assim_flow as flow


def _grid_table(index):
    """utilities/pandas.py:70-102 ``index_to_array``: (Multi)Index -> float (n, levels)."""
    vals = np.asarray(index)
    if vals.dtype == object:
        vals = np.array([np.atleast_1d(v) for v in vals], dtype=np.float64)
    return vals.astype(np.float64).reshape(len(vals), -1)


def _state(da):
    """DataArray -> ModelState (dims are carried as they are: validity is judged by the flow, base.py:129-137)."""
    if da is None:
        return None
    return flow.ModelState(np.asarray(da.values), da.indexes["time"].values, _grid_table(da.indexes["grid"]),
                           dims=tuple(da.dims), source=da,
                           ensemble=da.indexes["ensemble"].values if "ensemble" in da.dims else None)


def _xr_of(st):
    """The xarray view of a (possibly time-sliced) ModelState that came through :func:`_state`."""
    return st.source.isel(time=list(st.time_index))


def _subset(ds):
    names = list(getattr(ds.indexes["obs_grid_1"], "names", None) or ["obs_grid_1"])
    cov = ds["covariance"]
    op = getattr(getattr(ds, "obs", None), "operator", None)        # pytassim's accessor, when registered

    def operator(sub, pseudo):            # the reference calls obs.obs.operator(sliced obs, sliced pseudo state)
        if op is None:
            raise NotImplementedError("No observation operator is set!")
        hx = op(ds.isel(time=list(sub.time_index)), _xr_of(pseudo))
        return np.asarray(hx.transpose("ensemble", "time", "obs_grid_1").values)

    return flow.ObsSubset(ds["observations"].values, cov.values, ds.indexes["time"].values,
                          _grid_table(ds.indexes["obs_grid_1"]), operator, correlated="obs_grid_2" in cov.dims,
                          cov_has_time="time" in cov.dims, grid_names=[n or "obs_grid_1" for n in names])


class _XrAlgo:
    """The algorithm as the flow sees it: user callables (forward model, transforms) keep receiving xarray objects."""

    def __init__(self, algo):
        self._algo = algo
        self.pre_transform = self.post_transform = None      # applied by assimilate() below, on xarray objects

    def __getattr__(self, name):
        return getattr(self._algo, name)

    @property
    def forward_model(self):
        fm = self._algo.forward_model
        if fm is None:
            return None
        return lambda state, iter_num: (None, _state(fm(_xr_of(state), iter_num)[1]))


def assimilate(algo, state, observations, pseudo_state=None, analysis_time=None):
    import xarray as xr
    if not observations:                                     # base.py:478-481 (before any validation)
        return flow.assimilate_arrays(algo, state, ())
    if not isinstance(observations, (list, set, tuple)):
        observations = (observations,)
    if not isinstance(state, xr.DataArray):
        raise TypeError("*** Given state is not a valid ``xarray.DataArray`` ***\n{0}".format(type(state)))
    for obs in observations:
        if not isinstance(obs, xr.Dataset):
            raise TypeError("*** Given observation is not a valid ``xarray.Dataset`` ***\n{0}".format(obs))
    st, subs = _state(state), [_subset(o) for o in observations]
    flow.validate_state(st)
    flow.validate_observations(subs)
    t_ana = flow.get_analysis_time(st, analysis_time)
    for trans in (algo.pre_transform or ()):                 # base.py:493-497: after validation and analysis time
        state, observations, pseudo_state = trans.pre(state, observations, pseudo_state)
        st, subs = _state(state), [_subset(o) for o in observations]
    ana = flow.update_state(_XrAlgo(algo), st, subs, _state(pseudo_state), t_ana)
    vals = ana.values.cpu().numpy() if hasattr(ana.values, "cpu") else np.asarray(ana.values)
    analysis = _xr_of(ana).copy(data=vals.astype(state.dtype))
    for trans in (algo.post_transform or ()):
        analysis = trans.post(analysis, state, observations, pseudo_state)
    flow.validate_state(_state(analysis))
    return analysis

"""Array-level ``assimilate()`` flow: everything pytassim does between the user's call and the per-grid-point
weights, on plain arrays, so that it can be tested without xarray and driven by the thin shim in ``xr_adapter``.

Mirrors, step by step (citations into /root/reference/pytassim):

* ``BaseAssimilation.assimilate``            interface/base.py:419-512  -> :func:`assimilate_arrays`
* ``_validate_state / _validate_observations`` interface/base.py:129-151 -> :func:`validate_state`, :func:`validate_observations`
* ``_get_analysis_time``                     interface/base.py:154-179  -> :func:`get_analysis_time`
* ``FilterAssimilation._slice_analysis``     interface/filter.py:39-55  -> :func:`slice_analysis`
* ``get_pseudo_state / propagate_model``     interface/base.py:331-357  -> :func:`get_pseudo_state`
* ``_apply_obs_operator``                    interface/base.py:181-220  -> :func:`apply_obs_operator`
* ``_get_obs_space_variables / _stack_obs``  interface/base.py:359-379, 223-241 -> :func:`obs_space_blocks`, :func:`obs_space_variables`
* ``_extract_state_information``             interface/mixin_local.py:50-69 -> :func:`state_information`
* ``FilterAssimilation.update_state``        interface/filter.py:96-165 -> :func:`update_state`

The data model is the reference's, minus xarray: :class:`ModelState` carries ``(var_name, time, ensemble, grid)``
values with a time axis and a grid coordinate table, :class:`ObsSubset` carries ``observations (time, obs_grid_1)``,
``covariance`` and an optional observation operator.  Times are float seconds since 1970-01-01 (what the reference
itself stacks into ``obs_id``, utilities/pandas.py:28-45) or anything ``numpy.datetime64`` understands.
Nothing here touches the GPU: the numerical work is delegated to the ``algo`` object (``get_obs_space_variables``,
``analyse_arrays`` of the interface classes).
"""
from __future__ import annotations

import logging
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

logger = logging.getLogger(__name__)

__all__ = ["StateError", "ObservationError", "ModelState", "ObsSubset", "to_seconds", "validate_state",
           "validate_observations", "get_analysis_time", "slice_analysis", "get_pseudo_state",
           "apply_obs_operator", "obs_space_blocks", "obs_space_variables", "state_information",
           "update_state", "assimilate_arrays"]

STATE_DIMS = ("var_name", "time", "ensemble", "grid")


class StateError(Exception):
    """pytassim.state.StateError (state.py:44)."""


class ObservationError(Exception):
    """pytassim.observation.ObservationError (observation.py:44)."""


def to_seconds(times) -> np.ndarray:
    """Time axis -> float64 seconds since 1970-01-01 (utilities/pandas.py:28-45).  Numbers pass through."""
    t = np.atleast_1d(np.asarray(times))
    if np.issubdtype(t.dtype, np.datetime64):
        return (t.astype("datetime64[ns]") - np.datetime64(0, "ns")).astype(np.float64) * 1e-9
    if t.dtype == object or t.dtype.kind in "US":
        return to_seconds(t.astype("datetime64[ns]"))
    return t.astype(np.float64)


def _coord_table(coords) -> np.ndarray:
    """utilities/pandas.py:70-102 ``index_to_array``: (n,) or (n, levels) -> float (n, levels)."""
    c = np.asarray(coords, dtype=np.float64)
    return c.reshape(len(c), -1) if c.ndim != 2 else c


@dataclass
class ModelState:
    """Array stand-in for the reference's state DataArray (state.py:60-113)."""
    values: Any                       # (var_name, time, ensemble, grid) numpy array or torch tensor
    time: Any                         # (T,)
    grid: Any                         # (G,) or (G, n_coord) grid coordinates
    dims: Tuple[str, ...] = STATE_DIMS
    ensemble: Optional[Sequence] = None
    time_index: Optional[np.ndarray] = None     # positions of ``time`` in the state it was sliced from (for shims)
    source: Any = None                # opaque handle of the object ``time_index`` refers to (the shim's DataArray)

    def __post_init__(self):
        self.time = to_seconds(self.time)
        self.grid = _coord_table(self.grid)
        if self.time_index is None:
            self.time_index = np.arange(len(self.time))

    @property
    def valid(self) -> bool:          # state.py:103-129: names AND order
        shp = tuple(getattr(self.values, "shape", ()))      # (numpy array or torch tensor, possibly on the GPU)
        return tuple(self.dims) == STATE_DIMS and len(shp) == 4 and shp[1] == len(self.time) and shp[3] == len(self.grid)

    def sel_time(self, t: float) -> "ModelState":
        """``.sel(time=[t])``: exact label selection; KeyError when the label is absent."""
        hit = np.nonzero(self.time == t)[0]
        if hit.size == 0:
            raise KeyError("time {0!r} not in state".format(t))
        i = int(hit[0])
        return ModelState(self.values[:, i:i + 1], self.time[i:i + 1], self.grid, self.dims, self.ensemble,
                          self.time_index[i:i + 1], self.source)

    def with_values(self, values) -> "ModelState":
        return ModelState(values, self.time, self.grid, self.dims, self.ensemble, self.time_index, self.source)


def _no_operator(obs_subset, state):      # observation.py:297-299
    raise NotImplementedError("No observation operator is set!")


@dataclass
class ObsSubset:
    """Array stand-in for one observation Dataset (observation.py:60-239): ``observations (time, obs_grid_1)``,
    ``covariance`` (obs_grid_1) | (time, obs_grid_1) [variances] | (obs_grid_1, obs_grid_2) | (time, obs_grid_1,
    obs_grid_2), the observation coordinates and the operator ``operator(obs_subset, pseudo_state) -> (ensemble,
    time, obs_grid_1)``."""
    observations: Any
    covariance: Any
    time: Any
    grid: Any
    operator: Callable = _no_operator
    correlated: Optional[bool] = None           # None: told from the covariance's rank
    cov_has_time: Optional[bool] = None
    grid_names: Optional[Sequence[str]] = None  # column names of the coordinates (obs_info DataFrame)
    time_index: Optional[np.ndarray] = None

    def __post_init__(self):
        self.time = to_seconds(self.time)
        self.grid = _coord_table(self.grid)
        self.observations = np.asarray(self.observations)
        self.covariance = np.asarray(self.covariance)
        T, P = (self.observations.shape + (0, 0))[:2] if self.observations.ndim == 2 else (len(self.time), len(self.grid))
        nd = self.covariance.ndim
        if self.correlated is None or self.cov_has_time is None:
            # rank alone is ambiguous for a (T, P) variance table with T == P; the explicit flags settle it
            if nd == 3:
                corr, has_t = True, True
            elif nd == 1:
                corr, has_t = False, False
            else:
                has_t = self.covariance.shape == (T, P) and T != P
                corr = not has_t
            self.correlated = corr if self.correlated is None else self.correlated
            self.cov_has_time = has_t if self.cov_has_time is None else self.cov_has_time
        if self.time_index is None:
            self.time_index = np.arange(len(self.time))

    @property
    def valid(self) -> bool:          # observation.py:113-239
        obs = self.observations
        if obs.ndim != 2 or obs.shape != (len(self.time), len(self.grid)):
            return False
        T, P = obs.shape
        want = ((T,) if self.cov_has_time else ()) + ((P, P) if self.correlated else (P,))
        return self.covariance.shape == want

    def sel_time(self, t: float) -> "ObsSubset":
        """``obs.sel(time=[t])`` with the operator re-attached (filter.py:50-53)."""
        hit = np.nonzero(self.time == t)[0]
        if hit.size == 0:
            raise KeyError("time {0!r} not in observations".format(t))
        i = int(hit[0])
        cov = self.covariance[i:i + 1] if self.cov_has_time else self.covariance
        return ObsSubset(self.observations[i:i + 1], cov, self.time[i:i + 1], self.grid, self.operator,
                         self.correlated, self.cov_has_time, self.grid_names, self.time_index[i:i + 1])


# ---------------------------------------------------------------------------------------------------------
def validate_state(state) -> None:
    """interface/base.py:129-137."""
    if not isinstance(state, ModelState):
        raise TypeError("*** Given state is not a valid ``ModelState`` ***\n{0}".format(type(state)))
    if not state.valid:
        raise StateError("*** Given state is not a valid state ***\n{0:s}".format(str(state.dims)))


def validate_observations(observations: Iterable) -> None:
    """interface/base.py:139-151."""
    for obs in observations:
        if not isinstance(obs, ObsSubset):
            raise TypeError("*** Given observation is not a valid ``ObsSubset`` ***\n{0}".format(type(obs)))
        if not obs.valid:
            raise ObservationError("*** Given observation is not a valid observation ***\n{0:s}".format(
                str((obs.observations.shape, obs.covariance.shape))))


def get_analysis_time(state: ModelState, analysis_time=None) -> float:
    """interface/base.py:154-179: None -> the state's last time; an absent time -> nearest state time + UserWarning."""
    if analysis_time is None:
        return float(state.time[-1])
    t = float(to_seconds(analysis_time)[-1])
    if np.any(state.time == t):
        return t
    near = float(state.time[int(np.argmin(np.abs(state.time - t)))])
    warnings.warn("Given analysis time {0:s} is not within state, used instead nearest neighbor {1:s}".format(
        str(analysis_time), str(near)), category=UserWarning)
    return near


def slice_analysis(analysis_time: float, state: ModelState, observations: Iterable[ObsSubset],
                   pseudo_state: ModelState):
    """interface/filter.py:39-55: state, pseudo state and EVERY observation subset are cut to [analysis_time]."""
    logger.info("Assimilation in filtering mode")
    return state.sel_time(analysis_time), [o.sel_time(analysis_time) for o in observations], \
        pseudo_state.sel_time(analysis_time)


def get_pseudo_state(algo, pseudo_state: Optional[ModelState], state: ModelState, iter_num: int = 0) -> ModelState:
    """interface/base.py:331-357.  The prior weights are the identity (base.py:243-254), under which
    ``_apply_weights`` returns the state itself, so the forward model is handed the state."""
    if pseudo_state is None and getattr(algo, "forward_model", None) is not None:
        _, pseudo_state = algo.forward_model(state, iter_num)
        validate_state(pseudo_state)
    elif pseudo_state is None:
        pseudo_state = state
    return pseudo_state


def apply_obs_operator(pseudo_state: ModelState, observations: Iterable[ObsSubset]):
    """interface/base.py:181-220: subsets whose operator raises NotImplementedError are dropped silently."""
    ens_obs, used = [], []
    for obs in observations:
        try:
            hx = obs.operator(obs, pseudo_state)
        except NotImplementedError:
            continue
        ens_obs.append(hx)
        used.append(obs)
    logger.info("Applied the observation operators")
    return ens_obs, used


def obs_space_blocks(ens_obs: Sequence, observations: Sequence[ObsSubset]):
    """What ``_get_obs_space_variables`` normalises, cut into independent blocks in the reference's stacking order
    (base.py:223-241: per subset, ``stack(obs_id=('time', 'obs_grid_1'))`` = time-major, subsets concatenated).
    Every block is ``(hx (k, n), y (n,), var (n,) | None, cov (n, n) | None)``: an uncorrelated subset is ONE block
    over all its times, a correlated subset one block per time (R is block diagonal in time, observation.py:254-265).
    Also returns the observation table (P, 1 + n_coord): column 0 = time in seconds, then the coordinates."""
    hxs, ys, vars_, covs, tabs = [], [], [], [], []
    for hx, obs in zip(ens_obs, observations):
        hx = hx if hasattr(hx, "shape") else np.asarray(hx)
        if hx.ndim != 3 or tuple(hx.shape[1:]) != obs.observations.shape:
            raise ValueError("Observational size between ensemble {0} and observations {1} do not match!".format(
                tuple(hx.shape[1:]), obs.observations.shape))
        T, P = obs.observations.shape
        k = hx.shape[0]
        if obs.correlated:
            for t in range(T):
                hxs.append(hx[:, t]); ys.append(obs.observations[t]); vars_.append(None)
                covs.append(obs.covariance[t] if obs.cov_has_time else obs.covariance)
        else:
            var = obs.covariance if obs.cov_has_time else np.broadcast_to(obs.covariance, (T, P))
            hxs.append(hx.reshape(k, T * P)); ys.append(obs.observations.reshape(T * P))
            vars_.append(np.ascontiguousarray(var).reshape(T * P)); covs.append(None)
        tabs.append(np.hstack([np.repeat(obs.time, P)[:, None], np.tile(obs.grid, (T, 1))]))
    table = np.concatenate(tabs, axis=0) if tabs else np.zeros((0, 2))
    return (hxs, ys, vars_, covs), table


def obs_space_variables(algo, ens_obs, observations):
    """base.py:359-379 through the algorithm's device routine: -> (d (P,), Yb (k, P), obs table)."""
    (hxs, ys, vars_, covs), table = obs_space_blocks(ens_obs, observations)
    d, yb = algo.get_obs_space_variables(hxs, ys, variances=vars_, covariances=covs)
    logger.info("Normalized data in observational space")
    return d, yb, table


def state_information(state: ModelState) -> np.ndarray:
    """interface/mixin_local.py:50-69: rows ``[t0_unix_seconds, *grid_coords]`` (the FIRST time of the state)."""
    return np.hstack([np.full((len(state.grid), 1), state.time[0]), state.grid])


def obs_information(table: np.ndarray, observations: Sequence[ObsSubset]):
    """mixin_local.py:44-47: what a user ``dist_func`` receives as ``obs_info`` -- a DataFrame with one column per
    level of ``obs_id`` (``time`` first).  Falls back to the bare table when pandas is absent."""
    try:
        import pandas as pd
    except ImportError:      # pragma: no cover
        return table
    names = None
    for o in observations:
        if o.grid_names is not None:
            names = list(o.grid_names)
            break
    ncoord = table.shape[1] - 1
    if names is None or len(names) != ncoord:
        names = ["obs_grid_1"] if ncoord == 1 else ["obs_grid_1_level_{0}".format(i) for i in range(ncoord)]
    return pd.DataFrame(table, columns=["time"] + names)


def update_state(algo, state: ModelState, observations: Iterable[ObsSubset],
                 pseudo_state: Optional[ModelState], analysis_time: float) -> ModelState:
    """interface/filter.py:96-165 with estimate_weights + _apply_weights fused into ``algo.analyse_arrays``."""
    pseudo_state = get_pseudo_state(algo, pseudo_state, state)
    validate_state(pseudo_state)
    if not algo.smoother:
        state, observations, pseudo_state = slice_analysis(analysis_time, state, observations, pseudo_state)
    ens_obs, used = apply_obs_operator(pseudo_state, observations)
    logger.info("Start to estimate the weights")
    d, yb, table = obs_space_variables(algo, ens_obs, used)
    xa = algo.analyse_arrays(state.values, yb, d, grid_coords=state.grid, obs_coords=table[:, 1:],
                             grid_info=state_information(state), obs_info=obs_information(table, used))
    return state.with_values(xa)


def assimilate_arrays(algo, state: ModelState, observations, pseudo_state: Optional[ModelState] = None,
                      analysis_time=None) -> ModelState:
    """interface/base.py:419-512."""
    if observations is None or (not isinstance(observations, ObsSubset) and len(observations) == 0):
        warnings.warn("No observation is given, I will return the background state!", UserWarning)
        return state
    if not isinstance(observations, (list, set, tuple)):
        observations = (observations,)
    validate_state(state)
    validate_observations(observations)
    analysis_time = get_analysis_time(state, analysis_time)
    for trans in (getattr(algo, "pre_transform", None) or ()):
        state, observations, pseudo_state = trans.pre(state, observations, pseudo_state)
    analysis = update_state(algo, state, observations, pseudo_state, analysis_time)
    for trans in (getattr(algo, "post_transform", None) or ()):
        analysis = trans.post(analysis, state, observations, pseudo_state)
    validate_state(analysis)
    return analysis

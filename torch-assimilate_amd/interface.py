"""Drivers with the constructor / method surface of pytassim.interface.{ETKF, LETKF, KETKF, LKETKF}
(pytassim/interface/etkf.py:66-120, letkf.py:72-148, ketkf.py:69-123, lketkf.py:77-115), backed
by the gfx950 engine.

Two levels:

* array level (always available): ``analyse_arrays`` / ``estimate_weights_arrays`` take the plain
  arrays the reference hands to its per-grid-point closure -- normalised obs-space perturbations
  ``Yb (k, P)``, innovations ``d (P,)``, grid / observation coordinates, the prior ensemble
  ``(..., k, G)`` -- and return the analysis (or the weights ``(G, k, k)`` that
  ``estimate_weights`` returns, letkf.py:145-146);
* xarray level (when xarray is importable): ``assimilate(state, observations, pseudo_state,
  analysis_time)`` with the reference's data model (state dims ``var_name, time, ensemble, grid``;
  observation Datasets with ``observations`` / ``covariance``), reproducing
  ``_get_obs_space_variables`` (base.py:359-379), the stacking order of ``_stack_obs``
  (base.py:223-241), ``state_info`` (mixin_local.py:50-69) and ``_apply_weights`` (base.py:257-278).
  xarray is not installed in the build / GPU images, so that layer is exercised only where it is.

There is no CPU path: ``gpu`` is accepted for signature compatibility and ignored.
"""
from __future__ import annotations

import logging
import warnings
from typing import Callable, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from .engine import LetkfEngine
from .kernels import kernel_route
from .localization import GaspariCohn

logger = logging.getLogger(__name__)

__all__ = ["ETKF", "LETKF", "KETKF", "LKETKF"]


class _NoLocalization:
    """``localization=None`` in the reference: every observation, weight 1 (wrapper.py:87)."""


class ETKF:
    """Global ensemble transform Kalman filter (interface/etkf.py:33-120)."""

    def __init__(self, inf_factor: float = 1.0, smoother: bool = False, gpu: bool = True,
                 pre_transform: Optional[Iterable] = None, post_transform: Optional[Iterable] = None,
                 weight_save_path: Optional[str] = None, forward_model: Optional[Callable] = None,
                 dtype: torch.dtype = torch.float64, engine: Optional[LetkfEngine] = None):
        self._inf_factor = float(inf_factor)
        self.smoother = smoother
        self.gpu = gpu
        self.pre_transform = pre_transform
        self.post_transform = post_transform
        self.weight_save_path = weight_save_path
        self.forward_model = forward_model
        # the reference's working precision: float64 unless the caller says otherwise (interface/base.py:68,73).  The float32
        # tile kernels -- the benchmarked hot path -- are chosen with an explicit dtype=torch.float32
        self._dtype = torch.float64
        self.dtype = dtype
        self._engine = engine
        self._kernel = None

    # ---- properties mirroring the reference ------------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return self._dtype

    @dtype.setter
    def dtype(self, new_type):
        """interface/base.py:106-118: anything but a torch.dtype is a TypeError."""
        if not isinstance(new_type, torch.dtype):
            raise TypeError("Given object is not a valid torch.dtype, instead it has as type: {0}".format(type(new_type)))
        if new_type not in (torch.float32, torch.float64):
            raise TypeError("the gfx950 engine computes in torch.float32 or torch.float64, not {0}".format(new_type))
        self._dtype = new_type

    @property
    def inf_factor(self) -> float:
        return self._inf_factor

    @inf_factor.setter
    def inf_factor(self, new_factor):
        self._inf_factor = float(new_factor)

    @property
    def engine(self) -> LetkfEngine:
        if self._engine is None:
            self._engine = LetkfEngine()
        return self._engine

    def _kernel_args(self) -> dict:
        # (a per-observation lengthscale vector is meaningful for the global solve only: kernels.GaussKernel)
        gamma, prog = kernel_route(self._kernel, allow_feature_scale=not hasattr(self, "localization"))
        return dict(rbf_gamma=gamma, kernel_program=prog)

    def __str__(self):
        return "Global ETKF(inf_factor={0})".format(self.inf_factor)

    def __repr__(self):
        return "ETKF({0})".format(self.inf_factor)

    # ---- array level ----------------------------------------------------------------------
    def _dev(self, a, dtype=None):
        return torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(
            device=self.engine.device, dtype=dtype or self.dtype)

    def estimate_weights_arrays(self, yb, d, **_unused) -> torch.Tensor:
        """(k, P), (P,) -> weights (k, k) (ETKF.estimate_weights, etkf.py:99-120)."""
        if any(v is not None for v in self._kernel_args().values()):     # kernelised global solve: one "grid point" seeing every observation
            from .core import KETKFModule
            return KETKFModule(self._kernel, self.inf_factor, self.engine)(self._dev(yb), self._dev(d))
        return self.engine.etkf_weights(self._dev(yb), self._dev(d), self.inf_factor)

    def get_obs_space_variables(self, ens_obs, observations, variances=None, covariances=None):
        """Array-level ``_get_obs_space_variables`` (interface/base.py:359-379) on the device: for every
        observation subset j, ``ens_obs[j]`` (k, P_j) = H_j(x) of the ensemble, ``observations[j]`` (P_j,) and
        its R as ``variances[j]`` (P_j,) (uncorrelated, observation.py:241-245) or ``covariances[j]``
        (P_j, P_j) (correlated, observation.py:247-275; None entries fall back to the variance).
        Returns the stacked (innovations (P,), ens_obs_perts (k, P)) -- the order of the reference's return."""
        n = len(ens_obs)
        if len(observations) != n:
            raise ValueError("one observation vector per ensemble-in-observation-space array")
        sizes = [int(torch.as_tensor(h).shape[-1]) for h in ens_obs]
        k = int(torch.as_tensor(ens_obs[0]).shape[0])
        total = sum(sizes)
        Yb = torch.empty((k, total), dtype=self.dtype, device=self.engine.device)
        d = torch.empty(total, dtype=self.dtype, device=self.engine.device)
        off = 0
        for j in range(n):
            cov = covariances[j] if covariances is not None else None
            var = variances[j] if (variances is not None and cov is None) else None
            self.engine.obs_space(ens_obs[j], observations[j], var=var, cov=cov, dtype=self.dtype, out=(Yb, d), offset=off)
            off += sizes[j]
        return d, Yb

    def store_weights(self, weights: torch.Tensor, grid_index=None, ensemble=None, grid_levels=None) -> None:
        """BaseAssimilation.store_weights (base.py:280-300): the weights go to ``weight_save_path`` as netCDF."""
        from . import weights_io
        weights_io.store_weights(self.weight_save_path, weights, grid_index=grid_index, ensemble=ensemble,
                                 grid_levels=grid_levels)

    def load_weights(self) -> torch.Tensor:
        """BaseAssimilation.load_weights (base.py:302-325): back onto the GPU, in the working dtype."""
        from . import weights_io
        return weights_io.load_weights(self.weight_save_path, self.engine.device, self.dtype)[0]

    def _through_disk(self, weights: torch.Tensor, grid_index=None) -> torch.Tensor:
        """FilterAssimilation.update_state (filter.py:160-163): with a weight_save_path the weights are stored and
        the analysis uses what was loaded back."""
        if self.weight_save_path is None:
            return weights
        self.store_weights(weights, grid_index=grid_index)
        return self.load_weights()

    def analyse_arrays(self, state, yb, d, **_unused) -> torch.Tensor:
        """state (..., k, G) -> analysis of the same shape: weights + _apply_weights (base.py:257-278)."""
        st = self._dev(state)
        shp = st.shape
        W = self._through_disk(self.estimate_weights_arrays(yb, d))
        xa = self.engine.apply_weights(st.reshape(-1, shp[-2], shp[-1]), W)
        return xa.reshape(shp)

    # ---- assimilate(): the reference's entry point (interface/base.py:419-512) ------------------------
    def assimilate(self, state, observations, pseudo_state=None, analysis_time=None):
        """xarray in / xarray out as in the reference; with the array-level data model of ``assim_flow``
        (``ModelState`` / ``ObsSubset``) in, the same flow runs without xarray and returns a ``ModelState``."""
        from . import assim_flow
        if isinstance(state, assim_flow.ModelState):
            return assim_flow.assimilate_arrays(self, state, observations, pseudo_state, analysis_time)
        from . import xr_adapter
        return xr_adapter.assimilate(self, state, observations, pseudo_state, analysis_time)


class LETKF(ETKF):
    """Localised ETKF (interface/letkf.py:34-148): independent local analysis per grid point."""

    def __init__(self, localization: Optional[GaspariCohn] = None, inf_factor: float = 1.0,
                 smoother: bool = False, gpu: bool = True, pre_transform=None, post_transform=None,
                 chunksize: int = 10, weight_save_path=None, forward_model=None,
                 dtype: torch.dtype = torch.float64, engine: Optional[LetkfEngine] = None):
        super().__init__(inf_factor=inf_factor, smoother=smoother, gpu=gpu, pre_transform=pre_transform,
                         post_transform=post_transform, weight_save_path=weight_save_path,
                         forward_model=forward_model, dtype=dtype, engine=engine)
        self.localization = localization
        self.chunksize = chunksize          # dask chunking of the reference; the GPU path does not chunk

    @property
    def chunks(self):
        return dict(grid=self.chunksize)

    def __str__(self):
        return "Localized ETKF(inf_factor={0}, loc={1})".format(self.inf_factor, str(self.localization))

    def __repr__(self):
        return "LETKF({0},{1})".format(repr(self.inf_factor), repr(self.localization))

    def _lists(self, grid_coords, obs_coords, g0=0, g1=None, grid_info=None, obs_info=None):
        eng = self.engine
        G = len(grid_coords)
        g1 = G if g1 is None else g1
        if self.localization is None:
            P = len(obs_coords)
            cap = max(P, 1)
            cand = torch.arange(cap, dtype=torch.int32, device=eng.device)[None].expand(g1 - g0, -1).contiguous()
            if P == 0:
                cand = cand - 1
            dist = torch.zeros((1, g1 - g0, cap), dtype=torch.float64, device=eng.device)
            return eng.localize_from_dist(dist, cand, [1.0], g0=g0)
        return self.localization.neighbour_lists(eng, grid_coords, obs_coords, g0, g1, grid_info, obs_info)

    def estimate_weights_arrays(self, yb, d, grid_coords=None, obs_coords=None, g0=0, g1=None,
                                grid_info=None, obs_info=None) -> torch.Tensor:
        """weights (G, k, k) with [g, i, j] = w_mean_i + W_ij (letkf.py:127-146)."""
        if self.localization is None:
            # no localisation: every grid point sees every observation with weight 1 (wrapper.py:87), i.e. the one
            # global solve repeated G times in the reference -- done once here, for any number of observations
            if grid_coords is None and g1 is None:
                raise ValueError("estimate_weights_arrays without localisation needs grid_coords or g1 (the number of "
                                 "grid points the (G, k, k) weights are repeated for)")
            G = len(grid_coords) if g1 is None else g1
            W = ETKF.estimate_weights_arrays(self, yb, d)
            return W[None].expand(G - g0, -1, -1).contiguous()
        yb, d = self._dev(yb), self._dev(d)
        k = yb.shape[0]
        # the list bound of the last call on this object, when the problem has the same sizes: tile lists are built for it at once
        # and checked afterwards (longest list, unions that fit); the per-point lists -- a third of this call's time at config 2
        # -- are only built when that fails or a declined point needs them
        G = len(grid_coords)
        hint_key = (G, int(yb.shape[1]), g0, G if g1 is None else g1)
        hint = getattr(self, "_w_hint", None)
        if hint is not None and hint[0] == hint_key and self.localization is not None:
            x = torch.zeros((1, k, hint_key[3]), dtype=self.dtype, device=self.engine.device)
            W = self._weights_on_tiles(x, yb, d, None, grid_coords, obs_coords, p_max=hint[1], g0=g0, g1=hint_key[3],
                                       lists=lambda: self._lists(grid_coords, obs_coords, g0, g1, grid_info, obs_info))
            if W is not None:
                return W
        nb = self._lists(grid_coords, obs_coords, g0, g1, grid_info, obs_info)
        self._w_hint = (hint_key, int(nb.p_max))
        x = torch.zeros((1, k, nb.g1), dtype=self.dtype, device=self.engine.device)
        W = self._weights_on_tiles(x, yb, d, nb, grid_coords, obs_coords)
        if W is not None:
            return W
        _, W = self.engine.analysis(x, yb, d, nb, self.inf_factor, return_weights=True, **self._kernel_args())
        return W

    def _weights_on_tiles(self, x, yb, d, nb, grid_coords, obs_coords, p_max=None, g0=None, g1=None, lists=None):
        """The tile route of the weights (engine.weights_tiles: tile lists + split records, csrc/letkf_tile2w.hip) where it
        applies -- float32, plain ETKF core, a built-in distance, unions of at most 32 observations per tile; declined points are
        redone with weights by the eigensolver kernel from the per-point lists.  None: the caller takes the per-point route.
        ``nb`` None: the lists are not built yet -- ``p_max`` is a bound carried over from an earlier call (checked against the
        longest list the tile kernel met: None when it does not hold), ``lists()`` builds them if a declined point needs them."""
        eng = self.engine
        ka = self._kernel_args()
        if nb is not None:
            p_max, g0, g1 = nb.p_max, nb.g0, nb.g1
        if (ka.get("rbf_gamma") is not None or ka.get("kernel_program") is not None
                or not hasattr(self.localization, "tile_lists") or g1 - g0 <= 0):
            return None
        tiles = None
        for extra in (0, 1):
            if not eng.tile_route_applies(x, p_max, extra) or max(1, (p_max + 8 + 15) // 16) + extra > 2:
                return None
            tiles = self.localization.tile_lists(eng, grid_coords, obs_coords, p_max, g0, g1, extra_blocks=extra)
            if tiles is None:
                return None
            st_ = tiles.stats.tolist()
            if nb is None and st_[0] > p_max:
                return None                      # (the carried bound does not hold for this network: exact lists first)
            if st_[1] == 0:
                break
        else:
            return None
        rec = eng.pack_split(yb, d)
        res = eng.weights_tiles(x, rec, yb.shape[1], tiles, self.inf_factor)
        if res is None:
            return None
        xa, W, flags, retry = res
        if int(retry.item()):
            if nb is None:
                nb = lists()
            eng.weights_retry(x, yb, d, nb, self.inf_factor, xa, W, flags)
        return W

    def _analysis_by_step_driver(self, x, yb, d, grid_coords, obs_coords, g0, g1):
        """The whole analysis as ONE call of the native step driver (sharded.ShardedLetkf on one rank: observation index, the
        tiles' localisation inside the analysis wavefronts, float64 redo of declined points, the list bound carried from call
        to call) where it applies: float32, Gaspari-Cohn on a built-in distance, the plain ETKF core or the RBF / Gauss kernel,
        the whole grid.  The class route below (per-point lists -> tile lists -> records -> analysis, each an engine call with
        its own host round trips) takes 0.32 ms per call at config 2 where this takes 0.09, with the same bits.  None: not this
        shape -- the caller goes on."""
        ka = self._kernel_args()
        loc = self.localization
        G = x.shape[-1]
        if (ka.get("kernel_program") is not None or loc is None or getattr(loc, "_taper", 1) != 0
                or getattr(loc, "builtin_metric", None) is None or x.dtype != torch.float32 or yb.shape[1] == 0
                or g0 != 0 or (g1 is not None and g1 != G) or G == 0):
            return None
        from .sharded import ShardedLetkf
        grid = torch.as_tensor(np.asarray(grid_coords) if not torch.is_tensor(grid_coords) else grid_coords)
        obs = torch.as_tensor(np.asarray(obs_coords) if not torch.is_tensor(obs_coords) else obs_coords)
        nc = 1 if grid.dim() == 1 else int(grid.shape[1])
        if grid.shape[0] != G or obs.shape[0] != yb.shape[1]:
            return None
        key = (tuple(float(r) for r in loc.radius), tuple(loc.builtin_metric.groups(nc, len(loc.radius))), float(loc.epsilon),
               float(self.inf_factor), ka.get("rbf_gamma"), nc)
        runner = getattr(self, "_step_runner", None)
        if runner is None or getattr(self, "_step_runner_key", None) != key:
            if runner is not None:
                runner.close()
            runner = ShardedLetkf(self.engine.device, 0, 1, radii=list(key[0]), coord_group=list(key[1]), eps=key[2],
                                  inf_factor=key[3], rbf_gamma=key[4])
            runner._engine = self.engine
            self._step_runner, self._step_runner_key = runner, key
        dev = self.engine.device
        xa = runner.assimilate(x, grid.to(device=dev, dtype=torch.float64), obs.to(device=dev, dtype=torch.float64), yb, d)
        bad = runner.last_flags_summary          # (known from the step's status word on the default route: no scan of 1e5 flags, no sync)
        if bad is None:
            fl = runner._last_flags
            bad = int((fl & 0xff).max().item()) if fl is not None and fl.numel() else 0
        if bad & 1:
            raise RuntimeError("LETKF kernel: local observation list overflow (engine bug: lists are sized from counts)")
        if bad & 4:
            warnings.warn("LETKF kernel met non-finite values in at least one local block", RuntimeWarning)
        if bad & 2:
            warnings.warn("LETKF eigensolver reached its sweep cap for at least one grid point (result returned)",
                          RuntimeWarning)
        return xa

    def _analysis_on_tiles(self, x, yb, d, nb, grid_coords, obs_coords):
        """The fused analysis on the TILE route -- tile lists from the built-in metric, then letkf_tile2_kernel / letkf_tile2p_kernel
        (plain ETKF core: split records) or lketkf_tile_kernel (RBF / Gauss kernel: the perturbations themselves) -- where it
        applies: float32, a built-in distance, a shape the kernels take (mia_letkf_tiles_cover).  Declined points are redone by
        the eigensolver kernel from the per-point lists.  Returns (Xa, flags) or None: the caller takes the per-point route."""
        eng = self.engine
        ka = self._kernel_args()
        gamma = ka.get("rbf_gamma")
        if (ka.get("kernel_program") is not None or self.localization is None or not hasattr(self.localization, "tile_lists")
                or nb.g1 - nb.g0 <= 0 or x.dtype != torch.float32 or yb.shape[1] == 0):
            return None
        P = int(yb.shape[1])
        tiles = None
        for extra in range(0, 6):
            if not eng.tile_route_applies(x, nb.p_max, extra, rbf_gamma=gamma, P=P, n_points=nb.g1 - nb.g0):
                return None
            tiles = self.localization.tile_lists(eng, grid_coords, obs_coords, nb.p_max, nb.g0, nb.g1, extra_blocks=extra)
            if tiles is None:
                return None
            n_over = int(tiles.stats[1].item())
            if n_over == 0:
                break
            if n_over & (1 << 30):          # MIA_TILE_BOX_OVERFLOW: more slots cannot help
                return None
        else:
            return None
        if gamma is not None:
            res = eng.analysis_tiles_rbf(x, yb, d, tiles, self.inf_factor, gamma)
        else:
            res = eng.analysis_tiles(x, eng.pack_split(yb, d), P, tiles, self.inf_factor)
        if res is None:
            return None
        xa, flags, retry = res
        if int(retry.item()):
            eng.retry_points(x, yb, d, nb, self.inf_factor, xa, flags, rbf_gamma=gamma)
        return xa, flags

    def analyse_arrays(self, state, yb, d, grid_coords=None, obs_coords=None, g0=0, g1=None,
                       grid_info=None, obs_info=None) -> torch.Tensor:
        """Fused path: the weights never leave the GPU's LDS -- unless a ``weight_save_path`` asks for them: then
        estimate_weights -> store -> load -> _apply_weights as in the reference (filter.py:157-164)."""
        st = self._dev(state)
        shp = st.shape
        if self.localization is None and self.weight_save_path is None:
            W = ETKF.estimate_weights_arrays(self, yb, d)          # one global solve (see estimate_weights_arrays)
            g1_ = shp[-1] if g1 is None else g1
            xa = self.engine.apply_weights(st.reshape(-1, shp[-2], shp[-1]), W, g0, g1_)
            return xa.reshape(shp[:-1] + (g1_ - g0,))
        if self.weight_save_path is not None:
            W = self.estimate_weights_arrays(yb, d, grid_coords, obs_coords, g0, g1, grid_info, obs_info)
            gidx = np.arange(g0, g0 + W.shape[0])
            W = self._through_disk(W, grid_index=gidx)
            xa = self.engine.apply_local_weights(st.reshape(-1, shp[-2], shp[-1]), W, g0, g0 + W.shape[0])
            return xa.reshape(shp[:-1] + (W.shape[0],))
        st3, ybd, dd = st.reshape(-1, shp[-2], shp[-1]), self._dev(yb), self._dev(d)
        xa = self._analysis_by_step_driver(st3, ybd, dd, grid_coords, obs_coords, g0, g1)
        if xa is not None:
            return xa.reshape(shp[:-1] + (xa.shape[-1],))
        nb = self._lists(grid_coords, obs_coords, g0, g1, grid_info, obs_info)
        res = self._analysis_on_tiles(st3, ybd, dd, nb, grid_coords, obs_coords)
        if res is not None:
            xa, flags = res
        else:
            xa, flags = self.engine.analysis(st3, ybd, dd, nb, self.inf_factor, return_flags=True, **self._kernel_args())
        bad = int((flags & 0xff).max().item()) if flags.numel() else 0
        if bad & 1:
            raise RuntimeError("LETKF kernel: local observation list overflow (engine bug: lists are sized from counts)")
        if bad & 4:
            warnings.warn("LETKF kernel met non-finite values in at least one local block", RuntimeWarning)
        if bad & 2:
            warnings.warn("LETKF eigensolver reached its sweep cap for at least one grid point (result returned)",
                          RuntimeWarning)
        return xa.reshape(shp[:-1] + (nb.g1 - nb.g0,))


class KETKF(ETKF):
    """Kernelised ETKF (interface/ketkf.py:34-123)."""

    def __init__(self, kernel, inf_factor: float = 1.0, smoother: bool = False, gpu: bool = True,
                 pre_transform=None, post_transform=None, weight_save_path=None, forward_model=None,
                 dtype: torch.dtype = torch.float64, engine: Optional[LetkfEngine] = None):
        super().__init__(inf_factor=inf_factor, smoother=smoother, gpu=gpu, pre_transform=pre_transform,
                         post_transform=post_transform, weight_save_path=weight_save_path,
                         forward_model=forward_model, dtype=dtype, engine=engine)
        self.kernel = kernel

    @property
    def kernel(self):
        return self._kernel

    @kernel.setter
    def kernel(self, new_kernel):
        self._kernel = new_kernel

    def __str__(self):
        return "Global KETKF(inf_factor={0}, kernel={1})".format(self.inf_factor, str(self.kernel))

    def __repr__(self):
        return "KETKF({0})".format(repr(self.kernel))


class LKETKF(LETKF):
    """Localised kernelised ETKF (interface/lketkf.py:37-115; estimate_weights is LETKF's, :77)."""

    def __init__(self, kernel, localization: Optional[GaspariCohn] = None, inf_factor: float = 1.0,
                 smoother: bool = False, gpu: bool = True, pre_transform=None, post_transform=None,
                 chunksize: int = 10, weight_save_path=None, forward_model=None,
                 dtype: torch.dtype = torch.float64, engine: Optional[LetkfEngine] = None):
        super().__init__(localization=localization, inf_factor=inf_factor, smoother=smoother, gpu=gpu,
                         pre_transform=pre_transform, post_transform=post_transform, chunksize=chunksize,
                         weight_save_path=weight_save_path, forward_model=forward_model, dtype=dtype, engine=engine)
        self._kernel = kernel

    kernel = KETKF.kernel

    def __str__(self):
        return "Localized KETKF(inf_factor={0}, loc={1}, kernel={2})".format(
            self.inf_factor, str(self.localization), str(self.kernel))

    def __repr__(self):
        return "LKETKF({0},{1})".format(repr(self.localization), repr(self.kernel))

"""Iterative ensemble Kalman smoother on the gfx950 engine: mirror of pytassim.core.ienks and of
pytassim.interface.{ienks, lienks} for the per-grid-point weight update.

Core modules (B3 seam): ``IEnKSTransformModule(tau)(weights, normed_perts, normed_obs) -> weights`` and
``IEnKSBundleModule(epsilon, tau)`` with the call convention and error behaviour of core/ienks.py:128-141
(ValueError on mismatching observation sizes, weights unchanged for an empty observation block).

Drivers: ``IEnKSTransform / IEnKSBundle`` (global, interface/ienks.py:33-164) and ``LocalizedIEnKSTransform /
LocalizedIEnKSBundle`` (interface/lienks.py:34-163) at the array level: ``inner_loop_arrays`` is the reference's
``inner_loop`` (one weight update per grid point), ``update_state_arrays`` is ``VarAssimilation.update_state``
(interface/variational.py:107-135): ``max_iter`` iterations of {model weights -> ensemble transform -> user
forward model -> user observation operator -> obs-space normalisation -> weight update}, then the final
transform.  Forward model and observation operator are user code and run wherever the user runs them; the
weights, the transforms, the normalisation and the update stay on the GPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .core import ETKFModule
from .engine import LetkfEngine
from .interface import LETKF

__all__ = ["IEnKSTransformModule", "IEnKSBundleModule", "IEnKSTransform", "IEnKSBundle",
           "LocalizedIEnKSTransform", "LocalizedIEnKSBundle"]


class IEnKSTransformModule(ETKFModule):
    """core/ienks.py:29-141."""
    epsilon: Optional[float] = None

    def __init__(self, tau: float = 1.0, engine: Optional[LetkfEngine] = None):
        super().__init__(1.0, engine)
        self.tau = float(tau)

    def __str__(self) -> str:
        return "TransformModule(tau={0})".format(self.tau)

    def __repr__(self) -> str:
        return "TransformModule"

    def __call__(self, weights, normed_perts, normed_obs) -> torch.Tensor:
        eng = self.engine
        perts, obs, w = torch.as_tensor(normed_perts), torch.as_tensor(normed_obs), torch.as_tensor(weights)
        self._test_sizes(perts, obs)
        dtype = w.dtype if w.dtype in (torch.float32, torch.float64) else torch.float64
        k = w.shape[-1]
        w = w.reshape(k, k).to(device=eng.device, dtype=dtype)
        p = perts.shape[-1]
        perts = perts.reshape(k, p).to(device=eng.device, dtype=dtype)
        obs = obs.reshape(-1).to(device=eng.device, dtype=dtype)
        cap = max(p, 1)
        cand = torch.arange(cap, dtype=torch.int32, device=eng.device)[None]
        if p == 0:
            cand = cand - 1
        nbrs = eng.localize_from_dist(torch.zeros((1, 1, cap), dtype=torch.float64, device=eng.device), cand, [1.0])
        return eng.ienks_update(w, perts, obs, nbrs, self.tau, self.epsilon)[0]

    forward = __call__


class IEnKSBundleModule(IEnKSTransformModule):
    """core/ienks.py:144-173: dh/dw = normed_perts / epsilon."""

    def __init__(self, epsilon: float = 1e-4, tau: float = 1.0, engine: Optional[LetkfEngine] = None):
        super().__init__(tau, engine)
        self.epsilon = float(epsilon)

    def __str__(self) -> str:
        return "IEnKSBundleModule(eps={0}, tau={1}".format(self.epsilon, self.tau)

    def __repr__(self) -> str:
        return "IEnKSBundle({0}, {1})".format(self.epsilon, self.tau)


class LocalizedIEnKSTransform(LETKF):
    """interface/lienks.py:34-118 (localization=None: interface/ienks.py:33-96, one global block)."""
    _epsilon: Optional[float] = None

    def __init__(self, forward_model: Callable, localization=None, tau: float = 1.0, max_iter: int = 10,
                 smoother: bool = False, gpu: bool = True, pre_transform=None, post_transform=None,
                 chunksize: int = 10, weight_save_path=None, dtype: torch.dtype = torch.float32,
                 engine: Optional[LetkfEngine] = None):
        super().__init__(localization=localization, inf_factor=1.0, smoother=smoother, gpu=gpu,
                         pre_transform=pre_transform, post_transform=post_transform, chunksize=chunksize,
                         weight_save_path=weight_save_path, forward_model=forward_model, dtype=dtype, engine=engine)
        self.max_iter = int(max_iter)
        self.tau = tau

    @property
    def tau(self) -> float:
        return self._tau

    @tau.setter
    def tau(self, new_tau):
        new_tau = float(new_tau)
        if not 0.0 <= new_tau <= 1.0:          # bound_tensor(min_val=0.0, max_val=1.0), interface/ienks.py:82-86
            raise ValueError("tau must lie in [0, 1]")
        self._tau = new_tau

    def __str__(self):
        return "Localized IEnKSTransform(loc={0}, tau={1})".format(str(self.localization), self.tau)

    def __repr__(self):
        return "LIEnKSTransform({0},{1})".format(repr(self.localization), self.tau)

    # ---- pieces of VarAssimilation -----------------------------------------------------------------------
    def generate_prior_weights(self, k: int) -> torch.Tensor:
        """base.py:244-254: the identity."""
        return torch.eye(k, dtype=self.dtype, device=self.engine.device)

    def get_model_weights(self, weights: torch.Tensor) -> torch.Tensor:
        """base.py:327-328: the transform variant propagates the weights themselves."""
        return weights

    def apply_weights_arrays(self, state, weights: torch.Tensor) -> torch.Tensor:
        """_apply_weights (base.py:257-278) for weights (k, k) or (G, k, k); state (..., k, G)."""
        st = self._dev(state)
        shp = st.shape
        x = st.reshape(-1, shp[-2], shp[-1])
        xa = self.engine.apply_weights(x, weights) if weights.dim() == 2 else self.engine.apply_local_weights(x, weights)
        return xa.reshape(shp)

    def inner_loop_arrays(self, weights, yb, d, grid_coords=None, obs_coords=None, grid_info=None,
                          obs_info=None) -> torch.Tensor:
        """``inner_loop`` (lienks.py:75-118 / ienks.py:72-96): weights (k, k) or (G, k, k), normalised obs-space
        perturbations (k, P) and innovations (P,) -> weights (G, k, k) (or (k, k) without a grid)."""
        if grid_coords is None:                     # global IEnKS: one block that sees every observation
            mod = IEnKSBundleModule(self._epsilon, self.tau, self.engine) if self._epsilon is not None \
                else IEnKSTransformModule(self.tau, self.engine)
            return mod(self._dev(weights), self._dev(yb), self._dev(d))
        nb = self._lists(grid_coords, obs_coords, 0, None, grid_info, obs_info)
        return self.engine.ienks_update(self._dev(weights), self._dev(yb), self._dev(d), nb, self.tau, self._epsilon)

    def update_state_arrays(self, state, observe: Callable, grid_coords=None, obs_coords=None,
                            pseudo_state=None) -> torch.Tensor:
        """``VarAssimilation.update_state`` (variational.py:107-135) on arrays.

        state (..., k, G) at the analysis time; ``self.forward_model(model_state, iter_num) -> (_, pseudo_state)``
        as in the reference (base.py:338); ``observe(pseudo_state) -> (ens_obs, observations, variances,
        covariances)`` are the per-subset lists ``get_obs_space_variables`` takes (the user's observation operator
        + observation data; base.py:359-379)."""
        st = self._dev(state)
        k = st.shape[-2]
        weights = self.generate_prior_weights(k)
        for iter_num in range(self.max_iter):
            if pseudo_state is None:                                      # get_pseudo_state -> propagate_model
                model_state = self.apply_weights_arrays(st, self.get_model_weights(weights))
                _, pseudo_state = self.forward_model(model_state, iter_num)
            ens_obs, obs, var, cov = observe(pseudo_state)
            d, yb = self.get_obs_space_variables(ens_obs, obs, var, cov)
            weights = self.inner_loop_arrays(weights, yb, d, grid_coords, obs_coords)
            pseudo_state = None
        analysis = self.apply_weights_arrays(st, weights)
        if self.smoother:
            analysis, _ = self.forward_model(analysis, self.max_iter)
        return analysis


class LocalizedIEnKSBundle(LocalizedIEnKSTransform):
    """interface/lienks.py:121-163 (localization=None: interface/ienks.py:99-164)."""

    def __init__(self, forward_model: Callable, localization=None, tau: float = 1.0, epsilon: float = 1e-4,
                 max_iter: int = 10, smoother: bool = False, gpu: bool = True, pre_transform=None,
                 post_transform=None, chunksize: int = 10, weight_save_path=None,
                 dtype: torch.dtype = torch.float32, engine: Optional[LetkfEngine] = None):
        super().__init__(forward_model, localization=localization, tau=tau, max_iter=max_iter, smoother=smoother,
                         gpu=gpu, pre_transform=pre_transform, post_transform=post_transform, chunksize=chunksize,
                         weight_save_path=weight_save_path, dtype=dtype, engine=engine)
        self.epsilon = epsilon

    @property
    def epsilon(self) -> float:
        return self._epsilon

    @epsilon.setter
    def epsilon(self, new_epsilon):
        new_epsilon = float(new_epsilon)
        if not new_epsilon >= 0.0:              # bound_tensor(min_val=0.0), interface/ienks.py:139-141
            raise ValueError("epsilon must not be negative")
        self._epsilon = new_epsilon

    def __str__(self):
        return "Localized IEnKSBundle(loc={0}, eps={1}, tau={2})".format(str(self.localization), self.epsilon, self.tau)

    def __repr__(self):
        return "LIEnKSBundle({0},{1},{2})".format(repr(self.localization), self.epsilon, self.tau)

    def get_model_weights(self, weights: torch.Tensor) -> torch.Tensor:
        """interface/ienks.py:157-164: epsilon * I + mean over ensemble_new of the weights."""
        eye = self.generate_prior_weights(weights.shape[-1])
        return self.epsilon * eye + weights.mean(dim=-1, keepdim=True)


class IEnKSTransform(LocalizedIEnKSTransform):
    """interface/ienks.py:33-96: the unlocalised driver."""

    def __init__(self, forward_model: Callable, tau: float = 1.0, max_iter: int = 10, smoother: bool = False,
                 gpu: bool = True, pre_transform=None, post_transform=None, weight_save_path=None,
                 dtype: torch.dtype = torch.float32, engine: Optional[LetkfEngine] = None):
        super().__init__(forward_model, localization=None, tau=tau, max_iter=max_iter, smoother=smoother, gpu=gpu,
                         pre_transform=pre_transform, post_transform=post_transform,
                         weight_save_path=weight_save_path, dtype=dtype, engine=engine)

    def __str__(self):
        return "IEnKSTransform(tau={0})".format(self.tau)

    def __repr__(self):
        return "IEnKSTransform({0})".format(self.tau)


class IEnKSBundle(LocalizedIEnKSBundle):
    """interface/ienks.py:99-164."""

    def __init__(self, forward_model: Callable, tau: float = 1.0, epsilon: float = 1e-4, max_iter: int = 10,
                 smoother: bool = False, gpu: bool = True, pre_transform=None, post_transform=None,
                 weight_save_path=None, dtype: torch.dtype = torch.float32, engine: Optional[LetkfEngine] = None):
        super().__init__(forward_model, localization=None, tau=tau, epsilon=epsilon, max_iter=max_iter,
                         smoother=smoother, gpu=gpu, pre_transform=pre_transform, post_transform=post_transform,
                         weight_save_path=weight_save_path, dtype=dtype, engine=engine)

    def __str__(self):
        return "IEnKSBundle(epsilon={0}, tau={1})".format(self.epsilon, self.tau)

    def __repr__(self):
        return "IEnKSBundle({0},{1})".format(self.epsilon, self.tau)

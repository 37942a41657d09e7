"""Per-block core modules: the B3 seam of SURVEY.md section 8(b).

``ETKFModule(inf_factor)(normed_perts, normed_obs) -> weights (k, k)`` with the call convention
and error behaviour of pytassim/core/etkf.py:79-103 (ValueError on mismatching last dimensions,
base.py:28-39; inflated prior for an empty observation block, etkf.py:91-95), evaluated by the
gfx950 kernels: a one-point shard whose neighbour list is every observation with weight 1.
Stateless and re-entrant.  Inference only (no autograd), tensors live on the GPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import LetkfEngine
from .kernels import kernel_route

__all__ = ["ETKFModule", "KETKFModule"]

_engine: Optional[LetkfEngine] = None


def _default_engine() -> LetkfEngine:
    global _engine
    if _engine is None:
        _engine = LetkfEngine()
    return _engine


class ETKFModule:
    def __init__(self, inf_factor: float = 1.0, engine: Optional[LetkfEngine] = None):
        self.inf_factor = float(inf_factor)
        self._engine = engine

    def __str__(self) -> str:
        return "ETKFCore({0})".format(self.inf_factor)

    def __repr__(self) -> str:
        return "ETKFCore"

    kernel = None

    @property
    def engine(self) -> LetkfEngine:
        return self._engine or _default_engine()

    @staticmethod
    def _test_sizes(normed_perts, normed_obs):
        if normed_perts.shape[-1] != normed_obs.shape[-1]:
            raise ValueError(
                "Observational size between ensemble ({0:d}) and observations "
                "({1:d}) do not match!".format(normed_perts.shape[-1], normed_obs.shape[-1]))

    def __call__(self, normed_perts, normed_obs) -> torch.Tensor:
        eng = self.engine
        perts = torch.as_tensor(normed_perts)
        obs = torch.as_tensor(normed_obs)
        self._test_sizes(perts, obs)
        dtype = perts.dtype if perts.dtype in (torch.float32, torch.float64) else torch.float64
        perts = perts.reshape(perts.shape[-2] if perts.dim() >= 2 else 1, perts.shape[-1]).to(device=eng.device, dtype=dtype)
        obs = obs.reshape(-1).to(device=eng.device, dtype=dtype)
        k, p = perts.shape
        cap = max(p, 1)
        cand = torch.arange(cap, dtype=torch.int32, device=eng.device)[None]
        if p == 0:
            cand = cand - 1
        nbrs = eng.localize_from_dist(torch.zeros((1, 1, cap), dtype=torch.float64, device=eng.device), cand, [1.0])
        x = torch.zeros((1, k, 1), dtype=dtype, device=eng.device)
        gamma, prog = kernel_route(self.kernel)
        _, w = eng.analysis(x, perts, obs, nbrs, self.inf_factor, return_weights=True, rbf_gamma=gamma,
                            kernel_program=prog)
        return w[0]

    forward = __call__


class KETKFModule(ETKFModule):
    """core/ketkf.py:29-94 for the kernels of :mod:`.kernels` (every reference kernel and composition except
    ModuleKernel; NotImplementedError for anything else)."""

    def __init__(self, kernel, inf_factor: float = 1.0, engine: Optional[LetkfEngine] = None):
        super().__init__(inf_factor, engine)
        self.kernel = kernel

    def __call__(self, normed_perts, normed_obs) -> torch.Tensor:
        """One global block of any size: pair statistics over observation chunks + one k x k solve
        (``mia_ketkf_weights_*``); a LinearKernel is the ETKF (linear.py:66-67)."""
        gamma, prog = kernel_route(self.kernel, allow_feature_scale=True)
        if gamma is None and prog is None:
            return super().__call__(normed_perts, normed_obs)
        if prog is None:      # (a lone Gauss / RBF kernel: its inputs are divided by a lengthscale vector below, rbf.py:75-78)
            prog = self.kernel.program(allow_vector=True)
        eng = self.engine
        perts = torch.as_tensor(normed_perts)
        obs = torch.as_tensor(normed_obs)
        self._test_sizes(perts, obs)
        dtype = perts.dtype if perts.dtype in (torch.float32, torch.float64) else torch.float64
        k = perts.shape[-2] if perts.dim() >= 2 else 1
        perts = perts.reshape(k, perts.shape[-1]).to(device=eng.device, dtype=dtype)
        obs = obs.reshape(-1).to(device=eng.device, dtype=dtype)
        scale = getattr(self.kernel, "feature_scale", None)
        if scale is not None:          # per-observation lengthscales (rbf.py:75-78): both kernel arguments divided by l
            if len(scale) != perts.shape[-1]:
                raise ValueError("GaussKernel lengthscale vector has {0:d} entries for {1:d} observations".format(
                    len(scale), perts.shape[-1]))
            sc = torch.as_tensor(scale, device=eng.device, dtype=dtype)
            perts, obs = perts * sc, obs * sc
        return eng.ketkf_weights(perts, obs, prog, self.inf_factor)

    forward = __call__

    def __str__(self):
        return "KETKFModule({0:s}, {1})".format(str(self.kernel), self.inf_factor)

    def __repr__(self):
        return "KETKF({0:s})".format(repr(self.kernel))

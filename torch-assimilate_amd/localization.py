"""Gaspari-Cohn localisation: host-side mirror of pytassim.localization.GaspariCohn
(pytassim/localization/gaspari_cohn.py:40-136).

Constructor and ``localize_obs`` keep the reference's signature.  ``dist_func`` may be

* one of the built-in metric descriptors below (``EuclideanMetric`` / ``AbsoluteDistance``):
  then the whole localisation (distance, taper, mask, compaction) runs on the GPU through the
  cell index of csrc/localize.hip, O(G * local) instead of the reference's O(G * P);
* any Python callable with the reference's contract ``dist_func(grid_ind, obs_grid) ->
  array (P,) or tuple of arrays``: user code, evaluated on the host per grid point exactly as the
  reference does; taper, mask and compaction still run on the GPU
  (``mia_letkf_localize_from_dist_f64``).
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Sequence, Tuple, Union

import numpy as np
import torch

__all__ = ["GaspariCohn", "GaspariCohnInf", "EuclideanMetric", "AbsoluteDistance"]


class EuclideanMetric:
    """Built-in metric family: coordinate c belongs to radius group ``coord_group[c]``; the
    distance handed to radius i is the Euclidean norm over the coordinates of group i."""

    def __init__(self, coord_group: Optional[Sequence[int]] = None):
        self.coord_group = None if coord_group is None else [int(c) for c in coord_group]

    def groups(self, n_coord: int, n_r: int):
        cg = [0] * n_coord if self.coord_group is None else self.coord_group
        if len(cg) != n_coord or max(cg) >= n_r:
            raise ValueError("coord_group does not match the coordinates / radii")
        return cg

    def __call__(self, grid_ind, obs_grid):     # host evaluation, same contract as the reference
        g = np.asarray(grid_ind, dtype=np.float64).reshape(-1)
        o = np.asarray(obs_grid, dtype=np.float64)
        if o.ndim == 1:
            o = o[:, None]
        g = g[-o.shape[1]:]                       # tolerate the reference's leading time column
        cg = [0] * o.shape[1] if self.coord_group is None else self.coord_group
        out = np.zeros((max(cg) + 1, o.shape[0]))
        for c, grp in enumerate(cg):
            out[grp] += (o[:, c] - g[c]) ** 2
        return tuple(np.sqrt(out))


class AbsoluteDistance(EuclideanMetric):
    """|x_g - x_o| on one coordinate: the metric of examples/benchmark_letkf.py:85-87."""

    def __init__(self):
        super().__init__([0])


class GaspariCohn:
    _taper = 0       # MIA_TAPER_GC

    def __init__(self, length_scale: Union[float, Tuple[float, ...]], dist_func: Callable,
                 epsilon: float = 1e-5):
        self.radius = np.atleast_1d(np.asarray(length_scale, dtype=np.float64))
        self.dist_func = dist_func
        self.epsilon = float(epsilon)

    def __str__(self) -> str:
        return "GaspariCohn(l={0})".format(str(self.radius))

    def __repr__(self) -> str:
        return "GaspariCohn"

    @property
    def builtin_metric(self) -> Optional[EuclideanMetric]:
        return self.dist_func if isinstance(self.dist_func, EuclideanMetric) else None

    def localize_obs(self, grid_ind: Any, obs_grid: Any, engine=None) -> Tuple[np.ndarray, np.ndarray]:
        """Per-grid-point API of the reference (gaspari_cohn.py:97-136): (use_obs, obs_weights)
        over ALL observations; the taper is evaluated by ``mia_gaspari_cohn_f64`` on the GPU."""
        from .core import _default_engine
        eng = engine or _default_engine()
        dist = np.atleast_2d(np.asarray(self.dist_func(grid_ind, obs_grid), dtype=np.float64))
        r = torch.as_tensor(dist / self.radius[:dist.shape[0], None], dtype=torch.float64)
        w = eng.gaspari_cohn(r, self._taper).cpu().numpy()
        weights = np.prod(w, axis=0)
        return weights > self.epsilon, weights

    # ---- batched, used by the LETKF driver -------------------------------------------------
    def neighbour_lists(self, engine, grid_xyz, obs_xyz, g0: int = 0, g1: Optional[int] = None,
                        grid_info=None, obs_info=None, chunk_bytes: int = 1 << 28):
        """Neighbour lists of grid points [g0, g1).  grid_info / obs_info are what the reference
        would hand to ``dist_func`` (state_info rows / obs DataFrame); they default to the
        coordinate arrays."""
        metric = self.builtin_metric
        if metric is not None:
            nc = 1 if np.ndim(grid_xyz) == 1 else np.shape(grid_xyz)[1]
            return engine.localize(grid_xyz, obs_xyz, list(self.radius), metric.groups(nc, len(self.radius)),
                                   self.epsilon, g0, g1, taper=self._taper)
        # arbitrary callable: evaluate on the host (user code), in chunks of grid points
        G = len(grid_xyz)
        g1 = G if g1 is None else g1
        ginfo = grid_xyz if grid_info is None else grid_info
        oinfo = obs_xyz if obs_info is None else obs_info
        P = len(obs_xyz)
        n_r = len(self.radius)
        chunk = max(1, int(chunk_bytes // max(1, 8 * P * n_r)))
        parts = []
        cand = torch.arange(max(P, 1), dtype=torch.int32)
        if P == 0:
            cand = cand - 1
        for c0 in range(g0, g1, chunk):
            c1 = min(g1, c0 + chunk)
            dist = np.empty((n_r, c1 - c0, max(P, 1)))
            for gi in range(c0, c1):
                dv = np.atleast_2d(np.asarray(self.dist_func(ginfo[gi], oinfo), dtype=np.float64))
                dist[:, gi - c0, :P] = dv[:n_r]
            parts.append(engine.localize_from_dist(dist, cand[None].expand(c1 - c0, -1), list(self.radius),
                                                   self.epsilon, g0=c0, taper=self._taper))
        return engine.merge_neighbour_lists(parts)


    def tile_lists(self, engine, grid_xyz, obs_xyz, p_max: int, g0: int = 0, g1: Optional[int] = None, extra_blocks: int = 0):
        """Tile lists (engine.localize_tiles) of grid points [g0, g1) for the built-in metrics; None for a user callable."""
        metric = self.builtin_metric
        if metric is None:
            return None
        nc = 1 if np.ndim(grid_xyz) == 1 else np.shape(grid_xyz)[1]
        return engine.localize_tiles(grid_xyz, obs_xyz, list(self.radius), p_max, metric.groups(nc, len(self.radius)),
                                     self.epsilon, g0, g1, taper=self._taper, extra_blocks=extra_blocks)


class GaspariCohnInf(GaspariCohn):
    """Gaspari-Cohn correlation function with form factor infinity, C_0(z, inf, c): mirror of
    pytassim.localization.GaspariCohnInf (gaspari_cohn.py:139-254).  One length scale (the reference divides the
    distance by ``self.radius`` as a whole, :243); four polynomial pieces on [0, 0.5), [0.5, 1), [1, 1.5), [1.5, 2)
    (:176-215, assembled :244-251), evaluated by ``mia_gaspari_cohn_inf_f64`` / the taper switch of the
    localisation kernels."""
    _taper = 1       # MIA_TAPER_GC_INF

    def __init__(self, length_scale: float, dist_func: Callable, epsilon: float = 1e-5):
        if np.size(length_scale) != 1:
            raise ValueError("GaspariCohnInf takes one length scale")
        super().__init__(float(np.asarray(length_scale).reshape(-1)[0]), dist_func, epsilon)
        self._thres = [2, 1.5, 1, 0.5]

    def __str__(self) -> str:
        return "GaspariCohnInf(l={0})".format(str(self.radius))

    def __repr__(self) -> str:
        return "GaspariCohnInf"

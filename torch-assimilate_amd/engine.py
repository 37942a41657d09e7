"""Array-level host engine: torch-ROCm tensors in, torch-ROCm tensors out, every
floating-point operation of the hot path done by the gfx950 kernels behind the C ABI.

torch is plumbing here (device memory, streams); nothing in this file computes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _cabi

__all__ = ["LetkfEngine", "NeighbourLists", "ObsIndex"]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


@dataclass
class NeighbourLists:
    """Per-grid-point local observation lists of one shard [g0, g1)."""
    cnt: torch.Tensor      # (n,) int32
    idx: torch.Tensor      # (n, p_cap) int32, -1 padded
    w: torch.Tensor        # (n, p_cap) float64, sqrt(Gaspari-Cohn weight)
    p_cap: int
    p_max: int
    g0: int
    g1: int
    stats: Optional[torch.Tensor] = None   # device [max count, #truncated]; set when the read-back was deferred
    observed_p_max: Optional[int] = None

    def confirm(self) -> bool:
        """Deferred check of an assumed ``p_max`` (host sync): True when every list fitted."""
        if self.stats is None:
            return True
        p_max, n_over = (int(v) for v in self.stats.tolist())
        self.observed_p_max = p_max
        ok = n_over == 0 and p_max <= self.p_max
        if ok:
            self.stats = None
        return ok


class TileLists:
    """Tile-shaped lists of grid points [g0, g1) (csrc/mia_tiles.h): device buffer + the bound they were sized for."""

    def __init__(self, lists, p_max, g0, g1, stats, extra_blocks=0):
        self.lists, self.p_max, self.g0, self.g1, self.stats, self.extra_blocks = lists, p_max, g0, g1, stats, extra_blocks

    @property
    def ut(self):
        return max(1, (self.p_max + 8 + 15) // 16) + self.extra_blocks

    def unpack(self):
        """host view for tests: (hdr [ntile, 4], uidx [ntile, 16 ut], D [ntile, ut, 64, 4])"""
        import numpy as np
        n = self.g1 - self.g0
        ntile, ut = (n + 15) // 16, self.ut
        raw = self.lists.cpu().numpy()

        def up(x):
            return (x + 255) // 256 * 256
        o_idx = up(max(ntile, 1) * 16)
        o_d = o_idx + up(max(ntile, 1) * 16 * ut * 4)
        hdr = raw[:ntile * 16].view(np.int32).reshape(ntile, 4)
        uidx = raw[o_idx:o_idx + ntile * 16 * ut * 4].view(np.int32).reshape(ntile, 16 * ut)
        D = raw[o_d:o_d + ntile * ut * 1024].view(np.float32).reshape(ntile, ut, 64, 4)
        return hdr, uidx, D


@dataclass
class ObsIndex:
    """Cell index of one observation set for one localisation (built by mia_letkf_index_build_f64)."""
    ws: torch.Tensor
    obs: torch.Tensor          # (P, nc) float64, kept alive: the index refers to it
    radii: list
    coord_group: list
    P: int
    nc: int


class LetkfEngine:
    """One engine per process / GPU.  All work is enqueued on the current torch stream of
    ``device``; results are ordinary device tensors."""

    def __init__(self, device: Optional[torch.device] = None):
        self.lib = _cabi.lib()
        if not torch.cuda.is_available():
            raise _cabi.MiaError("no MI355X visible: the LETKF hot path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._ws = {}
        self._p_cap_hint = 32

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, tag: str, nbytes: int) -> torch.Tensor:
        cur = self._ws.get(tag)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            assert cur.data_ptr() % 256 == 0
            self._ws[tag] = cur
        return cur

    def _dev(self, a, dtype):
        t = torch.as_tensor(a)
        return t.to(device=self.device, dtype=dtype).contiguous()

    # ------------------------------------------------------------ localisation
    def gaspari_cohn(self, r: torch.Tensor, taper: int = 0) -> torch.Tensor:
        """Taper of normalised distances: taper 0 = GaspariCohn, 1 = GaspariCohnInf (MIA_TAPER_*)."""
        r = r.to(self.device).contiguous()
        out = torch.empty_like(r)
        name = "mia_gaspari_cohn_" + ("inf_" if taper else "") + {torch.float64: "f64", torch.float32: "f32"}[r.dtype]
        fn = getattr(self.lib, name)
        _cabi.check(fn(_ptr(r), r.numel(), _ptr(out), self._stream()), "mia_gaspari_cohn")
        return out

    def localize(self, grid_xyz, obs_xyz, radii: Sequence[float], coord_group: Optional[Sequence[int]] = None,
                 eps: float = 1e-5, g0: int = 0, g1: Optional[int] = None,
                 p_cap: Optional[int] = None, assume_p_max: Optional[int] = None,
                 stats_out: Optional[torch.Tensor] = None, taper: int = 0) -> NeighbourLists:
        """Neighbour lists of grid points [g0, g1).  By default the maximum list length is read back
        (one 8-byte host sync) to size the analysis launch.  With ``assume_p_max`` (e.g. the value of the
        previous cycle on the same geometry) nothing is read back here: the returned lists carry the
        assumption and ``NeighbourLists.confirm()`` checks it later, e.g. after the analysis has been
        enqueued; the analysis kernel itself flags any point whose list does not fit (never truncates)."""
        grid = self._dev(grid_xyz, torch.float64)
        obs = self._dev(obs_xyz, torch.float64)
        if grid.dim() == 1:
            grid = grid[:, None].contiguous()
        if obs.dim() == 1:
            obs = obs[:, None].contiguous()
        G, nc = grid.shape
        P = obs.shape[0]
        if P and obs.shape[1] != nc:
            raise ValueError("grid and observation coordinates differ in dimensionality")
        g1 = G if g1 is None else g1
        n = g1 - g0
        radii = [float(r) for r in (radii if hasattr(radii, "__len__") else [radii])]
        n_r = len(radii)
        coord_group = [0] * nc if coord_group is None else [int(c) for c in coord_group]
        if len(coord_group) != nc:
            raise ValueError("coord_group needs one entry per coordinate")
        cg = (C.c_int32 * nc)(*coord_group)
        rc = (C.c_double * n_r)(*radii)
        nbytes = C.c_size_t(0)
        _cabi.check(self.lib.mia_letkf_localize_workspace_bytes(P, nc, C.byref(nbytes)), "localize_workspace_bytes")
        ws = self._workspace("loc", nbytes.value)
        stats = stats_out if stats_out is not None else torch.empty(2, dtype=torch.int32, device=self.device)
        cap = int(p_cap) if p_cap is not None else self._p_cap_hint
        if assume_p_max is not None:
            cap = max(8, (int(assume_p_max) + 7) // 8 * 8)
            cnt = torch.empty(n, dtype=torch.int32, device=self.device)
            idx = torch.empty((n, cap), dtype=torch.int32, device=self.device)
            w = torch.empty((n, cap), dtype=torch.float64, device=self.device)
            _cabi.check(self.lib.mia_letkf_localize_taper_f64(
                int(taper), _ptr(grid), g0, g1, _ptr(obs), P, nc, cg, rc, n_r, float(eps), cap,
                _ptr(cnt), _ptr(idx), _ptr(w), _ptr(stats), _ptr(ws), ws.numel(), self._stream()),
                "mia_letkf_localize_f64")
            return NeighbourLists(cnt, idx, w, cap, int(assume_p_max), g0, g1, stats)
        while True:
            cnt = torch.empty(n, dtype=torch.int32, device=self.device)
            idx = torch.empty((n, cap), dtype=torch.int32, device=self.device)
            w = torch.empty((n, cap), dtype=torch.float64, device=self.device)
            _cabi.check(self.lib.mia_letkf_localize_taper_f64(
                int(taper), _ptr(grid), g0, g1, _ptr(obs), P, nc, cg, rc, n_r, float(eps), cap,
                _ptr(cnt), _ptr(idx), _ptr(w), _ptr(stats), _ptr(ws), ws.numel(), self._stream()),
                "mia_letkf_localize_f64")
            p_max, n_over = (int(v) for v in stats.tolist())   # host sync: sizes the analysis launch
            if n_over == 0:
                break
            cap = (p_max + 7) // 8 * 8          # lists were truncated: retry with room for all
        self._p_cap_hint = max(8, (p_max + 7) // 8 * 8)
        return NeighbourLists(cnt, idx, w, cap, p_max, g0, g1)

    # ------------------------------------------------------------ tile route (round 3)
    def localize_tiles(self, grid_xyz, obs_xyz, radii: Sequence[float], p_max: int,
                       coord_group: Optional[Sequence[int]] = None, eps: float = 1e-5, g0: int = 0,
                       g1: Optional[int] = None, taper: int = 0, extra_blocks: int = 0) -> "TileLists":
        """Tile lists of grid points [g0, g1) (mia_letkf_localize_tiles_f64): per tile of sixteen points the union of
        their Gaspari-Cohn lists and the sqrt(weight) matrix, in the analysis kernel's register layout.  ``p_max`` bounds
        the local observations of a point (e.g. ``NeighbourLists.p_max`` of an earlier call on the geometry),
        ``extra_blocks`` adds sixteen slots each to what a tile's union may hold; the returned object's ``stats`` holds
        [longest list, tiles whose union did not fit] on the device."""
        grid = self._dev(grid_xyz, torch.float64)
        obs = self._dev(obs_xyz, torch.float64)
        if grid.dim() == 1:
            grid = grid[:, None].contiguous()
        if obs.dim() == 1:
            obs = obs[:, None].contiguous()
        G, nc = grid.shape
        P = obs.shape[0]
        if P and obs.shape[1] != nc:
            raise ValueError("grid and observation coordinates differ in dimensionality")
        g1 = G if g1 is None else g1
        radii = [float(r) for r in (radii if hasattr(radii, "__len__") else [radii])]
        coord_group = [0] * nc if coord_group is None else [int(c) for c in coord_group]
        cg = (C.c_int32 * nc)(*coord_group)
        rc = (C.c_double * len(radii))(*radii)
        nbytes = C.c_size_t(0)
        _cabi.check(self.lib.mia_letkf_localize_workspace_bytes(P, nc, C.byref(nbytes)), "localize_workspace_bytes")
        ws = self._workspace("loc", nbytes.value)
        tb = C.c_size_t(0)
        _cabi.check(self.lib.mia_letkf_tile_lists_bytes(g1 - g0, int(p_max), int(extra_blocks), C.byref(tb)),
                    "mia_letkf_tile_lists_bytes")
        lists = torch.empty(max(tb.value, 256), dtype=torch.uint8, device=self.device)
        stats = torch.zeros(2, dtype=torch.int32, device=self.device)
        _cabi.check(self.lib.mia_letkf_localize_tiles_f64(
            int(taper), _ptr(grid), g0, g1, _ptr(obs), P, nc, cg, rc, len(radii), float(eps), int(p_max), int(extra_blocks),
            _ptr(lists), lists.numel(), _ptr(stats), _ptr(ws), ws.numel(), self._stream()), "mia_letkf_localize_tiles_f64")
        return TileLists(lists, int(p_max), g0, g1, stats, int(extra_blocks))

    def pack_split(self, Yb: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
        """[k][P] perturbations + d[P] -> (P + 1) split records (uint8 tensor; mia_letkf_pack_split_f32)."""
        Yb = Yb.to(device=self.device, dtype=torch.float32).contiguous()
        d = d.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        k, P = Yb.shape
        rb = C.c_size_t(0)
        _cabi.check(self.lib.mia_letkf_split_record_bytes(k, C.byref(rb)), "mia_letkf_split_record_bytes")
        rec = torch.empty(((P + 1) * rb.value,), dtype=torch.uint8, device=self.device)
        _cabi.check(self.lib.mia_letkf_pack_split_f32(_ptr(Yb), _ptr(d), k, P, _ptr(rec), self._stream()),
                    "mia_letkf_pack_split_f32")
        return rec

    def analysis_tiles(self, X: torch.Tensor, split_rec: torch.Tensor, P: int, tiles: "TileLists", inf_factor: float = 1.0,
                       out: Optional[torch.Tensor] = None):
        """Analysis of the tile lists' grid points from split records (mia_letkf_analysis_tiles_f32).  Returns
        (Xa [m, k, g1 - g0], flags int32 [g1 - g0], retry_count int32 [1]); declined points carry MIA_FLAG_RETRY."""
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        m, k, G = X.shape
        n = tiles.g1 - tiles.g0
        xa = out if out is not None else torch.empty((m, k, n), dtype=torch.float32, device=self.device)
        flags = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        retry = torch.zeros(1, dtype=torch.int32, device=self.device)
        _cabi.check(self.lib.mia_letkf_analysis_tiles_f32(
            _ptr(X), G, m, k, tiles.g0, tiles.g1, _ptr(split_rec), int(P), _ptr(tiles.lists), tiles.p_max, tiles.extra_blocks,
            float(inf_factor), _ptr(xa), xa.shape[-1], 0, _ptr(flags), _ptr(retry), self._stream()),
            "mia_letkf_analysis_tiles_f32")
        return xa, flags[:n], retry

    def analysis_tiles_rbf(self, X: torch.Tensor, Yb: torch.Tensor, d: torch.Tensor, tiles: "TileLists", inf_factor: float,
                           rbf_gamma: float, out: Optional[torch.Tensor] = None):
        """RBF-kernelised analysis (KETKFModule + RBFKernel(gamma), core/ketkf.py:65-94) of the tile lists' grid points from the
        float32 perturbations themselves (mia_lketkf_rbf_analysis_tiles_f32).  Returns (Xa [m, k, n], flags [n], retry_count [1])
        or None when the shape is outside the kernel (k > 40, unions of more than 64 slots)."""
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        Yb = Yb.to(device=self.device, dtype=torch.float32).contiguous()
        d = d.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        m, k, G = X.shape
        n = tiles.g1 - tiles.g0
        xa = out if out is not None else torch.empty((m, k, n), dtype=torch.float32, device=self.device)
        flags = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        retry = torch.zeros(1, dtype=torch.int32, device=self.device)
        rc = self.lib.mia_lketkf_rbf_analysis_tiles_f32(
            _ptr(X), G, m, k, tiles.g0, tiles.g1, _ptr(Yb), _ptr(d), Yb.shape[1], _ptr(tiles.lists), tiles.p_max,
            tiles.extra_blocks, float(inf_factor), float(rbf_gamma), _ptr(xa), xa.shape[-1], 0, _ptr(flags), _ptr(retry),
            self._stream())
        if rc == -3:
            return None
        _cabi.check(rc, "mia_lketkf_rbf_analysis_tiles_f32")
        return xa, flags[:n], retry

    def weights_tiles(self, X: torch.Tensor, split_rec: torch.Tensor, P: int, tiles: "TileLists", inf_factor: float = 1.0):
        """Analysis + weights of the tile lists' grid points (mia_letkf_weights_tiles_f32).  Returns (Xa [m, k, n], W [n, k, k],
        flags [n], retry_count [1]) or None when the shape is outside the kernel (unions of more than 32 slots)."""
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        m, k, G = X.shape
        n = tiles.g1 - tiles.g0
        xa = torch.empty((m, k, n), dtype=torch.float32, device=self.device)
        W = torch.empty((n, k, k), dtype=torch.float32, device=self.device)
        flags = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        retry = torch.zeros(1, dtype=torch.int32, device=self.device)
        rc = self.lib.mia_letkf_weights_tiles_f32(
            _ptr(X), G, m, k, tiles.g0, tiles.g1, _ptr(split_rec), int(P), _ptr(tiles.lists), tiles.p_max, tiles.extra_blocks,
            float(inf_factor), _ptr(xa), xa.shape[-1], 0, _ptr(W), _ptr(flags), _ptr(retry), self._stream())
        if rc == -3:
            return None
        _cabi.check(rc, "mia_letkf_weights_tiles_f32")
        return xa, W, flags[:n], retry

    def weights_retry(self, X: torch.Tensor, Yb: torch.Tensor, d: torch.Tensor, nbrs: NeighbourLists, inf_factor: float,
                      out: torch.Tensor, W: torch.Tensor, flags: torch.Tensor):
        """Eigensolver redo (analysis + weights) of the points flagged MIA_FLAG_RETRY (mia_letkf_weights_retry_f32)."""
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        m, k, G = X.shape
        rec = self.pack_obs(Yb, d, torch.float32)
        _cabi.check(self.lib.mia_letkf_weights_retry_f32(
            _ptr(X), G, m, k, nbrs.g0, nbrs.g1, _ptr(rec), rec.shape[0], _ptr(nbrs.cnt), _ptr(nbrs.idx), _ptr(nbrs.w),
            nbrs.p_cap, nbrs.p_max, float(inf_factor), 0.0, _ptr(out), out.shape[-1], 0, _ptr(W), _ptr(flags), self._stream()),
            "mia_letkf_weights_retry_f32")

    def tile_route_applies(self, X: torch.Tensor, p_max: int, extra_blocks: int = 0, rbf_gamma=None, method: str = "auto",
                           P: int = 1, n_points: Optional[int] = None, ldo: Optional[int] = None) -> bool:
        """Whether the tile route (tile lists + csrc/letkf_tile2.hip, or csrc/lketkf_tile.hip for the RBF-kernelised filter)
        takes this analysis: float32, route options on, and the shape test the C side applies before launching
        (mia_letkf_tiles_cover: ensemble size, slots of a tile's union, LDS, 32-bit byte offsets)."""
        if X.dtype != torch.float32 or method == "eig" or X.dim() != 3:
            return False
        v = C.c_int(0)
        for name in (b"tile", b"tile_lists") + (() if rbf_gamma is not None else (b"tile_split",)):
            self.lib.mia_get_option(name, C.byref(v))
            if not v.value:
                return False
        m, k, G = X.shape
        n = G if n_points is None else int(n_points)
        return bool(self.lib.mia_letkf_tiles_cover(m, k, int(p_max), int(extra_blocks), G, n if ldo is None else int(ldo), n, int(P),
                                                   float(rbf_gamma) if rbf_gamma is not None else 0.0))

    def retry_points(self, X: torch.Tensor, Yb: torch.Tensor, d: torch.Tensor, nbrs: NeighbourLists, inf_factor: float,
                     out: torch.Tensor, flags: torch.Tensor, out_offset: int = 0, rbf_gamma: Optional[float] = None):
        """Eigensolver redo of the grid points flagged MIA_FLAG_RETRY (mia_letkf_analysis_retry_f32) from per-point lists."""
        X = X.to(device=self.device, dtype=torch.float32).contiguous()
        m, k, G = X.shape
        rec = self.pack_obs(Yb, d, torch.float32)
        _cabi.check(self.lib.mia_letkf_analysis_retry_f32(
            _ptr(X), G, m, k, nbrs.g0, nbrs.g1, _ptr(rec), rec.shape[0], _ptr(nbrs.cnt), _ptr(nbrs.idx), _ptr(nbrs.w),
            nbrs.p_cap, nbrs.p_max, float(inf_factor), float(rbf_gamma) if rbf_gamma is not None else 0.0, _ptr(out), out.shape[-1],
            out_offset, _ptr(flags), self._stream()), "mia_letkf_analysis_retry_f32")

    def build_index(self, obs_xyz, radii: Sequence[float], coord_group: Optional[Sequence[int]] = None) -> ObsIndex:
        """Bin the observations into the uniform cell grid used by the fused-localisation analysis."""
        obs = self._dev(obs_xyz, torch.float64)
        if obs.dim() == 1:
            obs = obs[:, None].contiguous()
        P, nc = obs.shape
        radii = [float(r) for r in (radii if hasattr(radii, "__len__") else [radii])]
        coord_group = [0] * nc if coord_group is None else [int(c) for c in coord_group]
        nbytes = C.c_size_t(0)
        _cabi.check(self.lib.mia_letkf_localize_workspace_bytes(P, nc, C.byref(nbytes)), "localize_workspace_bytes")
        ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=self.device)
        cg = (C.c_int32 * nc)(*coord_group)
        rc = (C.c_double * len(radii))(*radii)
        _cabi.check(self.lib.mia_letkf_index_build_f64(_ptr(obs), P, nc, cg, rc, len(radii), _ptr(ws), ws.numel(),
                                                       self._stream()), "mia_letkf_index_build_f64")
        return ObsIndex(ws, obs, radii, coord_group, P, nc)

    def analysis_fused(self, X: torch.Tensor, rec: torch.Tensor, grid_xyz, index: ObsIndex, p_max_assumed: int,
                       inf_factor: float = 1.0, eps: float = 1e-5, rbf_gamma: Optional[float] = None,
                       g0: int = 0, g1: Optional[int] = None):
        """matfun analysis with the localisation fused into the kernel (no neighbour lists).  Returns
        (Xa (m, k, n), flags, finish); ``finish()`` performs the single host sync of the step and returns
        (ok, observed p_max, declined points): ok False means some grid point has more local observations
        than assumed and the shard must be redone (list route); declined points are redone by the
        eigensolver kernel inside ``finish``."""
        if X.dim() == 2:
            X = X[None]
        X = X.to(self.device).contiguous()
        if X.dtype != torch.float32 or rec.dtype != torch.float32:
            raise TypeError("the fused matfun route is float32 only")
        m, k, G = X.shape
        grid = self._dev(grid_xyz, torch.float64)
        if grid.dim() == 1:
            grid = grid[:, None].contiguous()
        g1 = G if g1 is None else g1
        n = g1 - g0
        out = torch.empty((m, k, n), dtype=torch.float32, device=self.device)
        flags = torch.empty(n, dtype=torch.int32, device=self.device)
        retry = torch.zeros(1, dtype=torch.int32, device=self.device)
        stats = torch.empty(2, dtype=torch.int32, device=self.device)
        cg = (C.c_int32 * index.nc)(*index.coord_group)
        rc = (C.c_double * len(index.radii))(*index.radii)
        gamma = float(rbf_gamma) if rbf_gamma is not None else 0.0
        _cabi.check(self.lib.mia_letkf_analysis_matfun_fused_f32(
            _ptr(X), G, m, k, g0, g1, _ptr(rec), index.P, _ptr(grid), index.nc, cg, rc, len(index.radii), float(eps),
            _ptr(index.ws), index.ws.numel(), int(p_max_assumed), float(inf_factor), gamma, _ptr(out), n, 0,
            _ptr(flags), _ptr(retry), _ptr(stats), self._stream()), "mia_letkf_analysis_matfun_fused_f32")

        def finish():
            p_max, n_over = (int(v) for v in stats.tolist())        # host sync
            n_retry = int(retry.item())
            if n_over:
                return False, p_max, n_retry
            if n_retry:     # rare: the eigensolver kernel redoes the declined points and needs explicit lists
                nb = self.localize(grid, index.obs, index.radii, index.coord_group, eps, g0, g1)
                _cabi.check(self.lib.mia_letkf_analysis_retry_f32(
                    _ptr(X), G, m, k, g0, g1, _ptr(rec), index.P, _ptr(nb.cnt), _ptr(nb.idx), _ptr(nb.w), nb.p_cap,
                    nb.p_max, float(inf_factor), gamma, _ptr(out), n, 0, _ptr(flags), self._stream()),
                    "mia_letkf_analysis_retry_f32")
            return True, p_max, n_retry
        return out, flags, finish

    def localize_from_dist(self, dist, cand_idx, radii: Sequence[float], eps: float = 1e-5,
                           g0: int = 0, taper: int = 0) -> NeighbourLists:
        """dist (n_r, n, p_cap) float64 caller-evaluated distances, cand_idx (n, p_cap) int32 (-1 pad)."""
        dist = self._dev(dist, torch.float64)
        if dist.dim() == 2:
            dist = dist[None]
        cand = self._dev(cand_idx, torch.int32)
        n_r, n, cap = dist.shape
        radii = [float(r) for r in (radii if hasattr(radii, "__len__") else [radii])]
        if len(radii) != n_r:
            raise ValueError("one radius per distance component")
        rc = (C.c_double * n_r)(*radii)
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        idx = torch.empty((n, cap), dtype=torch.int32, device=self.device)
        w = torch.empty((n, cap), dtype=torch.float64, device=self.device)
        stats = torch.empty(2, dtype=torch.int32, device=self.device)
        _cabi.check(self.lib.mia_letkf_localize_from_dist_taper_f64(
            int(taper), _ptr(dist), _ptr(cand), n, cap, rc, n_r, float(eps), _ptr(cnt), _ptr(idx), _ptr(w), _ptr(stats),
            self._stream()), "mia_letkf_localize_from_dist_f64")
        p_max = int(stats[0].item())
        return NeighbourLists(cnt, idx, w, cap, p_max, g0, g0 + n)

    def merge_neighbour_lists(self, parts) -> NeighbourLists:
        """Concatenate consecutive shards' lists (trimmed to the common maximum count)."""
        p_max = max(p.p_max for p in parts)
        cap = max(8, (p_max + 7) // 8 * 8)
        idx, w = [], []
        for p in parts:
            if p.p_cap >= cap:
                idx.append(p.idx[:, :cap])
                w.append(p.w[:, :cap])
            else:
                n = p.idx.shape[0]
                i = torch.full((n, cap), -1, dtype=torch.int32, device=self.device)
                ww = torch.zeros((n, cap), dtype=torch.float64, device=self.device)
                i[:, :p.p_cap] = p.idx
                ww[:, :p.p_cap] = p.w
                idx.append(i)
                w.append(ww)
        return NeighbourLists(torch.cat([p.cnt for p in parts]), torch.cat(idx).contiguous(),
                              torch.cat(w).contiguous(), cap, p_max, parts[0].g0, parts[-1].g1)

    # ---------------------------------------------------------------- analysis
    def pack_obs(self, Yb: torch.Tensor, d: torch.Tensor, dtype=None) -> torch.Tensor:
        """(k, P) perturbations + (P,) innovations -> obs-major records (P, kp) on the device."""
        dtype = dtype or (Yb.dtype if Yb.dtype in (torch.float32, torch.float64) else torch.float32)
        Yb = Yb.to(device=self.device, dtype=dtype).contiguous()
        d = d.to(device=self.device, dtype=dtype).contiguous().reshape(-1)
        if Yb.dim() != 2:
            raise ValueError("Yb must be (k, P)")
        k, P = Yb.shape
        if d.shape[0] != P:
            raise ValueError(
                "Observational size between ensemble ({0:d}) and observations ({1:d}) do not match!".format(
                    P, d.shape[0]))
        kp = (k + 1 + 3) // 4 * 4
        rec = torch.empty((max(P, 1), kp), dtype=dtype, device=self.device)
        sfx = "f32" if dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_letkf_pack_obs_" + sfx)
        _cabi.check(fn(_ptr(Yb), _ptr(d), k, P, _ptr(rec), self._stream()), "mia_letkf_pack_obs_" + sfx)
        return rec[:P]

    def obs_space(self, hx, y, var=None, cov=None, dtype=None, out=None, offset: int = 0, want_rec: bool = False):
        """Observation-space variables of ONE observation subset on the device
        (AssimilationInterface._get_obs_space_variables, interface/base.py:359-379):
        hx (k, P) ensemble in observation space, y (P,) observations and either var (P,) [uncorrelated R,
        observation.py:241-245] or cov (P, P) [correlated R, observation.py:247-275].
        Returns (Yb (k, P), d (P,)) [+ rec (P, kp) with ``want_rec``].  ``out=(Yb_all, d_all[, rec_all])`` with
        ``offset`` writes into the subset's slice of the stacked arrays instead (base.py:374-377)."""
        if (var is None) == (cov is None):
            raise ValueError("give exactly one of var (uncorrelated R) and cov (correlated R)")
        hx = torch.as_tensor(hx)
        dtype = dtype or (hx.dtype if hx.dtype in (torch.float32, torch.float64) else torch.float64)
        hx = hx.to(device=self.device, dtype=dtype).contiguous()
        if hx.dim() != 2:
            raise ValueError("hx must be (k, P)")
        k, P = hx.shape
        y = torch.as_tensor(y).to(device=self.device, dtype=dtype).contiguous().reshape(-1)
        if y.shape[0] != P:
            raise ValueError("Observational size between ensemble ({0:d}) and observations ({1:d}) do not match!".format(
                P, y.shape[0]))
        kp = (k + 1 + 3) // 4 * 4
        if out is None:
            Yb = torch.empty((k, P), dtype=dtype, device=self.device)
            d = torch.empty(P, dtype=dtype, device=self.device)
            rec = torch.empty((max(P, 1), kp), dtype=dtype, device=self.device) if want_rec else None
            yb_v, d_v, rec_v, ldy = Yb, d, rec, P
        else:
            Yb, d = out[0], out[1]
            rec = out[2] if len(out) > 2 else None
            if Yb.dtype != dtype or d.dtype != dtype or not Yb.is_contiguous() or Yb.shape[0] != k:
                raise ValueError("stacked outputs must be contiguous, of the working dtype and (k, P_total)")
            if offset < 0 or offset + P > Yb.shape[1] or d.shape[0] != Yb.shape[1]:
                raise ValueError("subset does not fit the stacked arrays")
            ldy = Yb.shape[1]
            yb_v, d_v = Yb[:, offset:], d[offset:]
            rec_v = rec[offset:] if rec is not None else None
        sfx = "f32" if dtype == torch.float32 else "f64"
        if var is not None:
            var = torch.as_tensor(var).to(device=self.device, dtype=dtype).contiguous().reshape(-1)
            if var.shape[0] != P:
                raise ValueError("var must have one entry per observation")
            fn = getattr(self.lib, "mia_obs_space_uncorr_" + sfx)
            _cabi.check(fn(_ptr(hx), P, _ptr(y), _ptr(var), k, P, _ptr(yb_v), ldy, _ptr(d_v), _ptr(rec_v),
                           self._stream()), "mia_obs_space_uncorr_" + sfx)
        else:
            cov = torch.as_tensor(cov).to(device=self.device, dtype=dtype).contiguous()
            if cov.shape != (P, P):
                raise ValueError("cov must be (P, P)")
            nbytes = C.c_size_t(0)
            _cabi.check(self.lib.mia_obs_space_corr_workspace_bytes(k, P, 4 if dtype == torch.float32 else 8,
                                                                    C.byref(nbytes)), "obs_space_corr_workspace_bytes")
            ws = self._workspace("obs_corr", nbytes.value)
            info = torch.zeros(1, dtype=torch.int32, device=self.device)
            fn = getattr(self.lib, "mia_obs_space_corr_" + sfx)
            _cabi.check(fn(_ptr(hx), P, _ptr(y), _ptr(cov), k, P, _ptr(yb_v), ldy, _ptr(d_v), _ptr(rec_v), _ptr(info),
                           _ptr(ws), ws.numel(), self._stream()), "mia_obs_space_corr_" + sfx)
            bad = int(info.item())
            if bad:        # numpy.linalg.cholesky raises LinAlgError here (observation.py:249)
                raise ValueError("observation covariance is not positive definite (leading minor of order %d)" % bad)
        if out is not None:
            return None
        return (Yb, d, rec[:P]) if want_rec else (Yb, d)

    # state rows per grid point up to which the eigensolver-free route is preferred.  Measured on MI355X
    # (tools/time_rows.py): it wins at every m tried (C2 m = 32: 3.3 vs 5.7 ms, C4 m = 64: 15 vs 46 ms per 2e4 points,
    # C5 m = 64: 1.2 vs 6.3 ms), because the eigensolver kernel's per-row transform is no cheaper than one
    # Chebyshev recurrence -- so there is no hand-over; the attribute stays for experiments
    MATFUN_MAX_ROWS = 1 << 30

    def analysis(self, X: torch.Tensor, Yb: Optional[torch.Tensor], d: Optional[torch.Tensor],
                 nbrs: NeighbourLists, inf_factor: float = 1.0, return_weights: bool = False,
                 rbf_gamma: Optional[float] = None, out: Optional[torch.Tensor] = None, out_offset: int = 0,
                 return_flags: bool = False, rec: Optional[torch.Tensor] = None, method: str = "auto",
                 defer_retry: bool = False, retry: Optional[torch.Tensor] = None,
                 flags: Optional[torch.Tensor] = None, kernel_program=None):
        """X (m, k, G) prior ensemble (grid fastest), Yb (k, P), d (P,) [or their packed records
        ``rec`` from :meth:`pack_obs`]: analysis of the shard described by ``nbrs``.
        Returns Xa (m, k, n) [, W (n, k, k)] [, flags (n,)].

        ``method``: "eig" = fused Jacobi eigensolver kernel (always available, the only one that can
        return the weights); "matfun" = eigensolver-free Chebyshev matrix-function route (float32, few
        state rows, no weights), with the eigensolver redoing the grid points it declines; "auto" picks
        matfun when it applies.  With ``defer_retry`` the (8-byte, synchronising) read of the decline
        counter is left to the caller: the return value gains a trailing callable that must be invoked.
        ``retry`` (1 int32, zeroed by the caller) / ``flags`` (n int32): caller-owned counter and flag
        buffers, e.g. one counter shared by the launches of several sub-ranges.
        ``rbf_gamma`` selects the RBF-kernelised core (KETKFModule with RBFKernel), ``kernel_program``
        ([(MIA_KOP_*, value), ...], see kernels.py) the kernel-expression route for every other reference kernel."""
        if X.dim() == 2:
            X = X[None]
        X = X.to(self.device).contiguous()
        dtype = X.dtype
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("state must be float32 or float64")
        m, k, G = X.shape
        if rec is None:
            if Yb.dim() != 2 or Yb.shape[0] != k:
                raise ValueError("Yb must be (k, P) with the state's ensemble size")
            rec = self.pack_obs(Yb, d, dtype)
            self._keep_rec = rec      # (a record buffer packed during HIP-graph capture must outlive the capture)
        if rec.dtype != dtype or rec.shape[1] != (k + 1 + 3) // 4 * 4:
            raise ValueError("packed records do not match the state's dtype / ensemble size")
        if method not in ("auto", "eig", "matfun"):
            raise ValueError("method must be 'auto', 'eig' or 'matfun'")
        P = rec.shape[0]
        n = nbrs.g1 - nbrs.g0
        if out is None:
            out = torch.empty((m, k, n), dtype=dtype, device=self.device)
            out_offset = 0
        ldo = out.shape[-1]
        W = torch.empty((n, k, k), dtype=dtype, device=self.device) if return_weights else None
        if flags is None:
            flags = torch.empty(n, dtype=torch.int32, device=self.device)
        elif flags.numel() != n or flags.dtype != torch.int32 or not flags.is_contiguous():
            raise ValueError("flags must be a contiguous int32 tensor with one entry per grid point")
        sfx = "f32" if dtype == torch.float32 else "f64"
        gamma = float(rbf_gamma) if rbf_gamma is not None else 0.0
        args = (_ptr(X), G, m, k, nbrs.g0, nbrs.g1, _ptr(rec), P, _ptr(nbrs.cnt), _ptr(nbrs.idx),
                _ptr(nbrs.w), nbrs.p_cap, nbrs.p_max, float(inf_factor), gamma, _ptr(out), ldo, out_offset)
        if kernel_program is not None:
            if rbf_gamma is not None:
                raise ValueError("give rbf_gamma or kernel_program, not both")
            nops = len(kernel_program)
            prog = (_cabi.KernelOp * max(nops, 1))()
            for i, (op, val) in enumerate(kernel_program):
                prog[i].op, prog[i].value = int(op), float(val)
            fn = getattr(self.lib, "mia_lketkf_kernel_analysis_packed_" + sfx)
            _cabi.check(fn(*args[:13], float(inf_factor), prog, nops, *args[15:], _ptr(W), _ptr(flags), self._stream()),
                        "mia_lketkf_kernel_analysis_packed_" + sfx)
            res = [out]
            if return_weights:
                res.append(W)
            if return_flags:
                res.append(flags)
            if defer_retry:
                res.append(lambda: 0)
            return res[0] if len(res) == 1 else tuple(res)
        if (return_weights and dtype == torch.float32 and method != "eig" and n > 0):
            # weights without an eigensolver (dual route, order <= 32); -3 = shape outside that kernel
            if retry is None:
                retry = torch.zeros(1, dtype=torch.int32, device=self.device)
            wargs = args        # (..., inf_factor, gamma, Xa, ldo, o0)
            rc = self.lib.mia_letkf_weights_matfun_f32(*wargs, _ptr(W), _ptr(flags), _ptr(retry), self._stream())
            if rc != -3:
                _cabi.check(rc, "mia_letkf_weights_matfun_f32")

                def finish_w():
                    n_retry = int(retry.item())          # host sync (8 bytes)
                    if n_retry:
                        _cabi.check(self.lib.mia_letkf_weights_retry_f32(*wargs, _ptr(W), _ptr(flags), self._stream()),
                                    "mia_letkf_weights_retry_f32")
                    return n_retry
                if not defer_retry:
                    finish_w()
                res = [out, W]
                if return_flags:
                    res.append(flags)
                if defer_retry:
                    res.append(finish_w)
                return tuple(res)
        can_matfun = dtype == torch.float32 and not return_weights and n > 0
        if method == "matfun" and not can_matfun:
            raise ValueError("the matfun route needs float32 and cannot return the weights")
        use_matfun = can_matfun and (method == "matfun" or (method == "auto" and m <= self.MATFUN_MAX_ROWS))
        finish = None
        if use_matfun:
            if retry is None:
                retry = torch.zeros(1, dtype=torch.int32, device=self.device)
            rc = self.lib.mia_letkf_analysis_matfun_f32(*args, _ptr(flags), _ptr(retry), self._stream())
            if rc == -3:              # shape outside the matfun kernels: eigensolver route
                use_matfun = False
            else:
                _cabi.check(rc, "mia_letkf_analysis_matfun_f32")

                def finish():
                    n_retry = int(retry.item())          # host sync (8 bytes)
                    if n_retry:
                        _cabi.check(self.lib.mia_letkf_analysis_retry_f32(*args, _ptr(flags), self._stream()),
                                    "mia_letkf_analysis_retry_f32")
                    return n_retry
        if not use_matfun:
            fn = getattr(self.lib, "mia_letkf_analysis_packed_" + sfx)
            _cabi.check(fn(*args, _ptr(W), _ptr(flags), self._stream()), "mia_letkf_analysis_packed_" + sfx)
        if finish is not None and not defer_retry:
            finish()
            finish = None
        res = [out]
        if return_weights:
            res.append(W)
        if return_flags:
            res.append(flags)
        if defer_retry:
            res.append(finish if finish is not None else (lambda: 0))
        return res[0] if len(res) == 1 else tuple(res)

    # ------------------------------------------------------------------- IEnKS
    def ienks_update(self, weights: torch.Tensor, Yb: Optional[torch.Tensor], d: Optional[torch.Tensor],
                     nbrs: NeighbourLists, tau: float = 1.0, epsilon: Optional[float] = None,
                     rec: Optional[torch.Tensor] = None, return_flags: bool = False, method: str = "auto"):
        """One IEnKS weight update per grid point of the shard ``nbrs`` (core/ienks.py:108-141 on the localised
        block, interface/lienks.py:75-118).  ``weights``: (k, k) shared by all points (e.g. the prior weights) or
        (n, k, k); ``epsilon`` None = transform variant, a positive value = bundle variant.  Returns (n, k, k).
        ``method`` "auto": float32 updates with tau = 1 go through the eigensolver-free weights kernel (bundle variant
        always, transform variant while Wp = I), the general kernel redoing what it declines; "eig": general kernel."""
        weights = torch.as_tensor(weights)
        dtype = weights.dtype if weights.dtype in (torch.float32, torch.float64) else torch.float64
        weights = weights.to(device=self.device, dtype=dtype).contiguous()
        k = weights.shape[-1]
        n = nbrs.g1 - nbrs.g0
        if weights.shape[-2] != k or weights.dim() not in (2, 3) or (weights.dim() == 3 and weights.shape[0] != n):
            raise ValueError("weights must be (k, k) or (n, k, k) with n the number of grid points of the shard")
        if rec is None:
            if Yb.dim() != 2 or Yb.shape[0] != k:
                raise ValueError("Yb must be (k, P) with the weights' ensemble size")
            rec = self.pack_obs(Yb, d, dtype)
        if rec.dtype != dtype or rec.shape[1] != (k + 1 + 3) // 4 * 4:
            raise ValueError("packed records do not match the weights' dtype / ensemble size")
        if not 0.0 <= float(tau) <= 1.0:
            raise ValueError("tau must lie in [0, 1]")           # bound_tensor(0, 1), interface/ienks.py:82-86
        if epsilon is not None and not float(epsilon) > 0.0:
            raise ValueError("epsilon must be positive")
        out = torch.empty((n, k, k), dtype=dtype, device=self.device)
        flags = torch.empty(n, dtype=torch.int32, device=self.device)
        wstride = k * k if weights.dim() == 3 else 0
        eps_c = float(epsilon) if epsilon is not None else 0.0
        if dtype == torch.float32 and float(tau) == 1.0 and n > 0 and method != "eig":
            # tau = 1 through the eigensolver-free weights kernel; -3 = shape outside it (primal route, order > 32)
            retry = torch.zeros(1, dtype=torch.int32, device=self.device)
            rc = self.lib.mia_lienks_update_matfun_f32(
                _ptr(weights), wstride, k, nbrs.g0, nbrs.g1, _ptr(rec), rec.shape[0], _ptr(nbrs.cnt), _ptr(nbrs.idx),
                _ptr(nbrs.w), nbrs.p_cap, nbrs.p_max, eps_c, _ptr(out), _ptr(flags), _ptr(retry), self._stream())
            if rc != -3:
                _cabi.check(rc, "mia_lienks_update_matfun_f32")
                if int(retry.item()):        # host sync (8 bytes): the general kernel redoes the declined points
                    _cabi.check(self.lib.mia_lienks_update_retry_f32(
                        _ptr(weights), wstride, k, nbrs.g0, nbrs.g1, _ptr(rec), rec.shape[0], _ptr(nbrs.cnt),
                        _ptr(nbrs.idx), _ptr(nbrs.w), nbrs.p_cap, nbrs.p_max, 1.0, eps_c, _ptr(out), _ptr(flags),
                        self._stream()), "mia_lienks_update_retry_f32")
                return (out, flags) if return_flags else out
        sfx = "f32" if dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_lienks_update_" + sfx)
        _cabi.check(fn(_ptr(weights), k * k if weights.dim() == 3 else 0, k, nbrs.g0, nbrs.g1, _ptr(rec), rec.shape[0],
                       _ptr(nbrs.cnt), _ptr(nbrs.idx), _ptr(nbrs.w), nbrs.p_cap, nbrs.p_max, float(tau),
                       float(epsilon) if epsilon is not None else 0.0, _ptr(out), _ptr(flags), self._stream()),
                    "mia_lienks_update_" + sfx)
        return (out, flags) if return_flags else out

    def apply_local_weights(self, X: torch.Tensor, W: torch.Tensor, g0: int = 0, g1: Optional[int] = None) -> torch.Tensor:
        """_apply_weights with per-grid-point weights W (g1-g0, k, k) (interface/base.py:257-278)."""
        if X.dim() == 2:
            X = X[None]
        X = X.to(self.device).contiguous()
        m, k, G = X.shape
        g1 = G if g1 is None else g1
        W = W.to(device=self.device, dtype=X.dtype).contiguous()
        if W.shape != (g1 - g0, k, k):
            raise ValueError("weights must be (grid, ensemble, ensemble_new) = (%d, %d, %d)" % (g1 - g0, k, k))
        out = torch.empty((m, k, g1 - g0), dtype=X.dtype, device=self.device)
        sfx = "f32" if X.dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_apply_local_weights_" + sfx)
        _cabi.check(fn(_ptr(X), G, m, k, g0, g1, _ptr(W), _ptr(out), g1 - g0, 0, self._stream()),
                    "mia_apply_local_weights_" + sfx)
        return out

    # ------------------------------------------------------------- global ETKF
    def etkf_weights(self, Yb: torch.Tensor, d: torch.Tensor, inf_factor: float = 1.0) -> torch.Tensor:
        Yb = Yb.to(self.device).contiguous()
        dtype = Yb.dtype
        d = d.to(device=self.device, dtype=dtype).contiguous().reshape(-1)
        k, P = Yb.shape
        if d.shape[0] != P:
            raise ValueError(
                "Observational size between ensemble ({0:d}) and observations ({1:d}) do not match!".format(
                    P, d.shape[0]))
        eb = 4 if dtype == torch.float32 else 8
        nbytes = C.c_size_t(0)
        _cabi.check(self.lib.mia_etkf_workspace_bytes(k, P, eb, C.byref(nbytes)), "etkf_workspace_bytes")
        ws = self._workspace("etkf", nbytes.value)
        W = torch.empty((k, k), dtype=dtype, device=self.device)
        flags = torch.zeros(1, dtype=torch.int32, device=self.device)
        sfx = "f32" if dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_etkf_weights_" + sfx)
        _cabi.check(fn(_ptr(Yb), _ptr(d), k, P, float(inf_factor), _ptr(W), _ptr(flags), _ptr(ws), ws.numel(),
                       self._stream()), "mia_etkf_weights_" + sfx)
        return W

    def ketkf_weights(self, Yb: torch.Tensor, d: torch.Tensor, kernel_program, inf_factor: float = 1.0) -> torch.Tensor:
        """Global kernelised ETKF weights (k, k) for ANY number of observations (KETKFModule on the full block,
        core/ketkf.py:65-94): pair statistics accumulated over observation chunks, then one k x k solve."""
        Yb = Yb.to(self.device).contiguous()
        dtype = Yb.dtype if Yb.dtype in (torch.float32, torch.float64) else torch.float64
        Yb = Yb.to(dtype)
        d = d.to(device=self.device, dtype=dtype).contiguous().reshape(-1)
        k, P = Yb.shape
        if d.shape[0] != P:
            raise ValueError(
                "Observational size between ensemble ({0:d}) and observations ({1:d}) do not match!".format(
                    P, d.shape[0]))
        nops = len(kernel_program)
        prog = (_cabi.KernelOp * max(nops, 1))()
        for i, (op, val) in enumerate(kernel_program):
            prog[i].op, prog[i].value = int(op), float(val)
        eb = 4 if dtype == torch.float32 else 8
        nbytes = C.c_size_t(0)
        _cabi.check(self.lib.mia_ketkf_workspace_bytes(k, P, eb, C.byref(nbytes)), "ketkf_workspace_bytes")
        ws = self._workspace("ketkf", nbytes.value)
        W = torch.empty((k, k), dtype=dtype, device=self.device)
        flags = torch.zeros(1, dtype=torch.int32, device=self.device)
        sfx = "f32" if dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_ketkf_weights_" + sfx)
        _cabi.check(fn(_ptr(Yb), _ptr(d), k, P, float(inf_factor), prog, nops, _ptr(W), _ptr(flags), _ptr(ws), ws.numel(),
                       self._stream()), "mia_ketkf_weights_" + sfx)
        return W

    def apply_weights(self, X: torch.Tensor, W: torch.Tensor, g0: int = 0, g1: Optional[int] = None) -> torch.Tensor:
        if X.dim() == 2:
            X = X[None]
        X = X.to(self.device).contiguous()
        m, k, G = X.shape
        g1 = G if g1 is None else g1
        W = W.to(device=self.device, dtype=X.dtype).contiguous()
        out = torch.empty((m, k, g1 - g0), dtype=X.dtype, device=self.device)
        sfx = "f32" if X.dtype == torch.float32 else "f64"
        fn = getattr(self.lib, "mia_apply_weights_" + sfx)
        _cabi.check(fn(_ptr(X), G, m, k, g0, g1, _ptr(W), _ptr(out), g1 - g0, 0, self._stream()),
                    "mia_apply_weights_" + sfx)
        return out

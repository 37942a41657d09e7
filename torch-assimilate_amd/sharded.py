"""Grid-point sharding of the LETKF analysis across the GPUs of one node.

Grid points are independent (the reference vectorises over ``grid`` with no cross-point
term, pytassim/interface/letkf.py:127-143; its only parallelism is dask chunking of that
axis, letkf.py:121-123).  One process per GPU owns a contiguous block of grid points; the
read-only observation-space inputs (Yb, d, obs coordinates) are replicated; the single
exchange step is an all-gather of the analysis ensemble (RCCL over xGMI when the process
group's backend is "nccl").
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

__all__ = ["ShardedLetkf", "block_partition", "gather_blocks"]


def block_partition(G: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, equally sized (up to the tail) blocks: rank r owns [r*n, min(G, (r+1)*n))."""
    n = (G + world - 1) // world
    return [(min(G, r * n), min(G, (r + 1) * n)) for r in range(world)]


def gather_blocks(shard: torch.Tensor, G: int, world: int, group=None) -> torch.Tensor:
    """All-gather the per-rank blocks (m, k, n_r) into the full (m, k, G) analysis.

    Blocks are padded to the common block length so a single ``all_gather_into_tensor`` (one
    large collective instead of ``world`` small ones) moves everything.
    """
    import torch.distributed as dist
    if world == 1:
        return shard
    m, k, n_r = shard.shape
    n = (G + world - 1) // world
    if n_r != n:
        pad = torch.zeros((m, k, n), dtype=shard.dtype, device=shard.device)
        pad[:, :, :n_r] = shard
        shard = pad
    # dim-0 concatenation is the layout both RCCL and gloo accept for all_gather_into_tensor
    gathered = torch.empty((world * m, k, n), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, shard.contiguous(), group=group)
    return gathered.view(world, m, k, n).permute(1, 2, 0, 3).reshape(m, k, world * n)[:, :, :G].contiguous()


class ShardedLetkf:
    """LETKF analysis of this rank's grid block + all-gather.

    ``compute_shard(X, grid_xyz, obs_xyz, Yb, d, g0, g1) -> (m, k, g1-g0)`` defaults to the
    gfx950 engine; the multi-process CPU tests inject a stand-in to exercise the sharding /
    collective logic under the gloo backend.
    """

    @property
    def dominant_kernel_name(self):
        return "letkf_sys_kernel<20, 64>" if self.method == "eig" else "letkf_cheb_kernel<20, 1, false>"

    def __init__(self, device, rank: int = 0, world: int = 1, radii: Sequence[float] = (10.0,),
                 inf_factor: float = 1.0, coord_group: Optional[Sequence[int]] = None, eps: float = 1e-5,
                 rbf_gamma: Optional[float] = None, compute_shard: Optional[Callable] = None, group=None,
                 method: str = "auto", fused_localization: bool = False, use_graph: bool = False,
                 comm_chunks: int = 4, chunk_compute: Optional[Callable] = None):
        self.device, self.rank, self.world = device, rank, world
        self.radii, self.inf_factor, self.coord_group, self.eps = list(radii), inf_factor, coord_group, eps
        self.rbf_gamma = rbf_gamma
        self.method = method
        self.fused_localization = fused_localization
        self.use_graph = use_graph
        self.comm_chunks = int(comm_chunks)
        self._chunk_compute = chunk_compute
        self._comm_stream = None
        self._graph = None
        self.graph_replays = 0
        self.last_retries = 0
        self.group = group
        self._engine = None
        self._compute = compute_shard or self._engine_shard
        self.last_p_max = 0
        self._p_max_hint = None
        self._last_flags = None

    @property
    def engine(self):
        if self._engine is None:
            from .engine import LetkfEngine
            self._engine = LetkfEngine(self.device)
        return self._engine

    def _engine_shard(self, X, grid_xyz, obs_xyz, Yb, d, g0, g1):
        eng = self.engine
        fusable = (self.fused_localization and self.method != "eig" and self._p_max_hint is not None and X.dtype == torch.float32
                   and X.shape[0] <= eng.MATFUN_MAX_ROWS and len(obs_xyz) > 0)
        if fusable:
            # steady state: index build -> one fused kernel (localisation + analysis), then ONE host sync
            # that confirms the assumed list bound while the GPU is already busy / done
            rec = eng.pack_obs(Yb, d, X.dtype)
            index = eng.build_index(obs_xyz, self.radii, self.coord_group)
            xa, flags, finish = eng.analysis_fused(X, rec, grid_xyz, index, self._p_max_hint, self.inf_factor,
                                                   self.eps, self.rbf_gamma, g0, g1)
            ok, p_max, n_retry = finish()
            self._p_max_hint = p_max
            if ok:
                self.last_p_max, self.last_retries, self._last_flags = p_max, n_retry, flags
                return xa
        # explicit neighbour lists.  After the first call on a geometry the previous maximum list length is
        # assumed, so nothing is read back before the analysis launch; the assumption is confirmed right
        # after the launch (the one host sync of the step) and the shard redone if it did not hold.
        # The steady-state launch sequence (memset, 6 index kernels, list kernel, pack, analysis = ~11 nodes,
        # ~0.4 ms of GPU work) is partly launch-bound from Python; with use_graph=True it is captured once
        # into a HIP graph and replayed while the caller keeps passing the same device buffers (0.498 ->
        # 0.457 ms per step on MI355X).  OFF by default: on this ROCm 7.2 / torch 2.10 stack the replay raised
        # 'Memory access fault by GPU' after ~15-50 replays although every captured stage replays correctly
        # on its own; root cause not isolated yet.
        key = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in (X, grid_xyz, obs_xyz, Yb, d)
                    if torch.is_tensor(t)) + (g0, g1, self._p_max_hint, self.inf_factor)
        if self.use_graph and self._p_max_hint is not None and all(torch.is_tensor(t) and t.is_cuda for t in
                                                                   (X, grid_xyz, obs_xyz, Yb, d)):
            if self._graph is None or self._graph["key"] != key:
                self._graph = self._capture(key, X, grid_xyz, obs_xyz, Yb, d, g0, g1)
            if self._graph is not None:
                gr = self._graph
                gr["graph"].replay()
                self.graph_replays += 1
                p_max, n_over = (int(v) for v in gr["nb"].stats.tolist())       # the host sync of the step
                if n_over == 0 and p_max <= gr["nb"].p_max:
                    self.last_retries = gr["finish"]()
                    self._p_max_hint = p_max
                    self.last_p_max, self._last_flags = p_max, gr["flags"]
                    return gr["xa"]
                self._graph = None          # assumption broken: fall through to the eager route
                self._p_max_hint = None
        nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1,
                          assume_p_max=self._p_max_hint)
        xa, flags, finish = eng.analysis(X, Yb, d, nb, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                         return_flags=True, method=self.method, defer_retry=True)
        if not nb.confirm():
            nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1)
            xa, flags, finish = eng.analysis(X, Yb, d, nb, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                             return_flags=True, method=self.method, defer_retry=True)
        self.last_retries = finish()
        self._p_max_hint = nb.observed_p_max if nb.observed_p_max is not None else nb.p_max
        self.last_p_max = nb.p_max
        self._last_flags = flags
        return xa

    def _capture(self, key, X, grid_xyz, obs_xyz, Yb, d, g0, g1):
        """Record the steady-state launch sequence of this shard into a HIP graph (None if capture fails)."""
        eng = self.engine
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1,
                                      assume_p_max=self._p_max_hint)
                    xa, flags, finish = eng.analysis(X, Yb, d, nb, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                                     return_flags=True, method=self.method, defer_retry=True)
            torch.cuda.current_stream(self.device).wait_stream(side)
            return dict(key=key, graph=graph, nb=nb, xa=xa, flags=flags, finish=finish,
                        keep=(X, grid_xyz, obs_xyz, Yb, d))
        except Exception as err:      # capture unsupported in this environment: stay eager
            import warnings
            warnings.warn("HIP graph capture failed (%s); using eager launches" % (err,), RuntimeWarning)
            self.use_graph = False
            return None

    def assimilate(self, X, grid_xyz, obs_xyz, Yb, d) -> torch.Tensor:
        G = X.shape[-1]
        g0, g1 = block_partition(G, self.world)[self.rank]
        if self.world > 1 and self.comm_chunks > 1:
            return self._assimilate_overlapped(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1)
        shard = self._compute(X, grid_xyz, obs_xyz, Yb, d, g0, g1)
        return gather_blocks(shard, G, self.world, self.group)

    # ------------------------------------------------------------------ compute / exchange overlap
    def _chunk_engine(self, X, grid_xyz, obs_xyz, Yb, d, g0, g1, state):
        """Analysis of sub-range [g0, g1) of this rank's block with the shard-wide preparation (packed
        records, neighbour lists) done once and kept in ``state``.  Returns (Xa chunk, finish)."""
        eng = self.engine
        if "nb" not in state:
            b0, b1 = state["block"]
            state["rec"] = eng.pack_obs(Yb, d, X.dtype)
            nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, b0, b1,
                              assume_p_max=self._p_max_hint)
            state["nb"] = nb
        nb, (b0, _) = state["nb"], state["block"]
        from .engine import NeighbourLists
        sub = NeighbourLists(nb.cnt[g0 - b0:g1 - b0], nb.idx[g0 - b0:g1 - b0], nb.w[g0 - b0:g1 - b0],
                             nb.p_cap, nb.p_max, g0, g1)
        xa, flags, finish = eng.analysis(X, None, None, sub, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                         rec=state["rec"], return_flags=True, method=self.method, defer_retry=True)
        state.setdefault("flags", []).append(flags)
        return xa, finish

    def _assimilate_overlapped(self, X, grid_xyz, obs_xyz, Yb, d, G, g0, g1):
        """The rank's block is analysed in ``comm_chunks`` pieces; the all-gather of piece c (RCCL, on a side
        stream) runs while piece c+1 is being analysed, so at 8 GPUs the 16 MB-per-rank exchange hides behind
        the compute instead of adding to it.  One permute-copy at the end restores (m, k, G)."""
        import torch.distributed as dist
        world, C = self.world, self.comm_chunks
        n = (G + world - 1) // world                 # common block length
        nc = (n + C - 1) // C                        # common chunk length
        m, k = X.shape[0], X.shape[1]
        cuda = X.is_cuda
        chunk_fn = self._chunk_compute or self._chunk_engine
        gath = torch.empty((C, world * m, k, nc), dtype=X.dtype, device=X.device)
        state = {"block": (g0, g1)}
        finishes, bufs = [], []
        if cuda:
            comp = torch.cuda.current_stream(X.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=X.device)
            comm = self._comm_stream
            comm.wait_stream(comp)
        for c in range(C):
            c0, c1 = min(g1, g0 + c * nc), min(g1, g0 + (c + 1) * nc)
            buf = torch.zeros((m, k, nc), dtype=X.dtype, device=X.device) if c1 - c0 < nc else None
            xa = None
            if c1 > c0:
                xa, fin = chunk_fn(X, grid_xyz, obs_xyz, Yb, d, c0, c1, state)
                finishes.append(fin)
                if buf is None:
                    buf = xa
                else:
                    buf[:, :, :c1 - c0] = xa
            bufs.append((buf, xa))
            if cuda:
                ev = torch.cuda.Event()
                ev.record(comp)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    dist.all_gather_into_tensor(gath[c], buf.contiguous(), group=self.group)
            else:
                dist.all_gather_into_tensor(gath[c], buf.contiguous(), group=self.group)
        nb = state.get("nb")
        redo = 0
        if nb is not None and not nb.confirm():
            redo = 1                                  # list bound broken somewhere in this block
        n_retry = sum(f() for f in finishes)          # (host sync; declined points were redone in place)
        self.last_retries = n_retry
        if cuda:
            comp.wait_stream(comm)
        # rare paths must be agreed on by all ranks: redone chunks need a second exchange
        if self._chunk_compute is None:
            flag = torch.tensor([redo, 1 if n_retry else 0], dtype=torch.int32, device=X.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            any_redo, any_retry = (int(v) for v in flag.tolist())
            if any_redo:
                self._p_max_hint = None
                shard = self._engine_shard(X, grid_xyz, obs_xyz, Yb, d, g0, g1)
                return gather_blocks(shard, G, world, self.group)
            if any_retry:                             # buffers were updated in place by the retry kernels
                for c, (buf, xa) in enumerate(bufs):
                    if xa is not None and xa is not buf:
                        buf[:, :, :xa.shape[-1]] = xa
                    dist.all_gather_into_tensor(gath[c], buf.contiguous(), group=self.group)
            if nb is not None:
                self._p_max_hint = nb.observed_p_max if nb.observed_p_max is not None else nb.p_max
                self.last_p_max = nb.p_max
                self._last_flags = torch.cat(state["flags"]) if state.get("flags") else None
        out = gath.view(C, world, m, k, nc).permute(2, 3, 1, 0, 4).reshape(m, k, world * C * nc)
        if C * nc != n:                               # chunk padding inside each rank's block
            out = out.view(m, k, world, C * nc)[:, :, :, :n].reshape(m, k, world * n)
        return out[:, :, :G].contiguous()

    def mean_degree(self):
        """Mean Chebyshev degree of the last matfun launch (flags bits 8-15), None for the eigensolver route."""
        if self._last_flags is None or self.method == "eig":
            return None
        return float(((self._last_flags >> 8) & 0xff).float().mean().item())

    def last_flags_ok(self) -> bool:
        return self._last_flags is None or int((self._last_flags & 0xff).max().item()) == 0

    def time_stages(self, X, grid_xyz, obs_xyz, Yb, d, reps: int = 10):
        """HIP-event timing (on torch's current stream = the launch stream) of each stage of this
        rank's shard; returns (mean ms of the dominant analysis kernel, dict of stage means)."""
        eng = self.engine
        G = X.shape[-1]
        g0, g1 = block_partition(G, self.world)[self.rank]
        ev = lambda: torch.cuda.Event(enable_timing=True)
        fused = (self.fused_localization and self.method != "eig" and self._p_max_hint is not None
                 and X.dtype == torch.float32)
        names = ["pack_obs", "obs_index_build" if fused else "localize(index+lists, incl. host sync)", "analysis_kernel"]
        acc = dict.fromkeys(names, 0.0)
        for _ in range(reps):
            e = [ev() for _ in range(4)]
            e[0].record()
            rec = eng.pack_obs(Yb, d, X.dtype)
            e[1].record()
            if fused:
                index = eng.build_index(obs_xyz, self.radii, self.coord_group)
                e[2].record()
                _, _, fin = eng.analysis_fused(X, rec, grid_xyz, index, self._p_max_hint, self.inf_factor, self.eps,
                                               self.rbf_gamma, g0, g1)
            else:
                nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1)
                e[2].record()
                _, fin = eng.analysis(X, None, None, nb, self.inf_factor, rbf_gamma=self.rbf_gamma, rec=rec,
                                      method=self.method, defer_retry=True)
            e[3].record()
            fin()
            torch.cuda.synchronize()
            for name, a, b in zip(names, e[:-1], e[1:]):
                acc[name] += a.elapsed_time(b)
        stage = {k: v / reps for k, v in acc.items()}
        return stage["analysis_kernel"], stage

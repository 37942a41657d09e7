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

import os
import torch

__all__ = ["ShardedLetkf", "PendingStep", "block_partition", "gather_blocks"]


TILE_BOX_OVERFLOW = 1 << 30        # MIA_TILE_BOX_OVERFLOW (include/mia_letkf.h)
STATUS_NONFINITE = 128             # MIA_STEP_STATUS_NONFINITE: (with STATUS_SAMPLED) some point carries MIA_FLAG_NONFINITE
STATUS_SAMPLED = 64                # MIA_STEP_STATUS_SAMPLED: counters[0] / [4] of this step are a sampled maximum (fused kernel)


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


from . import _cabi          # (ctypes declarations only: the library itself is loaded on first use)

_F32, _F64 = torch.float32, torch.float64
_METHODS = {"auto": 0, "eig": 1, "matfun": 2}


def _steady_inputs(f, X, grid_xyz, obs_xyz, Yb, d) -> bool:
    """The inputs already are what the library reads -- device, dtype, contiguous -- and have the shapes of the state `f` a general
    call recorded: what the short paths of ShardedLetkf (_submit_fast, _run_fast) require before they skip the general set-up."""
    try:
        dev = f["device"]
        return (X.dtype is _F32 and Yb.dtype is _F32 and d.dtype is _F32 and grid_xyz.dtype is _F64 and obs_xyz.dtype is _F64
                and X.shape == f["xs"] and Yb.shape == f["ys"] and d.shape == f["ds"] and grid_xyz.shape == f["gs"]
                and obs_xyz.shape == f["os"] and X.is_contiguous() and Yb.is_contiguous() and d.is_contiguous()
                and grid_xyz.is_contiguous() and obs_xyz.is_contiguous()
                and X.device == dev and Yb.device == dev and d.device == dev and grid_xyz.device == dev and obs_xyz.device == dev)
    except AttributeError:
        return False


class PendingStep:
    """Handle of a step enqueued by :meth:`ShardedLetkf.submit`."""

    def __init__(self, runner, state, out=None):
        self._runner, self._st, self._out = runner, state, out
        self.batch_n = 1        # steps that shared this step's analysis launch (known once the step has been joined)

    def result(self) -> torch.Tensor:
        """The (m, k, G) analysis; performs the step's host read-back the first time it is called."""
        if self._out is None:
            self._runner._native_finish(self)
        return self._out


class ShardedLetkf:
    """LETKF analysis of this rank's grid block + all-gather.

    ``compute_shard(X, grid_xyz, obs_xyz, Yb, d, g0, g1) -> (m, k, g1-g0)`` defaults to the
    gfx950 engine; the multi-process CPU tests inject a stand-in to exercise the sharding /
    collective logic under the gloo backend.
    """

    @property
    def _last_flags(self):
        """Per-point flags of the last completed step (None before the first)."""
        lz = self._last_flags_lazy
        if lz is not None:
            self._last_flags_val, self._last_flags_lazy = lz[0][:lz[1]], None
        return self._last_flags_val

    @_last_flags.setter
    def _last_flags(self, value):
        self._last_flags_val, self._last_flags_lazy = value, None
        self.last_flags_summary = None          # (a route that does not report through the status word)

    @property
    def dominant_kernel_name(self):
        """The analysis kernel of this runner's last completed step as the library launched it and as rocprofv3 names it
        (mia_last_analysis_kernel: template arguments included); before the first step, a prediction from the route options."""
        return self._last_kernel or self._predicted_kernel_name()

    def _note_kernel(self):
        from . import _cabi
        name = _cabi.last_analysis_kernel()
        if name:
            self._last_kernel = name

    def _predicted_kernel_name(self):
        if self.method == "eig":
            return "letkf_sys_kernel<20, 64>"
        import ctypes as C
        v, sp = C.c_int(1), C.c_int(1)
        self.engine.lib.mia_get_option(b"tile", C.byref(v))
        self.engine.lib.mia_get_option(b"tile_split", C.byref(sp))
        tl = C.c_int(1)
        self.engine.lib.mia_get_option(b"tile_lists", C.byref(tl))
        if v.value and tl.value and not self._no_tile_lists and self.rbf_gamma is not None:
            return "lketkf_tile_kernel<10, 2, false>"
        if v.value and sp.value and tl.value and not self._no_tile_lists and self.native_step:
            fu, bk = C.c_int(1), C.c_int(1)
            self.engine.lib.mia_get_option(b"tile_fused", C.byref(fu))
            self.engine.lib.mia_get_option(b"bucket_index", C.byref(bk))
            if fu.value and bk.value and not self._scan_index and self.fuse_tile_lists is True:
                return "letkf_tile2f_kernel<2, 3, 1>"      # (every step outside a geometry epoch: the wavefronts localise themselves)
            return "letkf_tile2_kernel<2, 3, false>"
        return ("letkf_tile_kernel<2, 3, false, %s>" % ("true" if sp.value else "false")) if v.value else "letkf_cheb_kernel<20, 1, false>"

    def _fuse_now(self, pipelined: bool) -> bool:
        # "auto" (the default) = fused wherever the library can (unions of at most 32 slots, bucket index, no geometry epoch --
        # it decides per step and builds lists in memory otherwise), for steps one at a time and in flight alike: with three
        # analysis streams the fused kernels of consecutive steps share the chip, 2.08e9 against 1.72e9 analyses/s at config 2
        # (profiles/r05_stream_ab.txt; round 4 kept lists in memory for steps in flight to keep the analysis launch short)
        if self.fuse_tile_lists == "auto":
            return True
        return bool(self.fuse_tile_lists)

    @property
    def exchange_route(self):
        if self._native is not None and self._native.get("peer"):
            return "direct peer writes into IPC-mapped result buffers"
        return "RCCL all-gather + placement kernel"

    # ------------------------------------------------------------------ direct exchange (peer-mapped result buffers)
    @staticmethod
    def _wrap_device(ptr: int, shape, device) -> torch.Tensor:
        """float32 tensor over library-owned device memory (no copy, no ownership)."""
        class _Ext:
            pass
        e = _Ext()
        e.__cuda_array_interface__ = dict(shape=tuple(shape), typestr="<f4", data=(int(ptr), False), version=3, strides=None)
        return torch.as_tensor(e, device=device)

    def _peer_setup(self, st, m: int, k: int, G: int):
        """Collective (every rank, same step): allocate the slot result buffers, exchange their IPC handles through
        torch.distributed, map the peers, and run a pattern self-test of the exchange on every slot.  Any failure on any
        rank leaves every rank on the RCCL route.  Returns the per-slot result tensors or None."""
        import ctypes as C
        import warnings
        import torch.distributed as dist
        from . import _cabi
        lib, comm, dev = self.engine.lib, st["comm"], self.device
        n_slots = len(st["slots"])
        nbytes = m * k * G * 4
        handles = C.create_string_buffer(64 * (n_slots + 1))

        def agree(ok):
            # (a CPU tensor under gloo, which the two-processes-on-one-GPU test uses for the rendezvous)
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend(self.group) == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return bool(t.item())

        with torch.cuda.device(dev):
            ok = lib.mia_comm_peer_alloc(comm, nbytes, n_slots, handles) == 0
        why = "buffer allocation / hipIpcGetMemHandle"
        if agree(ok):
            tables = [None] * self.world
            dist.all_gather_object(tables, handles.raw, group=self.group)
            with torch.cuda.device(dev):
                ok = lib.mia_comm_peer_open(comm, C.create_string_buffer(b"".join(tables), 64 * (n_slots + 1) * self.world)) == 0
            why = "hipIpcOpenMemHandle"
            if agree(ok):
                bufs = [self._wrap_device(lib.mia_comm_peer_buffer(comm, s), (m, k, G), dev) for s in range(n_slots)]
                ok = self._peer_selftest(st, bufs, m * k, G)
                why = "exchange self-test"
                if agree(ok):
                    return bufs
        warnings.warn("direct peer exchange unavailable (%s failed on some rank: %s); using the RCCL all-gather"
                      % (why, lib.mia_comm_last_error().decode()), RuntimeWarning)
        return None

    def _peer_selftest(self, st, bufs, rows: int, G: int) -> bool:
        """Every rank writes a rank-specific pattern into its block of every slot buffer, runs the bare exchange and checks
        that all blocks of all ranks arrived (mapping, protocol, visibility -- before any analysis depends on them)."""
        import ctypes as C
        from .engine import _ptr
        lib, comm = self.engine.lib, st["comm"]
        parts = block_partition(G, self.world)
        g0, g1 = parts[self.rank]
        col = torch.arange(G, device=self.device, dtype=torch.float32)
        ctr = torch.zeros(8, dtype=torch.int32, device=self.device)
        ok = True
        for s, buf in enumerate(bufs):
            flat = buf.view(rows, G)
            flat.fill_(-1.0)
            pat = lambda r, a, b: (r + 1) * 4096.0 + (col[a:b] % 61.0)[None] + (torch.arange(rows, device=self.device) % 64)[:, None] * 64.0
            flat[:, g0:g1] = pat(self.rank, g0, g1)
            torch.cuda.synchronize(self.device)
            ctr.zero_()
            ctr[0] = 100 + self.rank
            rc = lib.mia_comm_peer_exchange(comm, s, rows, G, g0, g1, _ptr(ctr), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            torch.cuda.synchronize(self.device)
            want = torch.empty_like(flat)
            for r, (a, b) in enumerate(parts):
                want[:, a:b] = pat(r, a, b)
            c = ctr.tolist()
            ok = ok and rc == 0 and bool(torch.equal(flat, want)) and c[4] == 100 + self.world - 1 and c[7] == 0
        return ok

    def __init__(self, device, rank: int = 0, world: int = 1, radii: Sequence[float] = (10.0,),
                 inf_factor: float = 1.0, coord_group: Optional[Sequence[int]] = None, eps: float = 1e-5,
                 rbf_gamma: Optional[float] = None, compute_shard: Optional[Callable] = None, group=None,
                 method: str = "auto", fused_localization: bool = False,
                 comm_chunks: int = 4, chunk_compute: Optional[Callable] = None, native_step: bool = True,
                 max_in_flight: int = 3, peer_exchange: str = "auto", copy_results: bool = True, prep_streams: int = 3,
                 analysis_streams: int = 3, gather: bool = True, fuse_tile_lists="auto", prep_priority: int = 0):
        self.device, self.rank, self.world = device, rank, world
        self.radii, self.inf_factor, self.coord_group, self.eps = list(radii), inf_factor, coord_group, eps
        self.rbf_gamma = rbf_gamma
        # gather=False (world > 1): assimilate() / submit() return THIS RANK'S BLOCK of the analysis, (m, k, g1 - g0) with
        # (g0, g1) = block_partition(G, world)[rank]: the analysis stays chunked along `grid`, as the reference's dask arrays do
        # (interface/letkf.py:118-131); nothing crosses a link and no communicator is created
        self.gather = bool(gather)
        self.method = method
        self.fused_localization = fused_localization
        # native step driver, tile route: True / "auto" = the analysis wavefronts localise their tiles themselves
        # (csrc/letkf_tile2f.hip, option "tile_fused": no list kernel, no lists in memory; same bits) wherever the shape allows,
        # False = tile lists in memory first (round 4's route for steps in flight: 1.72e9 against 2.08e9 analyses/s at config 2
        # with three analysis and three preparation streams, profiles/r05_stream_ab.txt)
        self.fuse_tile_lists = fuse_tile_lists
        self.comm_chunks = int(comm_chunks)
        self._chunk_compute = chunk_compute
        self._comm_stream = None
        self._obuf = None
        self.native_step = native_step
        self._native = None
        self._in_flight = []
        self._time_next = False        # time_next_step(): the next native step brackets its analysis kernel with events
        self.kernel_timings = []       # [(start, stop)] torch events recorded on the analysis stream by the library
        self.kernel_batch = {}         # (start, stop) -> steps in the launch those events bracket (launch coalescing; absent: 1)
        self.last_batch_n = 1          # steps that shared the analysis launch of the step finished last
        # (eight measured best at config 2: 2.61e9 analyses/s against 2.41e9 with sixteen -- more steps in flight are more kernels sharing
        #  the chip, profiles/r05_coalesce.txt; rounds 1-4 capped the argument at eight silently)
        self.max_in_flight = max(1, min(int(max_in_flight), 16))
        self.prep_streams = max(0, min(int(prep_streams), 8))      # steps in flight: preparation streams taken in turn (0: none, tools)
        self.analysis_streams = max(1, min(int(analysis_streams), 4))
        self.prep_priority = int(prep_priority)      # HIP stream priority of the preparation streams (0 normal, -1 high)
        self.peer_rewaits = 3          # direct exchange: how often a waiter that gave up waits again before the peer is called dead
        self._peer_rewait_hook = None  # (tests: called before every re-wait)
        # exchange of the analysis blocks at world > 1: "auto" = direct peer writes into library-owned, IPC-mapped result
        # buffers when the node allows it and a self-test of the mapping passes, RCCL all-gather otherwise; "off" = RCCL.
        # With the direct route a result lives in its pipeline slot's buffer: copy_results (default) hands out a copy,
        # False the buffer itself, valid until max_in_flight further steps have been submitted.
        self.peer_exchange = peer_exchange
        self.copy_results = copy_results
        self._submitted = 0
        self.reused_steps = 0          # steps that ran on the tile lists of their geometry epoch (submit(..., geometry_id=))
        self._force_comm = False      # tests / tools: a one-rank RCCL communicator drives the exchange route
        self.native_steps = 0
        self.last_retries = 0
        self.group = group
        self._engine = None
        self._compute = compute_shard or self._engine_shard
        self.last_p_max = 0
        self._p_max_hint = None
        self._last_flags_val, self._last_flags_lazy = None, None
        self._fast = None              # what _submit_fast needs: recorded by _native_submit once a steady state exists
        self._fast_serial = None       # ... and what _run_fast needs (one step at a time)
        self.last_flags_summary = None # OR of the flag bits (low byte) of the last step's points when known without a scan
        # tile route of the native step driver (tile-shaped lists + split records, csrc/letkf_tile2.hip): switched off for this
        # object once a step reports tiles whose union does not fit their slots (scattered grids) -- per-point lists then
        self._no_tile_lists = False
        self._tile_extra = 0          # row blocks of sixteen slots added to the tiles' unions (MIA_STEP_TILE_EXTRA)
        self._fresh_box_once = False  # the next step recomputes the observations' bounding box (MIA_STEP_FRESH_BOX)
        self._scan_index = False      # scan-based observation index from now on (a cell overflowed its bucket: MIA_STEP_SCAN_INDEX)
        self._last_kernel = None      # analysis kernel of the last completed step (mia_last_analysis_kernel)

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
        # (Round 1 could replay this launch sequence from a HIP graph; the replay faulted after 15-50 replays, the cause
        #  was never isolated, and the one-call native step driver removed the launch overhead the graph was meant to hide:
        #  the path is gone, see DESIGN.md.)
        nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1,
                          assume_p_max=self._p_max_hint)
        if len(obs_xyz) > 0 and not self._no_tile_lists and torch.is_tensor(X):
            # the route the native step driver takes on this geometry (tile lists + split records): the entry-by-entry calls
            # take it too, so that a step's result does not depend on which call of a run it is
            xa = self._engine_shard_tiles(X, grid_xyz, obs_xyz, Yb, d, g0, g1, nb)
            if xa is not None:
                return xa
        xa, flags, finish = eng.analysis(X, Yb, d, nb, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                         return_flags=True, method=self.method, defer_retry=True)
        self._note_kernel()
        if not nb.confirm():
            nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1)
            xa, flags, finish = eng.analysis(X, Yb, d, nb, self.inf_factor, rbf_gamma=self.rbf_gamma,
                                             return_flags=True, method=self.method, defer_retry=True)
        self.last_retries = finish()
        self._p_max_hint = nb.observed_p_max if nb.observed_p_max is not None else nb.p_max
        self.last_p_max = nb.p_max
        self._last_flags = flags
        return xa

    def _engine_shard_tiles(self, X, grid_xyz, obs_xyz, Yb, d, g0, g1, nb):
        """Exact-list call on the tile route: the lists' true maximum sizes the tiles; a union that does not fit gets sixteen
        more slots per tile until the format has no more (then None: per-point lists).  Declined points are redone by the
        eigensolver kernel from the per-point lists ``nb``."""
        eng = self.engine
        P = int(Yb.shape[1])
        while eng.tile_route_applies(X, nb.p_max, self._tile_extra, self.rbf_gamma, self.method, P=P, n_points=g1 - g0):
            tiles = eng.localize_tiles(grid_xyz, obs_xyz, self.radii, nb.p_max, self.coord_group, self.eps, g0, g1,
                                       extra_blocks=self._tile_extra)
            n_over = int(tiles.stats[1].item())                 # host sync (first call on a geometry only)
            if n_over == 0:
                break
            if n_over & TILE_BOX_OVERFLOW:                      # a tile's cell box is too large: slots cannot help
                self._no_tile_lists, self._tile_extra = True, 0
                return None
            self._tile_extra += 1
        else:
            if self._tile_extra:
                self._no_tile_lists, self._tile_extra = True, 0
            return None
        Xc = X.to(eng.device, torch.float32).contiguous()
        if self.rbf_gamma is not None:      # RBF-kernelised filter: from the perturbations themselves (csrc/lketkf_tile.hip)
            res = eng.analysis_tiles_rbf(Xc, Yb, d, tiles, self.inf_factor, self.rbf_gamma)
        else:
            res = eng.analysis_tiles(Xc, eng.pack_split(Yb, d), P, tiles, self.inf_factor)
        if res is None:
            return None
        xa, flags, retry = res
        self._note_kernel()
        n_retry = int(retry.item())
        if n_retry:
            if not nb.confirm():                                # (lists built on an assumed bound that did not hold)
                nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1)
            eng.retry_points(Xc, Yb, d, nb, self.inf_factor, xa, flags, rbf_gamma=self.rbf_gamma)
        self.last_retries = n_retry
        self._p_max_hint = max(int(tiles.stats[0].item()), 0)
        self.last_p_max = nb.p_max
        self._last_flags = flags
        return xa

    def assimilate(self, X, grid_xyz, obs_xyz, Yb, d, geometry_id=None) -> torch.Tensor:
        G = X.shape[-1]
        g0, g1 = block_partition(G, self.world)[self.rank]
        if (self.native_step and self._compute == self._engine_shard and self._chunk_compute is None
                and torch.is_tensor(X) and X.is_cuda and X.dtype == torch.float32 and X.dim() == 3
                and not self.fused_localization):
            f = self._fast_serial
            if f is not None and geometry_id is None:
                out = self._run_fast(f, X, grid_xyz, obs_xyz, Yb, d)
                if out is not None:
                    return out
            return self._assimilate_native(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1, geometry_id)
        if self.world > 1 and self.comm_chunks > 1 and self.gather:
            return self._assimilate_overlapped(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1)
        shard = self._compute(X, grid_xyz, obs_xyz, Yb, d, g0, g1)
        return gather_blocks(shard, G, self.world, self.group) if self.gather else shard

    # ------------------------------------------------------------------ native step driver
    def _native_comm(self):
        """RCCL communicator owned by the C library (created once): rank 0 of the group draws the unique id,
        torch.distributed carries its 128 bytes to the others, every rank joins on its own device."""
        import ctypes as C
        import os
        import torch.distributed as dist
        from . import _cabi
        lib = self.engine.lib
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        # binding RCCL is a local step that can fail on one rank only: agree on it BEFORE anybody enters a collective
        # (id broadcast, ncclCommInitRank) that a failed rank would never join
        buf = C.create_string_buffer(128)
        err = None
        try:
            _cabi.check(lib.mia_comm_load(path.encode() if os.path.exists(path) else None), "mia_comm_load")
            if self.rank == 0:
                _cabi.check(lib.mia_comm_unique_id(buf), "mia_comm_unique_id")
        except (_cabi.MiaError, OSError) as e:
            err = e
        if dist.is_initialized() and self.world > 1:
            t = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            if int(t.item()) == 0:
                raise _cabi.MiaError("RCCL could not be bound on every rank (%s)" % (err if err is not None else "another rank failed"))
        elif err is not None:
            raise err
        box = [buf.raw]
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast_object_list(box, src=src, group=self.group)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(lib.mia_comm_create(C.create_string_buffer(box[0], 128), self.rank, self.world,
                                            C.byref(handle)), "mia_comm_create")
        return handle

    def _native_state(self):
        if self._native is None:
            st = dict(comm=None, stream=None, slots=[{} for _ in range(self.max_in_flight)])
            if (self.world > 1 and self.gather) or self._force_comm:
                st["comm"] = self._native_comm()
            elif self.world > 1:                       # the block partition only (no RCCL, no exchange)
                import ctypes as C
                from . import _cabi
                st["part"] = C.c_void_p()
                _cabi.check(self.engine.lib.mia_comm_create_partition(self.rank, self.world, C.byref(st["part"])),
                            "mia_comm_create_partition")
            # exchange stream at high priority: its (few, multi-wave) RCCL workgroups should be placed ahead of the
            # bulk analysis kernel's next workgroups when wave slots fall free, not queue behind 1e5 of them
            # (single GPU: this stream only carries the 32-byte counter read-back -- normal priority like every other stream of the
            #  step: with high-priority preparation / read-back streams the loop ran 10 % slower, tools/ab_prio_streams.sh)
            st["stream"] = torch.cuda.Stream(device=self.device, priority=-1 if st["comm"] is not None else 0)
            if st["comm"] is not None:
                # gathered pieces are copied into the result on a stream of their own, so that with steps in flight the
                # next all-gather starts as soon as the previous one has landed
                import ctypes as C
                from . import _cabi
                st["xstream"] = torch.cuda.Stream(device=self.device)
                _cabi.check(self.engine.lib.mia_comm_set_place_stream(st["comm"], C.c_void_p(st["xstream"].cuda_stream)),
                            "mia_comm_set_place_stream")
            self._native = st
        return self._native

    def _native_available(self) -> bool:
        """Create the library-owned communicator if needed.  Should that fail on ANY rank (RCCL library not found,
        ncclCommInitRank error), every rank falls back to the torch.distributed exchange route -- the decision is
        all-reduced so that no rank waits in a collective the others never enter."""
        if self._native is not None or self.world == 1 or not self.gather:
            return True
        import warnings
        import torch.distributed as dist
        from . import _cabi
        ok, err = 1, None
        try:
            self._native_state()
        except (_cabi.MiaError, OSError, RuntimeError) as e:      # noqa: PERF203
            ok, err = 0, e
            self._native = None
        t = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        if int(t.item()) == 0:
            if self._native is not None:
                self.close()
            warnings.warn("library-owned RCCL communicator unavailable (%s): using the torch.distributed exchange route"
                          % (err if err is not None else "another rank failed"), RuntimeWarning)
            self.native_step = False
            return False
        return True

    def close(self):
        """Release the library-owned communicator (idempotent)."""
        if self._native is not None:
            for slot in self._native.get("slots", []):
                if slot.get("busy") is not None:                   # (a step never collected: its kernels still use the workspace)
                    try:
                        slot["busy"].result()
                    except Exception:      # noqa: BLE001
                        pass
                if slot.get("ws") is not None:                     # what the library remembers about this address goes with it
                    self.engine.lib.mia_letkf_step_workspace_release(slot["ws"].data_ptr())
                    slot["ws"] = None
                    slot["key"] = None
                ev = slot.pop("in_event", None)
                if ev is not None and ev.value:
                    self.engine.lib.mia_event_destroy(ev)
            if self._native.get("comm") is not None and not self._native.get("custom"):
                self.engine.lib.mia_comm_destroy(self._native["comm"])
            if self._native.get("part") is not None:
                self.engine.lib.mia_comm_destroy(self._native["part"])
        self._native = None

    def _assimilate_native(self, X, grid_xyz, obs_xyz, Yb, d, G, g0, g1, geometry_id=None):
        """Steady state: ONE library call enqueues the whole step (mia_letkf_sharded_step_f32: records,
        cell index, neighbour lists, analysis chunks, per-chunk RCCL all-gather + placement on a second
        stream, 16-byte max-reduce of the redo counters), then one 32-byte read-back decides -- identically
        on every rank -- whether anything has to be redone.  The first call on a geometry (no bound for the
        local observation count yet) takes the exact-list route through torch.distributed."""
        return self._native_finish(self._native_submit(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1, pipelined=False,
                                                       geometry_id=geometry_id))

    def submit(self, X, grid_xyz, obs_xyz, Yb, d, geometry_id=None) -> "PendingStep":
        """Enqueue one assimilation step WITHOUT waiting for it: the returned handle's ``result()`` performs the
        step's one host read-back (validation + rare redo) and hands out the analysis.  Consecutive steps
        rotate through ``max_in_flight`` slots (own workspace and counters each) and share three streams -- a
        high-priority one for the index / list kernels, one for the analysis kernels, one for the exchange -- so the
        preparation of a later step (a chain of small latency-bound launches) runs beside the analysis kernel of
        an earlier one, and at N > 1 the all-gather of step i (the collectives keep one order on every rank)
        travels while step i+1 is computed.  A slot whose previous step was not collected
        yet is collected first; all ranks must submit and collect in the same order.

        The library reads X, Yb, d and the coordinates on ITS streams after this call has returned: the caller must not
        modify those tensors (in place, on any stream) before ``result()`` of this step -- pass clones if it has to.

        ``geometry_id`` (any hashable, default None): the caller's word that grid and observation COORDINATES are the ones of
        every earlier step submitted with the same id (a fixed observing network in a cycled filter).  Steps on the tile
        route then use the tile lists their pipeline slot already holds and rebuild only the split records
        (MIA_STEP_REUSE_LISTS); results are identical to a full rebuild, which is what the reference does on every call."""
        f = self._fast
        if f is not None and geometry_id is None:
            h = self._submit_fast(f, X, grid_xyz, obs_xyz, Yb, d)
            if h is not None:
                return h
        G = X.shape[-1]
        g0, g1 = block_partition(G, self.world)[self.rank]
        if not (self.native_step and self._compute == self._engine_shard and self._chunk_compute is None
                and torch.is_tensor(X) and X.is_cuda and X.dtype == torch.float32 and X.dim() == 3
                and not self.fused_localization):
            return PendingStep(self, None, out=self.assimilate(X, grid_xyz, obs_xyz, Yb, d))
        return self._native_submit(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1, pipelined=True, geometry_id=geometry_id)

    def _run_fast(self, f, X, grid_xyz, obs_xyz, Yb, d):
        """The steady state of :meth:`assimilate` on one GPU (a cycled filter: one step at a time): the same step as the general
        path -- same launches on torch's current stream, same flags, same validation of the counters -- through ONE library call
        (mia_letkf_step_run_args: step, read-back, wait, counters).  The general path's ~15 us of set-up code ran before the step's
        first launch, i.e. on the step's critical path: 0.080 -> 0.068 ms per step at config 2.  None whenever anything differs from
        the recorded state (the general path then takes the step)."""
        if not _steady_inputs(f, X, grid_xyz, obs_xyz, Yb, d):
            return None
        dev = f["device"]
        if self._native is not f["st"] or self._in_flight:
            return None
        slot = f["slot"]
        hint = self._p_max_hint
        key = (f["G"], f["m"], f["k"], f["P"], f["nc"], int(hint) if hint is not None else -1, 1)
        if hint is None or slot.get("busy") is not None or slot.get("key") != key or slot.get("serial_key") != key:
            return None
        a = f["args"]
        out = torch.empty(f["xs"], dtype=_F32, device=dev)
        step_flags = ((4 if slot.get("ws_clean") else 0) | (8 if self._no_tile_lists else 0) | (self._tile_extra << 4) |
                      (0x400 if self._fresh_box_once else 0) | (0x800 if self._scan_index else 0) |
                      (0 if self._fuse_now(False) else 0x4000))
        self._fresh_box_once = False
        slot["ws_clean"] = False
        cur_raw = torch._C._cuda_getCurrentRawStream(f["dev_index"])
        a.X, a.Yb, a.d, a.grid_xyz, a.obs_xyz, a.Xa = X.data_ptr(), Yb.data_ptr(), d.data_ptr(), grid_xyz.data_ptr(), obs_xyz.data_ptr(), out.data_ptr()
        a.gc_eps, a.inf_factor, a.gamma = float(self.eps), float(self.inf_factor), float(self.rbf_gamma) if self.rbf_gamma is not None else 0.0
        a.method, a.p_max_assumed, a.step_flags = _METHODS[self.method], int(hint), step_flags
        a.stream = a.after_stream = a.on_stream = cur_raw
        if self._time_next:                                        # bench: bracket this step's analysis kernel
            self._time_next = False
            timing = (_cabi.TimingEvent(), _cabi.TimingEvent())
            self.kernel_timings.append(timing)
            a.time_start_event, a.time_stop_event = timing[0].cuda_event, timing[1].cuda_event
        elif a.time_start_event:
            a.time_start_event, a.time_stop_event = None, None
        rc = f["lib"].mia_letkf_step_run_args(f["args_ref"], f["out8"])
        if rc != 0:
            _cabi.check(rc, "mia_letkf_step_run_args")
        G = f["G"]
        h = PendingStep(self, dict(slot=slot, call=None, comp=None, cur=None, cur_raw=cur_raw, dev_index=f["dev_index"], ev=None, job=None,
                                   out=out, flags=slot["flags"], hint=int(hint), last=None, peer=False, timing=None, C_chunks=1,
                                   args=(X, grid_xyz, obs_xyz, Yb, d, G, 0, G), keep=(X, grid_xyz, obs_xyz, Yb, d), geom_key=None,
                                   reused=False, geometry_id=None, cnt=list(f["out8"]), serial_args=a))
        slot["busy"] = h
        self._in_flight.append(h)
        self._submitted += 1
        return self._native_finish(h)

    def _submit_fast(self, f, X, grid_xyz, obs_xyz, Yb, d):
        """The steady state of :meth:`submit` on one GPU: the SAME step as ``_native_submit`` enqueues -- same argument block, same
        streams in the same rotation, same flags -- for inputs that already are what the library reads (device, dtype, contiguous,
        the shapes of the previous step), without the general path's set-up code.  At config 2 the caller's host time per step,
        not the GPU, bounded the pipeline (tools/host_bound.py: submit 25 us + result 12 us = the 37 us period).  Returns None
        whenever anything differs from the state ``_native_submit`` recorded: the general path then takes the step."""
        if not _steady_inputs(f, X, grid_xyz, obs_xyz, Yb, d):
            return None
        dev = f["device"]
        if self._native is not f["st"] or not self.native_step or self.fused_localization:
            return None
        n_sub = self._submitted
        slot = f["slots"][n_sub % f["n"]]
        busy = slot.get("busy")
        if busy is not None:                                      # its previous step was never collected
            busy.result()
        hint = self._p_max_hint
        if hint is None or slot.get("args_key") != (f["G"], f["m"], f["k"], f["P"], f["nc"], int(hint), 1) or self._fast is not f:
            return None
        a = slot["args"]
        out = torch.empty(f["xs"], dtype=_F32, device=dev)
        step_flags = (1 | (4 if slot.get("ws_clean") else 0) | (8 if self._no_tile_lists else 0) | (self._tile_extra << 4) |
                      (0x400 if self._fresh_box_once else 0) | (0x800 if self._scan_index else 0) |
                      (0 if self._fuse_now(True) else 0x4000))
        self._fresh_box_once = False
        slot["ws_clean"] = False
        timing = None
        if self._time_next:
            self._time_next = False
            timing = (_cabi.TimingEvent(), _cabi.TimingEvent())
            self.kernel_timings.append(timing)
            a.time_start_event, a.time_stop_event = timing[0].cuda_event, timing[1].cuda_event
        elif a.time_start_event:
            a.time_start_event, a.time_stop_event = None, None
        comp = f["astreams"][n_sub % f["na"]]
        prep = f["pstreams"][n_sub % f["np"]] if f["np"] else comp
        cur_raw = torch._C._cuda_getCurrentRawStream(f["dev_index"])
        a.X, a.Yb, a.d, a.grid_xyz, a.obs_xyz, a.Xa = X.data_ptr(), Yb.data_ptr(), d.data_ptr(), grid_xyz.data_ptr(), obs_xyz.data_ptr(), out.data_ptr()
        a.gc_eps, a.inf_factor, a.gamma = float(self.eps), float(self.inf_factor), float(self.rbf_gamma) if self.rbf_gamma is not None else 0.0
        a.method, a.p_max_assumed, a.step_flags = _METHODS[self.method], int(hint), step_flags
        a.comm, a.n_chunks = None, 1
        a.stream, a.comm_stream, a.prep_stream = comp.cuda_stream, f["side"], prep.cuda_stream
        a.after_stream, a.on_stream, a.caller_stream = comp.cuda_stream, f["side"], cur_raw
        rc = f["lib"].mia_letkf_step_submit_args(slot["args_ref"], slot["job_ref"])
        if rc != 0:
            _cabi.check(rc, "mia_letkf_step_submit_args")
        G = f["G"]
        h = PendingStep(self, dict(slot=slot, call=None, comp=comp, cur=None, cur_raw=cur_raw, dev_index=f["dev_index"], ev=slot["event"],
                                   job=slot["job"].value, out=out, flags=slot["flags"], hint=int(hint), last=f["last"], peer=False,
                                   timing=timing, C_chunks=1, args=(X, grid_xyz, obs_xyz, Yb, d, G, 0, G),
                                   keep=(X, grid_xyz, obs_xyz, Yb, d), geom_key=None, reused=False, geometry_id=None))
        slot["busy"] = h
        self._in_flight.append(h)
        self._submitted = n_sub + 1
        return h

    def _native_submit(self, X, grid_xyz, obs_xyz, Yb, d, G, g0, g1, pipelined, geometry_id=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _cabi
        from .engine import _ptr
        eng = self.engine
        if self._p_max_hint is None:
            for h in list(self._in_flight):                       # drain: the exact-list route is synchronous
                h.result()
            shard = self._engine_shard(X, grid_xyz, obs_xyz, Yb, d, g0, g1)
            if not self.gather:
                return PendingStep(self, None, out=shard)
            if self.world > 1:
                t = torch.tensor([self._p_max_hint, self._tile_extra, int(self._no_tile_lists)], dtype=torch.int32, device=X.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                # one bound (and one tile format) for all ranks
                self._p_max_hint, self._tile_extra, self._no_tile_lists = int(t[0].item()), int(t[1].item()), bool(t[2].item())
            return PendingStep(self, None, out=gather_blocks(shard, G, self.world, self.group))
        if not self._native_available():
            return PendingStep(self, None, out=self.assimilate(X, grid_xyz, obs_xyz, Yb, d))
        st = self._native_state()
        lib = eng.lib
        # (tensors that already are what the library reads -- device, dtype, contiguous: the steady state of a cycled run --
        #  pass through untouched: the conversions below cost ~15 us of host time per step when they are no-ops)
        def ready(t, dtype):
            return torch.is_tensor(t) and t.dtype is dtype and t.device == X.device and t.is_contiguous()
        if not X.is_contiguous():
            X = X.contiguous()
        m, k = X.shape[0], X.shape[1]
        grid = grid_xyz if ready(grid_xyz, torch.float64) else eng._dev(grid_xyz, torch.float64)
        obs = obs_xyz if ready(obs_xyz, torch.float64) else eng._dev(obs_xyz, torch.float64)
        if grid.dim() == 1:
            grid = grid[:, None]
        if obs.dim() == 1:
            obs = obs[:, None]
        nc = grid.shape[1]
        P = obs.shape[0]
        if not ready(Yb, torch.float32):
            Yb = Yb.to(device=X.device, dtype=torch.float32).contiguous()
        if not (ready(d, torch.float32) and d.dim() == 1):
            d = d.to(device=X.device, dtype=torch.float32).contiguous().reshape(-1)
        if Yb.shape != (k, P) or d.shape[0] != P or grid.shape[0] != G:
            raise ValueError("inconsistent shapes: X (m,k,G), Yb (k,P), d (P,), grid (G,nc), obs (P,nc)")
        hint = int(self._p_max_hint)
        if (st["comm"] is not None and self.world > 1 and "peer" not in st and not st.get("custom")):
            st["peer"] = self._peer_setup(st, m, k, G) if self.peer_exchange != "off" else None
            st["peer_shape"] = (m, k, G)
        peer = st.get("peer") if st.get("peer_shape") == (m, k, G) else None
        C_chunks = 1 if peer else (self.comm_chunks if st["comm"] is not None else 1)
        key = (G, m, k, P, nc, hint, C_chunks)
        slot_idx = self._submitted % len(st["slots"]) if pipelined else 0
        slot = st["slots"][slot_idx]
        if slot.get("busy") is not None:                          # its previous step was never collected
            slot["busy"].result()
        if slot.get("key") != key:
            slot["geom"] = None
            nbytes = C.c_size_t(0)
            _cabi.check(lib.mia_letkf_sharded_step_workspace_bytes(G, m, k, P, nc, self.world, C_chunks, hint,
                                                                   C.byref(nbytes)), "sharded_step_workspace_bytes")
            if slot.get("ws") is None or slot["ws"].numel() < nbytes.value:
                if slot.get("ws") is not None:                     # (the allocator may hand this address to anybody next)
                    lib.mia_letkf_step_workspace_release(slot["ws"].data_ptr())
                slot["ws"] = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=X.device)
            cg = [0] * nc if self.coord_group is None else [int(c) for c in self.coord_group]
            slot["cg"] = (C.c_int32 * nc)(*cg)
            slot["rc"] = (C.c_double * len(self.radii))(*[float(r) for r in self.radii])
            slot["counters"] = torch.zeros(8, dtype=torch.int32, device=X.device)
            slot["host"] = torch.zeros(8, dtype=torch.int32).pin_memory()
            slot["flags"] = torch.empty(max(g1 - g0, 1), dtype=torch.int32, device=X.device)
            slot["event"] = slot.get("event") or C.c_void_p()          # completion event of the read-back (library-made)
            slot["key"] = key
            slot["args"] = None                                        # (the argument block of steps in flight points at the old buffers)
            slot["serial_key"] = None                                  # (... and the one of steps taken one at a time)
            slot["ws_clean"] = False                                   # fresh workspace: the first step clears the index header
        # direct exchange: the result IS the slot's peer-mapped buffer (every rank uses the same slot for the same step)
        part = st.get("part")                       # gather=False: partition-only communicator, the result is this rank's block
        out = peer[slot_idx] if peer else torch.empty((m, k, (g1 - g0) if part is not None else G), dtype=torch.float32, device=X.device)
        flags = slot["flags"]          # (per slot: a step's flags are read when it is collected, before the slot is reused)
        # geometry epoch: this slot's workspace holds the tile lists of an earlier, completed step of the same geometry and format
        geom_key = None
        if geometry_id is not None:
            opts = []
            v = C.c_int(0)
            for name in (b"tile", b"tile_split", b"tile_lists", b"bucket_index"):
                lib.mia_get_option(name, C.byref(v))
                opts.append(v.value)
            geom_key = (geometry_id, key, self._tile_extra, self._scan_index, g0, g1, tuple(opts), self.method, self.rbf_gamma,
                        tuple(float(r) for r in self.radii), float(self.eps),
                        None if self.coord_group is None else tuple(int(c) for c in self.coord_group))
        reuse = (geom_key is not None and slot.get("geom") == geom_key and not self._no_tile_lists and not self._fresh_box_once
                 and C_chunks == 1 and st["comm"] is None)
        method = {"auto": 0, "eig": 1, "matfun": 2}[self.method]
        gamma = float(self.rbf_gamma) if self.rbf_gamma is not None else 0.0
        # (the caller's stream as a raw handle: torch.cuda.current_stream() + Stream.wait_stream() are ~15 us of Python objects
        #  per step; the Stream object is built only on the paths that need one)
        dev_index = X.device.index if X.device.index is not None else torch.cuda.current_device()
        cur_raw = torch._C._cuda_getCurrentRawStream(dev_index)
        cur = None if pipelined else torch.cuda.current_stream(X.device)
        side = st["stream"].cuda_stream if st["stream"] is not None else None
        exch = st["comm"] is not None and (self.world > 1 or C_chunks > 1)
        if pipelined:
            # streams shared by all steps in flight: ONE analysis stream (two analysis kernels sharing the CUs are slower than one
            # after the other: 1.4-1.6e9 against 1.8e9 /s with the tile kernel), `prep_streams` preparation streams taken in turn,
            # the exchange / read-back stream.  All at NORMAL priority: high-priority preparation streams (rounds 1-2, when one
            # chain of small launches had to get past a kernel that filled the chip for 250 us) cost the round-3 loop 10 %
            # (1.80e9 -> 2.00e9 /s with five streams at normal priority; three: 1.8e9, four: 1.95e9, six: 1.84e9)
            if st.get("astream") is None:
                st["astream"] = torch.cuda.Stream(device=X.device)
                st["astreams"] = [st["astream"]] + [torch.cuda.Stream(device=X.device) for _ in range(self.analysis_streams - 1)]
                # (round 3 sorted fresh streams by the hardware queue the runtime had given them -- 1 ms spin-kernel probes at
                #  set-up -- and took a fixed mix; under the driver's flags the plain set measures the same, 1.729e9 against
                #  1.738e9 analyses/s, profiles/r04_stream_ab.txt: the probes are gone.  Fewer streams cost: 2: 1.55e9, 3: 1.69e9)
                st["pstreams"] = [torch.cuda.Stream(device=X.device, priority=self.prep_priority) for _ in range(max(1, self.prep_streams))]
            comp = st["astreams"][self._submitted % len(st["astreams"])] if not exch else st["astream"]
            # (a step on reused lists prepares with ONE short kernel: one preparation stream for all of them -- every further
            #  queue in use costs the analysis queue dispatch time: 0.041 against 0.049 ms per step with three)
            prep = st["pstreams"][0 if reuse else self._submitted % len(st["pstreams"])]
            if self.prep_streams == 0 and not exch:
                # the whole step on its analysis stream: no preparation stream, no event between streams.  With the fused kernel
                # (two launches per step) and analysis_streams=2 the simplest pipeline and the highest rate measured -- 2.11e9
                # analyses/s at config 2 against 1.73e9, at 51 us per analysis launch instead of 35 -- WHEN the two streams get
                # hardware queues of their own: as the third runner of a process the same set-up ran at 1.24e9 (HISTORY.md)
                prep = comp
            # where the step's last work is enqueued: the placement stream when there is one, else the exchange stream
            last = ((st["stream"] if peer else (st.get("xstream") or st["stream"])) if exch else comp)
        else:
            comp, prep, last = cur, None, cur

        step_flags = ((1 if pipelined else 0) | (4 if slot.get("ws_clean") else 0) | (8 if self._no_tile_lists else 0) | (self._tile_extra << 4) |
                      (0x400 if self._fresh_box_once else 0) | (0x800 if self._scan_index else 0) | (0x1000 if reuse else 0) |
                      (0x4000 if (geom_key is not None or not self._fuse_now(pipelined)) else 0) | (0x2000 if part is not None else 0))
        # MIA_STEP_NO_JOIN for steps in flight; MIA_STEP_WS_CLEAN: this slot's workspace was last used by a step that
        # ran to completion (its index kernels leave the header zeroed)
        comm_arg = st["comm"] if part is None else part
        comm_val = getattr(comm_arg, "value", comm_arg)          # (a plain integer / None for the argument block)
        eps, inf = float(self.eps), float(self.inf_factor)

        def call(phase):
            # (plain integers for the pointer arguments: ctypes converts them itself, a C.c_void_p object per argument was a
            #  third of this function's host time)
            rc = lib.mia_letkf_sharded_step_streams_f32(
                X.data_ptr(), G, m, k, Yb.data_ptr(), d.data_ptr(), P, grid.data_ptr(), obs.data_ptr(), nc, slot["cg"],
                slot["rc"], len(self.radii), eps, inf, gamma, method, hint, comm_arg,
                C_chunks, phase, out.data_ptr(), flags.data_ptr(), slot["counters"].data_ptr(), slot["ws"].data_ptr(),
                slot["ws"].numel(), comp.cuda_stream, side, prep.cuda_stream if prep is not None else None, step_flags)
            if rc != 0:
                _cabi.check(rc, "mia_letkf_sharded_step_streams_f32")

        self._fresh_box_once = False
        slot["ws_clean"] = False            # (until this step has been collected without an error)
        timing = None
        if self._time_next:                                        # bench: bracket this step's analysis kernel
            self._time_next = False
            timing = (_cabi.TimingEvent(), _cabi.TimingEvent())    # (pooled events of the library: no torch.cuda.Event, no record)
            self.kernel_timings.append(timing)
        ev = job = None
        if pipelined:
            after = last
            if not exch:
                # read the counters back on the (otherwise idle) exchange stream: a copy enqueued on the analysis
                # stream sits between two analysis kernels and costs ~15 us of dispatch gaps per step
                after, last = comp, st["stream"]
            # The step call and its read-back (wait for `after`, copy the counters to pinned memory on `last`, record the
            # slot's event) run on the library's launch threads, in submission order: their ~65 us of HIP runtime calls
            # overlap this thread's own per-step work instead of adding to it.  The arguments travel in the slot's argument
            # block (mia_step_args_t): what a slot keeps from step to step is written once, and the wait of the preparation
            # stream for the caller's stream (inputs and `out`'s memory are ready) is part of the same call
            a = slot.get("args")
            if a is None or slot.get("args_key") != key:
                a = slot["args"] = _cabi.StepArgs()
                slot["args_key"] = key
                if "in_event" not in slot:
                    slot["in_event"] = C.c_void_p()
                a.G, a.m, a.k, a.P, a.n_coord, a.n_r = G, m, k, P, nc, len(self.radii)
                for i_ in range(nc):
                    a.coord_group[i_] = slot["cg"][i_]
                for i_ in range(len(self.radii)):
                    a.gc_c[i_] = slot["rc"][i_]
                a.flags, a.counters, a.ws, a.ws_bytes = flags.data_ptr(), slot["counters"].data_ptr(), slot["ws"].data_ptr(), slot["ws"].numel()
                a.host8 = slot["host"].data_ptr()
                a.done_event = C.pointer(slot["event"])
                a.in_event = C.pointer(slot["in_event"])
                a.phase = 0
                slot["job"] = C.c_void_p()
                slot["job_ref"] = C.byref(slot["job"])
                slot["args_ref"] = C.byref(a)
            a.X, a.Yb, a.d, a.grid_xyz, a.obs_xyz, a.Xa = X.data_ptr(), Yb.data_ptr(), d.data_ptr(), grid.data_ptr(), obs.data_ptr(), out.data_ptr()
            a.gc_eps, a.inf_factor, a.gamma, a.method, a.p_max_assumed = eps, inf, gamma, method, hint
            a.comm, a.n_chunks, a.step_flags = comm_val, C_chunks, step_flags
            a.stream, a.comm_stream, a.prep_stream = comp.cuda_stream, side, prep.cuda_stream
            a.after_stream, a.on_stream, a.caller_stream = after.cuda_stream, last.cuda_stream, cur_raw
            a.time_start_event, a.time_stop_event = (timing[0].cuda_event, timing[1].cuda_event) if timing else (None, None)
            rc = lib.mia_letkf_step_submit_args(slot["args_ref"], slot["job_ref"])
            if rc != 0:
                _cabi.check(rc, "mia_letkf_step_submit_args")
            job = slot["job"].value
            ev = slot["event"]
        else:
            lib.mia_letkf_step_drain()                             # (a synchronous step must not overtake queued ones)
            if timing:
                _cabi.check(lib.mia_letkf_step_timing_events(timing[0].cuda_event, timing[1].cuda_event),
                            "mia_letkf_step_timing_events")
            call(0)
            # read-back as the steps in flight do it: 32 bytes into pinned memory behind the analysis, one event to wait for
            # (a synchronous Tensor.tolist() of the device counters was ~10 us of this path's host time per step)
            raw = comp.cuda_stream if comp is not None else cur_raw
            if lib.mia_letkf_step_readback(slot["counters"].data_ptr(), slot["host"].data_ptr(), raw, raw, C.byref(slot["event"])) == 0:
                ev = slot["event"]
        h = PendingStep(self, dict(slot=slot, call=call, comp=comp, cur=cur, cur_raw=cur_raw, dev_index=dev_index, ev=ev, job=job, out=out, flags=flags, hint=hint,
                                   last=last, peer=bool(peer), timing=timing,
                                   C_chunks=C_chunks, args=(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1),
                                   keep=(X, grid, obs, Yb, d), geom_key=geom_key, reused=reuse, geometry_id=geometry_id))
        slot["busy"] = h
        self._in_flight.append(h)
        self._submitted += 1
        # the steady state for _run_fast: the same, one step at a time (slot 0, torch's current stream)
        if (not pipelined and st["comm"] is None and part is None and not peer and C_chunks == 1 and self.world == 1 and not exch
                and geometry_id is None and g0 == 0 and g1 == G and ev is not None):
            f = self._fast_serial
            if (f is None or f["st"] is not st or f["slot"] is not slot or slot.get("serial_key") != key or f["device"] != X.device):
                a = _cabi.StepArgs()
                a.G, a.m, a.k, a.P, a.n_coord, a.n_r = G, m, k, P, nc, len(self.radii)
                for i_ in range(nc):
                    a.coord_group[i_] = slot["cg"][i_]
                for i_ in range(len(self.radii)):
                    a.gc_c[i_] = slot["rc"][i_]
                a.flags, a.counters, a.ws, a.ws_bytes = flags.data_ptr(), slot["counters"].data_ptr(), slot["ws"].data_ptr(), slot["ws"].numel()
                a.host8 = slot["host"].data_ptr()
                a.done_event = C.pointer(slot["event"])
                a.phase, a.n_chunks, a.comm = 0, 1, None
                a.comm_stream = side
                slot["serial_key"] = key
                self._fast_serial = dict(st=st, lib=lib, slot=slot, args=a, args_ref=C.byref(a), out8=(C.c_int32 * 8)(), G=G, m=m, k=k, P=P,
                                         nc=nc, device=X.device, xs=X.shape, ys=Yb.shape, ds=d.shape,
                                         gs=grid_xyz.shape if torch.is_tensor(grid_xyz) else None,
                                         os=obs_xyz.shape if torch.is_tensor(obs_xyz) else None, dev_index=dev_index)
        elif not pipelined:
            self._fast_serial = None
        # the steady state for _submit_fast: one GPU, no exchange, no geometry epoch, steps in flight
        if (pipelined and st["comm"] is None and part is None and not peer and C_chunks == 1 and self.world == 1 and not exch
                and geometry_id is None and g0 == 0 and g1 == G):
            f = self._fast
            if f is None or f["st"] is not st or (f["G"], f["m"], f["k"], f["P"], f["nc"]) != (G, m, k, P, nc) or f["device"] != X.device:
                self._fast = dict(st=st, lib=lib, slots=st["slots"], n=len(st["slots"]), G=G, m=m, k=k, P=P, nc=nc, device=X.device,
                                  xs=X.shape, ys=Yb.shape, ds=d.shape, gs=grid_xyz.shape if torch.is_tensor(grid_xyz) else None,
                                  os=obs_xyz.shape if torch.is_tensor(obs_xyz) else None, dev_index=dev_index,
                                  astreams=st["astreams"], na=len(st["astreams"]), pstreams=st["pstreams"],
                                  np=len(st["pstreams"]) if self.prep_streams else 0, side=st["stream"].cuda_stream, last=st["stream"],
                                  )
        else:
            self._fast = None
        return h

    def _call_from_args(self, a):
        """The step call of a step submitted through an argument block (mia_step_args_t), for its rare second phase."""
        from . import _cabi
        lib = self.engine.lib

        def call(phase):
            rc = lib.mia_letkf_sharded_step_streams_f32(a.X, a.G, a.m, a.k, a.Yb, a.d, a.P, a.grid_xyz, a.obs_xyz, a.n_coord, a.coord_group,
                                                        a.gc_c, a.n_r, a.gc_eps, a.inf_factor, a.gamma, a.method, a.p_max_assumed, a.comm,
                                                        a.n_chunks, phase, a.Xa, a.flags, a.counters, a.ws, a.ws_bytes, a.stream,
                                                        a.comm_stream, a.prep_stream, a.step_flags)
            if rc != 0:
                _cabi.check(rc, "mia_letkf_sharded_step_streams_f32")
        return call

    def _native_finish(self, h: "PendingStep"):
        if h._out is not None or h._st is None:
            return h._out
        p = h._st
        slot, st = p["slot"], self._native_state()
        if p.get("cnt") is not None:
            cnt = p["cnt"]                                     # (the step's one library call has waited and read them back)
        elif p["ev"] is not None:
            if p["job"] is not None:
                # ONE call: the launch thread has enqueued this step (join), the host waits for its read-back event -- the one host
                # wait for the GPU --, the counters come back, and torch's current stream waits for that event too (consumers see
                # the result; THIS step's event only: the analysis stream as a whole also holds the later steps)
                import ctypes as C
                if "out8" not in slot:
                    slot["out8"], slot["bn"] = (C.c_int32 * 8)(), C.c_int(1)
                    slot["bn_ref"] = C.byref(slot["bn"])
                rc = self.engine.lib.mia_letkf_step_collect(p["job"], C.byref(p["ev"]), slot["host"].data_ptr(),
                                                            torch._C._cuda_getCurrentRawStream(p["dev_index"]), 1, slot["out8"], slot["bn_ref"])
                if rc != 0:
                    _cabi.check(rc, "mia_letkf_sharded_step_streams_f32 (launch thread)")
                self.last_batch_n = h.batch_n = slot["bn"].value
                if p["timing"] is not None:
                    self.kernel_batch[p["timing"]] = h.batch_n
                cnt = list(slot["out8"])
                p["waited"] = True
            else:
                self.engine.lib.mia_event_synchronize(p["ev"])     # ... and this is the one host wait for the GPU
                cnt = slot["host"].tolist()
        else:
            cnt = slot["counters"].tolist()                    # serial route: synchronous read-back
        slot["busy"] = None
        slot["ws_clean"] = True                                # the step's kernels have all run: its index header is zero again
        slot["geom"] = None                                    # (set again below once this step is known to be good)
        self._in_flight.remove(h)
        X, grid_xyz, obs_xyz, Yb, d, G, g0, g1 = p["args"]
        if st["comm"] is None or (self.world == 1 and p["C_chunks"] == 1):
            cnt[4:8] = cnt[0:4]                                # no exchange route: the rank's own counters
        # MIA_STEP_STATUS_SAMPLED: the step ran on the fused kernel, whose "longest list" is exact only when it EXCEEDS the bound
        # the step was sized for (then the step is redone below); otherwise it is the maximum over one tile in 64
        sampled = bool(cnt[7] & STATUS_SAMPLED)
        nonfinite = bool(cnt[7] & STATUS_NONFINITE)
        cnt[7] &= ~(STATUS_SAMPLED | STATUS_NONFINITE)
        p_seen, n_over, n_retry = cnt[4], cnt[5], cnt[6]
        redo = None
        if cnt[7] & 2 and p.get("peer"):
            # direct exchange: a waiter of this step gave up -- a peer is LATE.  Its push does not depend on this rank, so nothing is
            # redone: this rank waits AGAIN for the flags of that exchange (a late peer used to be a hard error),
            # up to peer_rewaits times; only a peer that stays silent through all of them is called dead.  Nothing collective: the
            # other ranks need not know, and the counters folded after the wait are the ones they decided on.
            import warnings
            slot_idx = st["slots"].index(slot)
            for attempt in range(self.peer_rewaits):
                warnings.warn("direct exchange: a peer had not delivered within the waiter's bound; waiting again (%d of %d)"
                              % (attempt + 1, self.peer_rewaits), RuntimeWarning)
                if self._peer_rewait_hook is not None:
                    self._peer_rewait_hook(attempt)
                stream = p["last"]
                _cabi.check(self.engine.lib.mia_comm_peer_rewait(st["comm"], slot_idx, slot["counters"].data_ptr(), stream.cuda_stream),
                            "mia_comm_peer_rewait")
                stream.synchronize()
                cnt = slot["counters"].tolist()
                if not (cnt[7] & 2):
                    break
            sampled = sampled or bool(cnt[7] & STATUS_SAMPLED)
            nonfinite = nonfinite or bool(cnt[7] & STATUS_NONFINITE)
            cnt[7] &= ~(STATUS_SAMPLED | STATUS_NONFINITE)
            p_seen, n_over, n_retry = cnt[4], cnt[5], cnt[6]
        if cnt[7] & 24:
            # bucket index of the tile route: an observation outside the bounding box this slot's workspace held (or other
            # radii) -> this step again with a fresh box; a cell with more observations than a bucket holds -> the scan-based
            # index from now on.  Identical on every rank (the error bits are OR-ed over the ranks).
            if cnt[7] & 16:
                self._scan_index = True
            else:
                self._fresh_box_once = True
            slot["ws_clean"] = False                           # (this slot's box is stale: its next step rebuilds it)
            redo = "same"
        elif cnt[7] & 2:
            raise _cabi.MiaError("direct exchange: a peer did not deliver within %d waits of the waiter's bound (error bits %d)"
                                 % (self.peer_rewaits + 1, cnt[7]))
        elif cnt[7]:
            # a segment waiter gave up (the analysis launch and the exchange stream must be able to run
            # concurrently: e.g. more HIP streams than hardware queues): all ranks switch to one launch + one
            # event per piece and repeat the step
            import warnings
            if _cabi.set_option("segment_signal", 0) == 0:
                raise _cabi.MiaError("native step driver: exchange error bits %d" % cnt[7])
            warnings.warn("segmented launch timed out waiting for a segment; falling back to per-piece launches",
                          RuntimeWarning)
            redo = "same"
        elif n_over & TILE_BOX_OVERFLOW and not self._no_tile_lists:
            self._no_tile_lists = True                         # a tile's cell box is too large for tile lists: per-point lists
            redo = "same"
        elif n_over and p_seen <= p["hint"] and not self._no_tile_lists:
            # tile route: the union of some tile's lists does not fit its slots (the bound on the lists themselves held): sixteen
            # more slots per tile, and once the format has no more to give this geometry goes back to per-point lists -- on
            # every rank alike (the counters are the maximum over the ranks)
            k_ = X.shape[1]
            most = min((k_ + 15) // 16 + 1, 6) - max(1, (int(p["hint"]) + 8 + 15) // 16)      # what the workspace was sized for
            if self._tile_extra < min(5, most):
                self._tile_extra += 1
            else:
                self._no_tile_lists = True
            redo = "same"
        elif n_over or p_seen > p["hint"]:
            self._p_max_hint = None                            # bound broken on some rank: all ranks redo
            redo = "exact"
        if redo:
            for other in list(self._in_flight):                # steps enqueued behind this one used the same bound
                other.result()
            if redo == "exact":
                self._p_max_hint = None                        # (draining may have set a hint again: exact lists now)
            cur_s = p["cur"] if p["cur"] is not None else torch.cuda.ExternalStream(p["cur_raw"], device=X.device)
            if p["comp"] is not None:                         # (None: the step ran on torch's stream itself, _run_fast)
                cur_s.wait_stream(p["comp"])
            if p["last"] is not None:
                cur_s.wait_stream(p["last"])
            h._out = self._assimilate_native(X, grid_xyz, obs_xyz, Yb, d, G, g0, g1)
            h._st = None
            return h._out
        if self._last_kernel is None or self.native_steps % 64 == 0:
            self._note_kernel()                                # (a ctypes call: not on every step of a timed loop)
        if n_retry:
            self.engine.lib.mia_letkf_step_drain()             # (after the steps already handed to the launch thread)
            (p["call"] or self._call_from_args(p.get("serial_args") or slot["args"]))(1)      # eigensolver redoes declined points; re-exchange
        if p["job"] is not None or p["comp"] is not p["cur"]:     # (a step taken one at a time ran on torch's stream itself)
            # consumers on torch's stream see the result.  Wait for THIS step's completion event only: waiting for
            # the analysis stream as a whole would also wait for the later steps already enqueued on it, and the
            # next submit's preparation (which waits for torch's stream) would serialise behind them
            if n_retry:
                now = torch.cuda.current_stream(X.device)
                for strm in {p["comp"], p["last"]}:
                    e2 = torch.cuda.Event()
                    e2.record(strm)
                    now.wait_event(e2)
            elif not p.get("waited"):
                self.engine.lib.mia_stream_wait_event(torch._C._cuda_getCurrentRawStream(p["dev_index"]), p["ev"])
        self.native_steps += 1
        # what the per-point flags of this step can hold, known WITHOUT reading them: the fused kernel (sampled bit) reports non-finite
        # points in the status word, overflowing unions redo the step above, and only a float64 redo of declined points can add the
        # eigensolver's own bits -- None: unknown, scan the flags (other kernels, a redo ran)
        self.last_flags_summary = (4 if nonfinite else 0) if (sampled and not n_retry) else None
        self.last_retries = cnt[2]
        self.reused_steps += 1 if p.get("reused") else 0
        if sampled:
            # a sampled maximum never RAISES the bound (a longer list would have broken it: redo above) and lowers it only across
            # a boundary of the tile format -- sixteen slots -- with a margin of four for what the sample may have missed
            ut_of = lambda q: max(1, (int(q) + 8 + 15) // 16)      # noqa: E731  (tile_ut_for, csrc/mia_tiles.h)
            if ut_of(p_seen + 4) < ut_of(p["hint"]):
                self._p_max_hint = p_seen + 4
        elif not p.get("reused"):                              # (a step on reused lists reports no list lengths)
            self._p_max_hint = p_seen if self.world > 1 else max(p_seen, 0)
        if not self._no_tile_lists:
            slot["geom"] = p.get("geom_key")                   # this slot's lists now belong to that geometry epoch
        self.last_p_max = p["hint"]
        self._last_flags_lazy = (p["flags"], g1 - g0)       # (sliced when somebody asks: a tensor view per step is ~2 us)
        h._out, h._st = (p["out"].clone() if p.get("peer") and self.copy_results else p["out"]), None
        return h._out

    # ------------------------------------------------------------------ compute / exchange overlap
    def _chunk_engine(self, X, grid_xyz, obs_xyz, Yb, d, c0, c1, state, buf):
        """Analysis of sub-range [c0, c1) of this rank's block into ``buf[:, :, :c1-c0]``.  The shard-wide
        preparation (packed records, neighbour lists of the whole block) is enqueued with the first chunk
        and kept in ``state``.  state["vec"] = device int32 [max list length, #truncated lists, #declined
        points]: written by the kernels, never read here.  Returns the deferred retry launcher."""
        eng = self.engine
        b0, b1 = state["block"]
        vec = state["vec"]
        if "nb" not in state:
            state["rec"] = eng.pack_obs(Yb, d, X.dtype)
            nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, b0, b1,
                              assume_p_max=self._p_max_hint, stats_out=vec[:2])
            if nb.stats is None:            # exact lists (first call on a geometry): publish their maximum
                vec[:2].copy_(torch.tensor([nb.p_max, 0], dtype=torch.int32), non_blocking=True)
            state["nb"] = nb
            state["flags"] = torch.empty(b1 - b0, dtype=torch.int32, device=X.device)
        nb = state["nb"]
        from .engine import NeighbourLists
        sub = NeighbourLists(nb.cnt[c0 - b0:c1 - b0], nb.idx[c0 - b0:c1 - b0], nb.w[c0 - b0:c1 - b0],
                             nb.p_cap, nb.p_max, c0, c1)
        _, finish = eng.analysis(X, None, None, sub, self.inf_factor, rbf_gamma=self.rbf_gamma, rec=state["rec"],
                                 out=buf, out_offset=0, method=self.method, defer_retry=True,
                                 retry=vec[2:3], flags=state["flags"][c0 - b0:c1 - b0])
        return finish

    def _overlap_buffers(self, X, world, C, n, nc):
        m, k = X.shape[0], X.shape[1]
        key = (X.dtype, X.device, m, k, world, C, n, nc)
        if self._obuf is None or self._obuf["key"] != key:
            self._obuf = dict(key=key,
                              gath=torch.empty((C, world * m, k, nc), dtype=X.dtype, device=X.device),
                              bufs=[torch.zeros((m, k, nc), dtype=X.dtype, device=X.device) for _ in range(C)],
                              vec=torch.zeros(3, dtype=torch.int32, device=X.device))
        return self._obuf

    def _assimilate_overlapped(self, X, grid_xyz, obs_xyz, Yb, d, G, g0, g1):
        """The rank's block is analysed in ``comm_chunks`` pieces; the all-gather of piece c (RCCL) and its
        copy into the (m, k, G) result run on a side stream while piece c+1 is being analysed, so at 8 GPUs
        the 16 MB-per-rank exchange hides behind the compute instead of adding to it.

        Nothing is read back by the host until everything is enqueued.  The three counters that can demand a
        redo (longest neighbour list vs the assumed bound, truncated lists, grid points the matfun kernel
        declined) stay on the device, are max-reduced over the ranks with one 12-byte all-reduce and read
        once: every rank takes the same decision, so the rare second exchange cannot dead-lock."""
        import torch.distributed as dist
        world, C = self.world, self.comm_chunks
        n = (G + world - 1) // world                 # common block length
        nc = (n + C - 1) // C                        # common chunk length
        m, k = X.shape[0], X.shape[1]
        cuda = X.is_cuda
        chunk_fn = self._chunk_compute or self._chunk_engine
        ob = self._overlap_buffers(X, world, C, n, nc)
        gath, bufs, vec = ob["gath"], ob["bufs"], ob["vec"]
        out = torch.empty((m, k, world, C * nc), dtype=X.dtype, device=X.device)
        hint = self._p_max_hint
        state = {"block": (g0, g1), "vec": vec}
        if cuda:
            comp = torch.cuda.current_stream(X.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=X.device)
            comm = self._comm_stream
            comm.wait_stream(comp)                   # the buffers' previous readers are done
        vec.zero_()

        def exchange(c):
            dist.all_gather_into_tensor(gath[c], bufs[c], group=self.group)
            out[:, :, :, c * nc:(c + 1) * nc].copy_(gath[c].view(world, m, k, nc).permute(1, 2, 0, 3))

        def exchange_all():
            for c in range(C):
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record(comp)
                    with torch.cuda.stream(comm):
                        comm.wait_event(ev)
                        exchange(c)
                else:
                    exchange(c)

        finishes = []
        for c in range(C):
            c0, c1 = min(g1, g0 + c * nc), min(g1, g0 + (c + 1) * nc)
            if c1 > c0:
                finishes.append(chunk_fn(X, grid_xyz, obs_xyz, Yb, d, c0, c1, state, bufs[c]))
            if cuda:
                ev = torch.cuda.Event()
                ev.record(comp)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    exchange(c)
            else:
                exchange(c)
        if cuda:
            with torch.cuda.stream(comm):
                red = vec.clone()
                dist.all_reduce(red, op=dist.ReduceOp.MAX, group=self.group)
            comp.wait_stream(comm)
        else:
            red = vec.clone()
            dist.all_reduce(red, op=dist.ReduceOp.MAX, group=self.group)
        p_seen, n_over, n_retry = (int(v) for v in red.tolist())      # the one host sync of the step
        self.last_retries = 0
        if n_over or (hint is not None and p_seen > hint):
            # some rank's assumed list bound did not hold: all ranks redo the step with exact lists
            self._p_max_hint = None
            shard = self._compute(X, grid_xyz, obs_xyz, Yb, d, g0, g1)
            return gather_blocks(shard, G, world, self.group)
        if n_retry:
            # some rank's matfun kernel declined grid points: the eigensolver redoes them in place
            # (no-op launches elsewhere) and all ranks exchange again
            self.last_retries = max([f() for f in finishes if f is not None] + [0])   # (one shared counter)
            exchange_all()
            if cuda:
                comp.wait_stream(comm)
        if self._chunk_compute is None:
            self._p_max_hint = p_seen
            self.last_p_max = state["nb"].p_max if "nb" in state else p_seen
            self._last_flags = state.get("flags")
        out = out.view(m, k, world, C * nc)
        if C * nc != n:                               # chunk padding inside each rank's block
            out = out[:, :, :, :n]
        out = out.reshape(m, k, world * n)
        return out if world * n == G else out[:, :, :G].contiguous()

    def time_exchange(self, m: int, k: int, G: int, reps: int = 10):
        """The exchange of one step ALONE (no analysis in front of it), HIP events on the exchange stream, mean over ``reps``:
        the direct peer exchange when that route is up (mia_comm_peer_exchange on slot 0: free flags, push of this rank's block to
        every peer, ready flags, waits), otherwise one all-gather of the blocks through torch.distributed (RCCL).  Collective:
        every rank calls it at the same point.  Returns (ms, route, bytes this rank sends per step)."""
        import ctypes as C
        import torch.distributed as dist
        from . import _cabi
        g0, g1 = block_partition(G, self.world)[self.rank]
        n = (G + self.world - 1) // self.world
        st = self._native if self._native is not None else {}
        peer = st.get("peer") if st.get("peer_shape") == (m, k, G) else None
        stream = st.get("stream") or torch.cuda.current_stream(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if peer:
            ctr = torch.zeros(8, dtype=torch.int32, device=self.device)
            lib = self.engine.lib
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
            with torch.cuda.stream(stream):
                e0.record()
                for _ in range(reps):
                    _cabi.check(lib.mia_comm_peer_exchange(st["comm"], 0, m * k, G, g0, g1, ctr.data_ptr(), C.c_void_p(stream.cuda_stream)),
                                "mia_comm_peer_exchange")
                e1.record()
            torch.cuda.synchronize(self.device)
            return e0.elapsed_time(e1) / reps, "direct", m * k * (g1 - g0) * 4 * (self.world - 1)
        send = torch.zeros((m, k, n), dtype=torch.float32, device=self.device)
        recv = torch.empty((self.world, m, k, n), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        e0.record()
        for _ in range(reps):
            dist.all_gather_into_tensor(recv, send, group=self.group)
        e1.record()
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) / reps, "rccl", m * k * n * 4 * (self.world - 1)

    def time_next_step(self):
        """Arm the library's profiling hook for the next native step (mia_letkf_step_timing_events): its analysis
        kernel is bracketed by two events on the stream it runs on; collect with :meth:`kernel_ms`."""
        self._time_next = True

    def kernel_ms(self):
        """Mean duration (ms) of the analysis LAUNCHES timed so far inside real steps, or None.  With launch coalescing
        (option ``step_coalesce``) a launch holds the tiles of :meth:`kernel_steps_per_launch` steps."""
        if not self.kernel_timings:
            return None
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self.kernel_timings) / len(self.kernel_timings)

    def kernel_steps_per_launch(self):
        """Mean number of steps in the launches :meth:`kernel_ms` averages over (1.0 without coalescing)."""
        if not self.kernel_timings:
            return 1.0
        return sum(self.kernel_batch.get(t, 1) for t in self.kernel_timings) / len(self.kernel_timings)

    def mean_degree(self):
        """Mean Chebyshev degree of the last matfun launch (flags bits 8-15), None for the eigensolver route."""
        if self._last_flags is None or self.method == "eig":
            return None
        return float(((self._last_flags >> 8) & 0xff).float().mean().item())

    def mean_tile_degree(self):
        """Mean over the tiles of sixteen points of the LARGEST degree in the tile (a tile's wavefront runs to the maximum of
        its points), None for the eigensolver route."""
        if self._last_flags is None or self.method == "eig":
            return None
        dg = ((self._last_flags >> 8) & 0xff).float()
        n = dg.numel() // 16 * 16
        if n == 0:
            return float(dg.max().item())
        return float(dg[:n].view(-1, 16).max(dim=1).values.mean().item())

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
        tiles_route = (not fused and not self._no_tile_lists and len(obs_xyz) > 0 and self._p_max_hint is not None
                       and eng.tile_route_applies(X, self._p_max_hint, self._tile_extra, self.rbf_gamma, self.method,
                                                  P=int(Yb.shape[1]), n_points=g1 - g0))
        rbf_tiles = tiles_route and self.rbf_gamma is not None
        names = ["(no records: the kernel reads Yb, d)" if rbf_tiles else ("pack_split" if tiles_route else "pack_obs"),
                 "obs_index_build" if fused else ("localize_tiles entry (scan index: 5 kernels, + tile lists; the step driver bins with ONE bucket kernel instead)" if tiles_route
                                                  else "localize(index+lists, incl. host sync)"), "analysis_kernel"]
        acc = dict.fromkeys(names, 0.0)
        burst = 5
        for _ in range(reps):
            nk = 1
            e = [ev() for _ in range(4)]
            e[0].record()
            rec = None if rbf_tiles else (eng.pack_split(Yb, d) if tiles_route else eng.pack_obs(Yb, d, X.dtype))
            e[1].record()
            if fused:
                index = eng.build_index(obs_xyz, self.radii, self.coord_group)
                e[2].record()
                _, _, fin = eng.analysis_fused(X, rec, grid_xyz, index, self._p_max_hint, self.inf_factor, self.eps,
                                               self.rbf_gamma, g0, g1)
            elif tiles_route:
                tiles = eng.localize_tiles(grid_xyz, obs_xyz, self.radii, self._p_max_hint, self.coord_group, self.eps, g0, g1,
                                           extra_blocks=self._tile_extra)
                torch.cuda.synchronize()          # (idle GPU in front of the burst, as on the list route)
                out = torch.empty((X.shape[0], X.shape[1], g1 - g0), dtype=torch.float32, device=X.device)
                e[2].record()
                for _ in range(burst):
                    if rbf_tiles:
                        eng.analysis_tiles_rbf(X, Yb, d, tiles, self.inf_factor, self.rbf_gamma, out=out)
                    else:
                        eng.analysis_tiles(X, rec, Yb.shape[1], tiles, self.inf_factor, out=out)
                nk = burst
                fin = lambda: 0
            else:
                nb = eng.localize(grid_xyz, obs_xyz, self.radii, self.coord_group, self.eps, g0, g1)
                # the list step ends in a host read-back, so the GPU is idle here: launch the analysis kernel
                # `burst` times back to back and divide, otherwise the interval between the two events measures
                # the host's launch latency (~40 us from Python) on top of the kernel
                e[2].record()
                for _ in range(burst):
                    _, fin = eng.analysis(X, None, None, nb, self.inf_factor, rbf_gamma=self.rbf_gamma, rec=rec,
                                          method=self.method, defer_retry=True)
                nk = burst
            e[3].record()
            fin()
            torch.cuda.synchronize()
            for name, a, b in zip(names, e[:-1], e[1:]):
                acc[name] += a.elapsed_time(b) / (nk if name == "analysis_kernel" else 1)
        stage = {k: v / reps for k, v in acc.items()}
        return stage["analysis_kernel"], stage

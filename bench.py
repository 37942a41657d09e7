#!/usr/bin/env python3
"""Headline benchmark: local analyses / second of the LETKF hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload at N = 1 is BASELINE config 2: LETKF, G = 1e5 grid points, k = 40 members,
observations at every 2nd grid point (P = 5e4), Gaspari-Cohn radius 10 (19-20 local obs),
inflation 1.1, float32, synthetic N(0,1) state/obs (recipe of the reference's
examples/benchmark_letkf.py, RandomState(42)).  For N > 1 every rank owns a block of 1e5
grid points of a G = N * 1e5 problem (config 3 at N = 8, weak scaling) and the analysis
ensemble is all-gathered over RCCL.

One step = one full pass of the hot path with inputs resident in HBM: pack obs records,
build the observation cell index, tile-shaped Gaspari-Cohn lists, fused local analysis
(Gram, matrix functions, transform) [, all-gather], enqueued by ONE call
into the C-ABI library (mia_letkf_sharded_step_f32) and ended by the one host read-back that
validates it.  Prints ONE JSON line on rank 0.
"""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_ENS = 40
OBS_STRIDE = 2
GC_RADIUS = 10.0
INF = 1.1
G_PER_GPU = 100000
PEAK_FP32_TFLOPS = 157.3     # MI355X FP32 vector = FP32 MFMA peak (MI355X_MICROARCH.md)
PEAK_F16_TFLOPS = 2516.8      # dense F16 / BF16 MFMA: 16 x the F32 rate (same guide)
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(k, p, m):
    """k^2 p [Gram] + 2kp [Yb d] + 9k^3 [sym-eig, Golub-Van Loan count] + k^3 [W]
    + 5k^2 [w_mean] + 2k^2 m [transform] + 25p [GC]."""
    return k * k * p + 2 * k * p + 9 * k ** 3 + k ** 3 + 5 * k * k + 2 * k * k * m + 25 * p


def executed_flops(k, p, m, deg, kernel="tile"):
    """What the dominant kernel really executes per analysis (dual route), padding included.
    tile (letkf_tile_kernel, 16 grid points per wavefront, csrc/letkf_tile.hip): v_mfma_f32_16x16x4_f32 instructions per
    tile -- Gram UT^2 4KT + Z UT 4KT, Gershgorin UT nks, recurrence deg UT nks, x'w_mean nks, output KT nks, with UT row
    blocks of 16 union slots, KT member blocks, nks = ceil(U / 4) steps over the tile's union U ~ p + 8 -- times 2048 flop,
    plus the three-term update on the vector unit (10 flop per union slot, step and point); per point = / 16.
    point (letkf_cheb_kernel): Gram on 16x16x4 tiles over the padded order, z / output products, recurrence."""
    if not deg:
        return None
    if kernel == "tile2":
        # letkf_tile2_kernel (tile lists + split records): THREE v_mfma_f32_16x16x32_f16 (hi hi, hi lo, lo hi; 16384 flop each)
        # per f32 product block of 16 x 16 x 32 -- Gram UT^2 NB + per state row Z UT NB, deg products UT NKB (the tile runs to the
        # largest degree of its 16 points), output KT NKB -- and one for the Gershgorin product (UT NKB).  Returned: the f32
        # products these implement, padding included, + the vector-unit update (4 fused multiply-adds per union slot, step and
        # point), per analysis; second value = what the matrix cores execute
        ut, kt = (p + 8 + 15) // 16, (k + 15) // 16
        nb, nkb = (kt + 1) // 2, (ut + 1) // 2
        dmax = int(deg + 0.999)
        triples = ut * ut * nb + m * (ut * nb + dmax * ut * nkb + kt * nkb)
        singles = ut * nkb
        valu = m * dmax * 16 * ut * 16 * 8
        return ((triples + singles) * 16384 + valu) / 16.0, (3 * triples + singles) * 16384 / 16.0
    if kernel == "tile_split":
        # split-precision instantiations: every product is THREE v_mfma_f32_16x16x32_f16 (hi hi, hi lo, lo hi; 16384 flop
        # each) per pair of 16-row blocks and 32 summation indices; the Gershgorin product takes one.  Returned: the f32
        # products these implement (one per triple), padding included, + the vector-unit update -- `f16` (second value)
        # is what the matrix cores execute
        ut, kt, u = (p + 8 + 15) // 16, (k + 15) // 16, p + 8
        nb, nkb = (kt + 1) // 2, (u + 31) // 32
        dmax = int(deg + 0.999)
        triples = ut * ut * nb + m * (ut * nb + dmax * ut * nkb + nkb + kt * nkb)
        singles = ut * nkb
        valu = m * dmax * 16 * ut * 16 * 10
        return ((triples + singles) * 16384 + valu) / 16.0, (3 * triples + singles) * 16384 / 16.0
    if kernel == "tile":
        ut, kt, u = (p + 8 + 15) // 16, (k + 15) // 16, p + 8
        nks = (u + 3) // 4
        dmax = int(deg + 0.999)                      # a tile runs to the largest degree of its 16 points
        mfma = ut * ut * 4 * kt + m * (ut * 4 * kt + dmax * ut * nks + nks + kt * nks) + ut * nks
        return (mfma * 2048 + m * dmax * 16 * ut * 16 * 10) / 16.0
    tt = (max(p, 1) + 15) // 16
    ks = (k + 3) // 4
    gram = tt * (tt + 1) // 2 * ks * 2048
    n = (p + 3) // 4 * 4
    return gram + m * (2 * k * n * 2 + deg * 2 * n * n) + 4 * (deg + 1) ** 2 + 25 * p


def useful_flops(k, p, m, deg):
    """The same analysis without padding or sharing: symmetric Gram 2 k p(p+1)/2, per state row z and output 4 k p and a
    degree-deg recurrence 2 deg p^2, Gaspari-Cohn 25 p ('useful work')."""
    if not deg:
        return None
    return k * p * (p + 1) + m * (4 * k * p + 2 * deg * p * p) + 25 * p


def useful_flops_shared(k, p, m, deg, u=None):
    """As useful_flops, but with the Gram matrix counted ONCE per tile of sixteen points over the tile's union of U ~ p + 8
    observations (what a tile-shaped algorithm needs at least): k U (U + 1) / 16 per analysis.  This count is a subset of what
    the tile kernels execute, so its rate over the FP32 peak cannot exceed the executed one."""
    if not deg:
        return None
    u = p + 8 if u is None else u
    return k * u * (u + 1) / 16.0 + m * (4 * k * p + 2 * deg * p * p) + 25 * p


def algorithmic_bytes(k, m, P_over_G, n_coord=1):
    return 4.0 * (2 * k * m + (k + 1) * P_over_G + 2 * n_coord * (1 + P_over_G))


def _cabi_stats():
    from torch_assimilate_amd import _cabi
    return _cabi.step_coalesce_stats()


def kernel_base_name(name):
    """'void mia::letkf_tile2f_kernel<2, 3, 1, false, 4>(mia::Tile2FParams)' -> 'letkf_tile2f_kernel<2, 3, 1, false, 4>'."""
    if not name:
        return ""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    if name.startswith("mia::"):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def profile_summary(world, launched=None, k=K_ENS, grid=G_PER_GPU):
    """The committed profile of the dominant kernel on this workload (profiles/latest_c2.json: the summary tools/prof_tile.sh writes
    from one rocprofv3 --kernel-trace --stats pass of bench.py and its --pmc passes, FETCH_SIZE / WRITE_SIZE in passes of their own).
    bench.py cannot run the profiler on itself: what the line says about counters -- bound evidence, HBM traffic, unit
    utilisations -- is READ from this one file, so a refreshed profile changes the line without a code edit.  None when the
    file is absent, was taken on another workload, or is a profile of ANOTHER KERNEL than the one this run launched (`launched`:
    mia_last_analysis_kernel, template arguments included): counter figures of one kernel never sit beside another's times."""
    path = os.path.join(ROOT, "profiles", "latest_c2.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("grid_points") != grid or rec.get("k", K_ENS) != k:
        return None
    if launched is not None and kernel_base_name(rec.get("kernel_name_recorded")) != kernel_base_name(launched):
        return None
    return rec


def bound_evidence(rec):
    """One sentence from the profile summary's numbers."""
    if not rec:
        return None
    dv = rec.get("derived", {})
    parts = ["profiles/latest_c2.json (rocprofv3 --pmc, kernel alone, %s)" % rec.get("source", "tools/prof_tile.sh")]
    if "valu_busy_frac" in dv:
        parts.append("vector unit %.0f %% busy" % (100 * dv["valu_busy_frac"]))
    if "mfma_busy_frac" in dv:
        parts.append("matrix pipe %.0f %%" % (100 * dv["mfma_busy_frac"]))
    if "hbm_bytes_raw" in dv and "kernel_cycles" in dv:
        parts.append("HBM %.0f MB per launch (raw FETCH + WRITE)" % (dv["hbm_bytes_raw"] / 1e6))
    if "sq_active_inst_any_over_wave_cycles" in dv:
        parts.append("a wave's cycles: %.0f %% issuing, %.0f %% issue-stalled, %.0f %% parked on s_waitcnt" % (
            100 * dv["sq_active_inst_any_over_wave_cycles"], 100 * dv.get("sq_wait_inst_any_over_wave_cycles", 0),
            100 * dv.get("sq_wait_any_over_wave_cycles", 0)))
    if "valu_instr_per_analysis" in dv:
        parts.append("%.0f vector + %.1f matrix instructions per analysis" % (dv["valu_instr_per_analysis"],
                                                                            dv.get("mfma_f16_instr_per_analysis", 0)))
    return "; ".join(parts)


def make_case(G, k, stride, device, seed=42):
    """Synthetic inputs generated directly on the device (same distribution as
    oracle.synthetic_case; the seeded numpy version is used wherever values are compared)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    X = torch.randn((1, k, G), generator=gen, device=device, dtype=torch.float32)
    obs_x = torch.arange(0, G, stride, device=device, dtype=torch.float64)
    y = torch.randn(obs_x.shape[0], generator=gen, device=device, dtype=torch.float32)
    hx = X[0][:, ::stride]
    mean = hx.mean(dim=0)
    Yb = (hx - mean).contiguous()
    d = (y - mean).contiguous()
    grid_x = torch.arange(G, device=device, dtype=torch.float64)
    return X, grid_x, obs_x, Yb, d


def make_case_2d(ny, nx, k, stride, device, seed=42, m=1):
    """A 2-D mesh of ny x nx grid points in row-major order (the last coordinate runs fastest), unit spacing, observations at every
    `stride`-th point in BOTH dimensions, identity observation operator on the first state row, unit observation variance: the
    recipe of make_case on a mesh (examples/benchmark_letkf.py:107-149 on two coordinates)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    G = ny * nx
    X = torch.randn((m, k, G), generator=gen, device=device, dtype=torch.float32)
    ii, jj = torch.meshgrid(torch.arange(ny, device=device), torch.arange(nx, device=device), indexing="ij")
    grid = torch.stack([ii.reshape(-1), jj.reshape(-1)], dim=1).to(torch.float64).contiguous()
    sel = ((ii % stride == 0) & (jj % stride == 0)).reshape(-1).nonzero().reshape(-1)
    obs = grid[sel].contiguous()
    y = torch.randn(sel.shape[0], generator=gen, device=device, dtype=torch.float32)
    hx = X[0][:, sel]
    mean = hx.mean(dim=0)
    return X, grid, obs, (hx - mean).contiguous(), (y - mean).contiguous()


def tile_route_case(eng, X, grid, obs, Yb, d, radius, inf, burst=5, n_check=64, seed=5):
    """One geometry through the engine's tile route entry by entry (lists -> tile lists -> split records -> letkf_tile2_kernel),
    kernel time by HIP events around a burst of launches, tile statistics, `n_check` grid points against the oracle."""
    from oracle import letkf_oracle as O
    nb = eng.localize(grid, obs, [radius])
    extra, tiles = 0, None
    while eng.tile_route_applies(X, nb.p_max, extra, P=int(Yb.shape[1])):
        tiles = eng.localize_tiles(grid, obs, [radius], nb.p_max, extra_blocks=extra)
        n_over = int(tiles.stats[1].item())
        if n_over == 0:
            break
        first_over, tiles = n_over, None
        if n_over & (1 << 30):
            break
        extra += 1
    G = X.shape[-1]
    rec = {"p_max": int(nb.p_max), "mean_local_observations": float(nb.cnt.float().mean().item())}
    if tiles is None:       # the unions do not fit the tile format (or the shape is outside it): per-point lists, one point per wavefront
        fn = lambda: eng.analysis(X, Yb, d, nb, inf, return_flags=True, method="matfun")      # noqa: E731
        xa, fl = fn()
        rec.update(route="per-point lists (a tile's union exceeds the slots the ensemble size allows)", declined_points=None)
    else:
        srec = eng.pack_split(Yb, d)
        P = int(Yb.shape[1])
        out_buf = torch.empty((X.shape[0], X.shape[1], X.shape[2]), dtype=torch.float32, device=X.device)
        fn = lambda: eng.analysis_tiles(X, srec, P, tiles, inf, out=out_buf)      # noqa: E731
        xa, fl, retry = fn()
        hdr = tiles.unpack()[0]
        rec.update(route="tile lists + split records + letkf_tile2_kernel", union_slots=16 * tiles.ut, extra_blocks=extra,
                   mean_union=float(hdr[:, 0].mean()), max_union=int(hdr[:, 0].max()), overflowed_tiles=int(tiles.stats[1].item()),
                   declined_points=int(retry.item()))
        if int(retry.item()):
            eng.retry_points(X, Yb, d, nb, inf, xa, fl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(burst):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / burst
    pts = np.random.RandomState(seed).choice(G, n_check, replace=False)
    st, yb_h, d_h = X.double().cpu().numpy(), Yb.double().cpu().numpy(), d.double().cpu().numpy()
    gh, oh = grid.cpu().numpy(), obs.cpu().numpy()
    if gh.ndim == 1:
        gh, oh = gh[:, None], oh[:, None]
    ref = []
    for g in pts:
        near = np.all(np.abs(oh - gh[g]) <= 2.0 * radius + 1.0, axis=1)      # (the taper's support; the rest weighs zero)
        dist = O.grouped_euclid_distance(gh[g], oh[near], [0] * gh.shape[1], 1)
        w = O.localized_weights(dist, yb_h[:, near], d_h[near], [radius], inf)
        ref.append(O.apply_weights(st[:, :, [g]], w[None])[:, :, 0])
    ref = np.stack(ref, axis=-1)
    got = xa[:, :, torch.as_tensor(pts, device=xa.device)].double().cpu().numpy()
    dg = ((fl >> 8) & 0xff).float()
    rec.update(kernel_ms=ms, analyses_per_s=G / (ms * 1e-3), flags=int((fl & 0xff & ~8).max().item()),
               mean_chebyshev_degree=float(dg.mean().item()),
               rel_frobenius_error_vs_oracle=float(np.linalg.norm(got - ref) / np.linalg.norm(ref)), oracle_points=int(n_check))
    return rec


_CPU_CASE = None      # the workload of the CPU baseline: set in the parent before the pool forks (tasks carry indices only)


def _cpu_worker(pts):
    import torch as _t
    _t.set_num_threads(1)
    from oracle import letkf_oracle as O
    state, grid_x, obs_x, yb, d = (_CPU_CASE[n] for n in ("state", "grid_x", "obs_x", "yb", "d"))
    t0 = time.perf_counter()
    for g in pts:
        dist = O.abs_distance_1d(grid_x[g], obs_x)
        w = O.localized_weights(dist, yb, d, [GC_RADIUS], INF)
        O.apply_weights(state[:, :, [g]], w[None])
    return time.perf_counter() - t0


def _cpu_batched(case, pts, p_cap=24):
    """Best-effort CPU figure (last bullet): the same analyses as ONE batched computation per chunk --
    window gather, batched Gram, torch.linalg.eigh on (n, k, k), batched weights and transform, float64, all the
    process's threads."""
    import torch as _t
    st = _t.from_numpy(case["state"][0])                 # (k, G)
    yb, d = _t.from_numpy(case["yb"]), _t.from_numpy(case["d"])
    ox = _t.from_numpy(case["obs_x"])
    k = st.shape[0]
    g = _t.from_numpy(case["grid_x"][pts])
    first = _t.clamp(_t.ceil((g - 2 * GC_RADIUS) / OBS_STRIDE).long(), 0, ox.shape[0] - 1)
    raw = first[:, None] + _t.arange(p_cap)[None]                                        # (n, p_cap) window of candidates
    idx = _t.clamp(raw, 0, ox.shape[0] - 1)
    r = (ox[idx] - g[:, None]).abs() / GC_RADIUS
    r = _t.where(raw == idx, r, _t.full_like(r, 3.0))                                   # (beyond the last observation)
    f1 = (((-0.25 * r + 0.5) * r + 0.625) * r - 5.0 / 3.0) * r * r + 1.0
    f2 = ((((r / 12.0 - 0.5) * r + 0.625) * r + 5.0 / 3.0) * r - 5.0) * r + 4.0 - (2.0 / 3.0) / r.clamp_min(1e-300)
    w = _t.where(r < 1, f1, _t.where(r < 2, f2, _t.zeros_like(r)))
    w = _t.where(w > 1e-5, w, _t.zeros_like(w)).sqrt()
    Yl = yb[:, idx].permute(1, 0, 2) * w[:, None, :]                                   # (n, k, p)
    dl = d[idx] * w
    reg = (k - 1) / INF
    C = Yl @ Yl.transpose(1, 2)
    ev, V = _t.linalg.eigh(C)
    ev = ev.clamp_min(0) + reg
    wm = (V * (1.0 / ev)[:, None, :]) @ (V.transpose(1, 2) @ (Yl @ dl[:, :, None]))
    W = (V * ((k - 1) ** 0.5 / ev.sqrt())[:, None, :]) @ V.transpose(1, 2) + wm
    x = st[:, pts].T                                                                    # (n, k)
    xm = x.mean(dim=1, keepdim=True)
    return xm + ((x - xm)[:, None, :] @ W)[:, 0]


def cpu_baseline(n_points_total=12000):
    """The oracle (CPU restatement of the reference's per-grid-point path, torch float64, one thread per process, one
    process per core) on a bounded sample of the same workload -- the grid split into tasks of `chunksize` points as the
    reference's dask graph does (default 10, interface/letkf.py:80; 1000 next to it) -- plus the batched best-effort
    figure of SURVEY 8(d)."""
    import multiprocessing as mp
    from oracle import letkf_oracle as O
    # one worker per core of this process's CPU share (a one-GPU box grants 16; affinity, not the machine's core count)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    case = O.synthetic_case(G_PER_GPU, K_ENS, OBS_STRIDE)
    pts = np.random.RandomState(0).choice(G_PER_GPU, n_points_total, replace=False)
    global _CPU_CASE
    _CPU_CASE = case
    ctx = mp.get_context("fork")
    rates = {}
    with ctx.Pool(cores) as pool:
        for chunksize, n in ((10, n_points_total), (1000, n_points_total)):
            sub = pts[:n]
            tasks = [sub[i:i + chunksize] for i in range(0, n, chunksize)]
            t0 = time.perf_counter()
            pool.map(_cpu_worker, tasks, chunksize=1)
            rates[chunksize] = n / (time.perf_counter() - t0)
    torch.set_num_threads(cores)
    nb = 100000
    bp = np.random.RandomState(1).choice(G_PER_GPU, nb, replace=True)
    t0 = time.perf_counter()
    for i in range(0, nb, 20000):
        _cpu_batched(case, bp[i:i + 20000])
    batched = nb / (time.perf_counter() - t0)
    chk = _cpu_batched(case, pts[:64]).numpy()
    ref = np.stack([O.apply_weights(case["state"][:, :, [g]], O.localized_weights(
        O.abs_distance_1d(case["grid_x"][g], case["obs_x"]), case["yb"], case["d"], [GC_RADIUS], INF)[None])[0, :, 0] for g in pts[:64]])
    return {"value": rates[10], "unit": "analyses/s", "cores": cores, "kind": "port",
            "sample": "%d random grid points of the same G=1e5/P=5e4 problem, per-point python loop "
                      "(localize over all P obs -> mask/scale -> torch fp64 eigh weights -> transform), "
                      "1 thread/process, tasks of 10 grid points (the reference's default chunksize)" % n_points_total,
            "variants": {"chunksize_10": rates[10], "chunksize_1000": rates[1000],
                         "batched_torch_eigh_float64": batched,
                         "batched_note": "%d analyses as batched torch float64 (window gather, Gram, linalg.eigh on (n, 40, 40), "
                                         "weights, transform), %d threads -- SLOWER than the per-point loop on this host (torch's batched "
                                         "eigh of small float64 matrices does not use the threads): reported, not a better baseline; max "
                                         "|diff| to the per-point port on 64 points %.1e"
                                         % (nb, cores, float(np.abs(chk - ref).max()))}}


def launch_workers(n, argv):
    """One worker process per GPU (rank r on device r), rendezvous on 127.0.0.1; returns the exit status.
    Rank 0's stdout (the JSON line) is relayed, the other ranks' stdout goes to stderr.  If any worker fails the rest
    are terminated (by PID) and the status is non-zero."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    box = []
    reader = threading.Thread(target=lambda: box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    status, deadline = 0, time.time() + float(os.environ.get("MIA_BENCH_TIMEOUT", "1500"))
    while any(p.poll() is None for p in procs):
        failed = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if failed or time.time() > deadline:       # a rank died (or the run hangs): the others wait in a collective
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            time.sleep(2.0)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        time.sleep(0.05)
    for r, p in enumerate(procs):
        rc = p.wait()
        if rc != 0:
            status = status or (rc if rc > 0 else 1)
            sys.stderr.write("bench.py: worker rank %d exited with status %s\n" % (r, rc))
    reader.join(timeout=5.0)
    sys.stdout.write(box[0] if box else "")
    sys.stdout.flush()
    return status


def dryrun(args, rank, world):
    """MIA_BENCH_DRYRUN=1: the launcher / rendezvous / max-over-ranks timing / JSON plumbing of the N-rank run on a
    box without GPUs -- gloo backend, a small grid, the CPU oracle as the per-shard compute (the stand-in of
    tests/test_sharded_gloo.py).  Measures nothing: the line is marked ``dryrun``."""
    import torch.distributed as dist
    import torch_assimilate_amd as mia
    from oracle import letkf_oracle as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("MIA_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # launcher test: one rank dies before the rendezvous
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    gpg, k = 24, 12
    G = gpg * world
    case = O.synthetic_case(G, k, OBS_STRIDE)

    def shard(X, grid_x, obs_x, Yb, d, g0, g1):
        ana, _ = O.letkf_analysis(X.numpy()[:, :, g0:g1], grid_x.numpy()[g0:g1], obs_x.numpy(), Yb.numpy(), d.numpy(),
                                  GC_RADIUS, INF)
        return torch.from_numpy(ana)

    runner = mia.ShardedLetkf("cpu", rank, world, radii=[GC_RADIUS], inf_factor=INF, compute_shard=shard, comm_chunks=1)
    a = [torch.from_numpy(case[n]) for n in ("state", "grid_x", "obs_x", "yb", "d")]
    for _ in range(args.warmup):
        runner.assimilate(*a)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = runner.assimilate(*a)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = dist.get_world_size()
    if rank == 0:
        print(json.dumps({"metric": "local analyses/sec (LETKF, 40-member)", "value": G * args.steps / float(t.item()),
                          "unit": "analyses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * float(t.item()) / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dryrun": True,
                          "config": {"workload": "DRY RUN (no GPU): G=%d, k=%d, CPU oracle per shard, gloo" % (G, k),
                                     "parallelism": "grid-point block shard x%d + all-gather (%d ranks joined)" % (world, ranks)},
                          "shape_ok": list(out.shape) == [1, k, G]}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~0.4 s (2000 steps of ~0.2 ms) -- at 400 steps a single host hiccup of 30-40 ms (measured:
    # a generation-2 pass of Python's garbage collector, tools/host_jitter.py) moved the result by a third
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-4 / config-5 kernel timings")
    ap.add_argument("--grid-per-gpu", type=int, default=G_PER_GPU)
    ap.add_argument("--pipeline-depth", type=int, default=int(os.environ.get("MIA_PIPELINE_DEPTH", "8")), choices=list(range(1, 17)),
                    help="steps in flight (ShardedLetkf.submit): d (default 8) = steps i+1 .. i+d-1 are enqueued before step i "
                         "is collected; 1 = serial steps")
    ap.add_argument("--method", default="auto", choices=["auto", "eig", "matfun"],
                    help="analysis route: auto = eigensolver-free matfun kernel (default), eig = fused Jacobi")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher.  This parent never touches the GPU
        # (no HIP call, no torch.cuda call): it starts one fresh worker process per GPU and relays rank 0's line.
        raise SystemExit(launch_workers(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("MIA_BENCH_DRYRUN"):
        return dryrun(args, rank, world)
    # the CPU baseline forks worker processes: run it BEFORE this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import torch_assimilate_amd as mia
    from torch_assimilate_amd.sharded import ShardedLetkf
    mia.build()
    for opt in os.environ.get("MIA_BENCH_OPTIONS", "").split():       # A/B runs of tools/ab_options.sh: name=value ...
        from torch_assimilate_amd import _cabi
        _cabi.set_option(opt.split("=")[0], int(opt.split("=")[1]))
    gpg = args.grid_per_gpu
    G = gpg * world
    X, grid_x, obs_x, Yb, d = make_case(G, K_ENS, OBS_STRIDE, device)
    P = obs_x.shape[0]
    runner = ShardedLetkf(device, rank, world, radii=[GC_RADIUS], inf_factor=INF, method=args.method,
                          # pieces of the per-step exchange.  One step at a time: 4 pieces hide all but the last behind
                          # the analysis.  Steps in flight: the exchange of step i already travels during step i+1, so
                          # what counts is the exchange stream's total time per step -- one large all-gather (per-link
                          # bandwidth of a ring grows with message size, one latency instead of four) and the plain,
                          # unsegmented analysis launch (245 vs 266 us, no device-side segment waiters at all)
                          comm_chunks=int(os.environ.get("MIA_COMM_CHUNKS", "4" if args.pipeline_depth == 1 else "1")),
                          native_step=os.environ.get("MIA_NATIVE_STEP", "1") != "0",
                          max_in_flight=max(2, args.pipeline_depth),
                          # the fastest route measured under `--steps 20` (profiles/r05_stream_ab.txt): the analysis wavefronts localise
                          # their own tiles (letkf_tile2f_kernel: two launches per step), three analysis and three preparation streams
                          prep_streams=int(os.environ.get("MIA_PREP_STREAMS", "3")),
                          analysis_streams=int(os.environ.get("MIA_ANALYSIS_STREAMS", "3")),
                          fuse_tile_lists={"auto": "auto", "1": True, "0": False}[os.environ.get("MIA_FUSE_TILE_LISTS", "auto")],
                          prep_priority=int(os.environ.get("MIA_PREP_PRIORITY", "0")),
                          # N > 1: direct peer writes into IPC-mapped result buffers when the node allows it (self-tested
                          # at set-up, RCCL all-gather otherwise); the result is consumed from the slot buffer, no copy
                          peer_exchange=os.environ.get("MIA_PEER_EXCHANGE", "auto"), copy_results=False)

    def step(geometry_id=None):
        return runner.assimilate(X, grid_x, obs_x, Yb, d, geometry_id=geometry_id)

    exp_geo = os.environ.get("MIA_BENCH_GEOMETRY")           # (tools/ only: the main loop itself on a geometry epoch)

    def run(n_steps, depth, geometry_id=None):
        geometry_id = geometry_id or exp_geo
        """n_steps complete steps.  depth 1: each step is collected (host read-back, validation) before the next
        is enqueued.  depth d > 1: software pipeline over independent batches -- up to d steps are in flight, a
        later step's index / list kernels run beside an earlier step's analysis kernel and (N > 1) step i's
        all-gather travels during step i+1; every step is still fully computed, exchanged and validated
        inside the timed region."""
        out, pend = None, collections.deque()
        for it in range(n_steps):
            if it % 4 == 0:
                runner.time_next_step()      # HIP events around this step's analysis kernel, on its own stream
            if depth == 1:
                out = step(geometry_id)
            else:
                pend.append(runner.submit(X, grid_x, obs_x, Yb, d, geometry_id=geometry_id))
                if len(pend) == depth:
                    out = pend.popleft().result()
        while pend:
            out = pend.popleft().result()
        return out

    def timed(n_steps, depth, geometry_id=None):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(n_steps, depth, geometry_id)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    depth = args.pipeline_depth
    run(2 * depth + 2, depth)     # one pass over every pipeline slot: workspaces, coefficient table (~50 ms, no step's)
    # the interpreter's heap as it stands (torch, numpy: ~1e6 objects) out of the collector's way: a full collection
    # scanning it takes 30-40 ms, i.e. ~150 steps' worth of GPU idle time; what the loop itself allocates is still collected
    import gc
    gc.collect()
    gc.freeze()
    # warm-up LAST, straight into the timed loop: the collection above leaves the GPU idle for ~0.1 s and the clocks
    # take a few milliseconds of work to come back (20 timed steps right after it ran 25 % slower than steady state)
    # Untimed conditioning of clocks and caches; then EXACTLY --warmup steps straight into the timed region.  A timed region of
    # fewer than 200 steps lasts a few milliseconds (20 steps: 1.5 ms), less than the host's scheduling noise: such a region is
    # timed `repeats` times back to back (each bracketed by barrier + synchronize as the contract asks) and the MEDIAN
    # region is reported -- `steps` stays what was asked, `repeats` says how many regions were measured.
    run(max(50, args.warmup), depth)
    repeats = 1 if args.steps >= 200 else max(25, min(200, 4000 // max(args.steps, 1)))
    warm = args.warmup
    run(warm, depth)
    runner.kernel_timings.clear()
    co_l0, co_s0 = _cabi_stats()
    regions = []
    for _ in range(repeats):
        el_i, out = timed(args.steps, depth)
        regions.append(el_i)
    elapsed = float(np.median(regions))
    loop_kernel_ms = runner.kernel_ms()      # the dominant kernel's LAUNCHES inside the timed loop (every 4th step opens one)
    spl = runner.kernel_steps_per_launch() if loop_kernel_ms else 1.0      # launch coalescing: steps whose tiles one launch holds
    co_l1, co_s1 = _cabi_stats()
    runner._note_kernel()                    # ... and its name as the library launched it (before the other routes below run)
    kname = runner.dominant_kernel_name
    n_timed = len(runner.kernel_timings)
    serial_ms = None
    if depth != 1:           # secondary figure: the unpipelined step (latency of one step incl. its read-back)
        n_ser = max(100, args.steps // 8)
        run(10, 1)               # (the switch from steps in flight to one at a time: other streams, first-use set-up)
        el1, _ = timed(n_ser, 1)
        serial_ms = 1e3 * el1 / n_ser
    # separate key, NOT `value`: a geometry epoch (unchanged coordinates: the tile lists of each pipeline slot are built once,
    # every step still packs new records and runs the analysis) -- the reference recomputes the localisation on every call
    fixed_geo = None
    if world == 1:
        run(2 * depth + 2, depth, geometry_id="bench")
        r0 = runner.reused_steps
        elg, outg = timed(max(args.steps, 200), depth, "bench")
        run(10, 1, "bench")
        elg1, _ = timed(100, 1, "bench")
        fixed_geo = {"ms_per_step": 1e3 * elg / max(args.steps, 200), "analyses_per_s": G * max(args.steps, 200) / elg,
                     "serial_ms_per_step": 1e3 * elg1 / 100, "steps_on_reused_lists": runner.reused_steps - r0,
                     "bitwise_equal_to_full_rebuild": bool(torch.equal(outg, out)),
                     "note": "submit(..., geometry_id=): tile lists reused while grid / observation coordinates stay the same "
                             "(MIA_STEP_REUSE_LISTS); records, analysis, read-back every step"}
    # separate key, NOT `value`: the same loop on round 4's default route -- tile lists in memory (index_bucket -> localize_tiles ->
    # letkf_tile2_kernel), one analysis stream, five preparation streams -- so that the two routes are measured in one process, and
    # their results compared bit for bit
    lists_flight = None
    if world == 1 and depth != 1 and args.method != "eig":
        r2 = ShardedLetkf(device, rank, world, radii=[GC_RADIUS], inf_factor=INF, method=args.method, comm_chunks=1,
                          max_in_flight=max(2, depth), prep_streams=5, analysis_streams=1, copy_results=False, fuse_tile_lists=False)

        def run2(n):
            pend, o = collections.deque(), None
            for it in range(n):
                if it % 4 == 0:
                    r2.time_next_step()
                pend.append(r2.submit(X, grid_x, obs_x, Yb, d))
                if len(pend) == depth:
                    o = pend.popleft().result()
            while pend:
                o = pend.popleft().result()
            return o
        run2(2 * depth + 2)
        run2(max(50, args.warmup))
        r2.kernel_timings.clear()
        reg = []
        for _ in range(max(5, repeats // 4)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o2 = run2(args.steps)
            torch.cuda.synchronize()
            reg.append(time.perf_counter() - t0)
        el2 = float(np.median(reg))
        lists_flight = {"ms_per_step": 1e3 * el2 / args.steps, "analyses_per_s": G * args.steps / el2,
                        "kernel_ms_in_loop": r2.kernel_ms(), "kernel": r2.dominant_kernel_name,
                        "analysis_streams": 1, "preparation_streams": 5, "bitwise_equal_to_value_route": bool(torch.equal(o2, out)),
                        "note": "ShardedLetkf(fuse_tile_lists=False, analysis_streams=1, prep_streams=5): round 4's default; the list "
                                "kernel of step i + 1 runs beside the analysis kernel of step i, whose launch is the shorter one"}
        r2.close()
    assert out.shape == (1, K_ENS, G) and bool(torch.isfinite(out).all())
    assert runner.last_flags_ok(), "kernel flagged grid points"

    # ---- dominant kernel alone, HIP events on the launch stream (torch's current stream)
    alone_ms, stage_ms = runner.time_stages(X, grid_x, obs_x, Yb, d, reps=max(5, min(args.steps, 20)))
    kern_ms = loop_kernel_ms if loop_kernel_ms else alone_ms
    p_max = runner.last_p_max
    deg = runner.mean_degree()
    deg_tile = runner.mean_tile_degree()
    kkind = "tile2" if "tile2" in kname else (("tile_split" if "true>" in kname else "tile") if "tile" in kname else "point")

    def rates(k, p, m, dg, n_pts, ms, kind):
        """executed / useful / reference-credit flop rates of one launch of n_pts analyses in `ms`"""
        ex, us = executed_flops(k, p, m, dg, kind), useful_flops(k, p, m, dg)
        f16 = None
        if isinstance(ex, tuple):
            ex, f16 = ex
        tf = lambda f: None if f is None else f * n_pts / (ms * 1e-3) / 1e12
        return ex, us, tf(ex), tf(us), tf(algorithmic_flops(k, p, m)), f16, tf(f16)

    gpl = gpg * spl                          # analyses per launch of the dominant kernel
    ex_f, us_f, ex_tf, us_tf, credit_tf, f16_f, f16_tf = rates(K_ENS, 20, 1, deg, gpl, kern_ms, kkind)
    hbm_alg = algorithmic_bytes(K_ENS, 1, P / G) * gpl / (kern_ms * 1e-3) / 1e9

    # secondary figure: the fused Jacobi-eigensolver route on the same shard (kernel only)
    eig_ms = None
    if args.method != "eig":
        r2 = ShardedLetkf(device, rank, world, radii=[GC_RADIUS], inf_factor=INF, method="eig")
        r2._engine = runner.engine
        eig_ms, _ = r2.time_stages(X, grid_x, obs_x, Yb, d, reps=5)

    # ---- N > 1: exchange and compute apart, so that a scaling figure can be read: this rank's exchange alone
    #      (HIP events on the exchange stream), its compute alone (serial step of its block without exchange = a one-rank
    #      runner on the block's grid points), the bytes it puts on its xGMI links per step and the time those bytes take at the
    #      link rate of MI355X_MICROARCH.md / DESIGN.md section 5 (153 GB/s per link taken as bidirectional: ~77 GB/s each way)
    multi = None
    if world > 1:
        try:
            ex_ms, ex_route, ex_bytes = runner.time_exchange(1, K_ENS, G, reps=10)
            g0b, g1b = rank * gpg, min(G, (rank + 1) * gpg)
            solo = ShardedLetkf(device, 0, 1, radii=[GC_RADIUS], inf_factor=INF, method=args.method)
            solo._engine = runner.engine
            Xb = X[:, :, g0b:g1b].contiguous()
            gb = grid_x[g0b:g1b].contiguous()
            for _ in range(5):
                solo.assimilate(Xb, gb, obs_x, Yb, d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                solo.assimilate(Xb, gb, obs_x, Yb, d)
            torch.cuda.synchronize()
            comp_ms = 1e3 * (time.perf_counter() - t0) / 50
            # the analysis kept block-sharded (ShardedLetkf(gather=False): what the reference's dask chunks along `grid` do,
            # interface/letkf.py:118-131): the same pipelined loop without the all-gather -- compute scaling by itself.  The loop
            # itself enters no collective (nothing is exchanged): a rank that fails here reports -1 and the maximum over the ranks,
            # which every rank reaches, says so
            n_keep = max(args.steps, 200)
            keep_local = -1.0
            try:
                keep = ShardedLetkf(device, rank, world, radii=[GC_RADIUS], inf_factor=INF, method=args.method, gather=False,
                                    max_in_flight=max(2, args.pipeline_depth))
                keep._engine = runner.engine
                for _ in range(3):
                    keep.assimilate(X, grid_x, obs_x, Yb, d)

                def keep_loop(n):
                    pend = []
                    for _ in range(n):
                        pend.append(keep.submit(X, grid_x, obs_x, Yb, d))
                        if len(pend) == max(2, args.pipeline_depth):
                            pend.pop(0).result()
                    for h in pend:
                        h.result()
                keep_loop(50)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                keep_loop(n_keep)
                torch.cuda.synchronize()
                keep_local = time.perf_counter() - t0
                keep.close()
            except Exception as exc:      # noqa: BLE001
                sys.stderr.write("sharded-output loop failed on rank %d: %r\n" % (rank, exc))
            keep_t = torch.tensor([keep_local, -keep_local], device=device, dtype=torch.float64)
            dist.all_reduce(keep_t, op=dist.ReduceOp.MAX)
            keep_ms = None if float(keep_t[1].item()) > 0 else 1e3 * float(keep_t[0].item()) / n_keep      # (any rank at -1: no figure)
            vals = torch.tensor([ex_ms, comp_ms], device=device, dtype=torch.float64)
            allv = [torch.zeros_like(vals) for _ in range(world)]
            dist.all_gather(allv, vals)
            per_link = ex_bytes / max(world - 1, 1)
            multi = {"exchange_route": runner.exchange_route, "exchange_ms_per_rank": [float(v[0]) for v in allv],
                     "compute_ms_per_rank_serial_block": [float(v[1]) for v in allv],
                     "sharded_output": {"ms_per_step": keep_ms, "analyses_per_s": (G / (keep_ms * 1e-3)) if keep_ms else None, "steps": n_keep,
                                        "note": "ShardedLetkf(gather=False): every rank keeps its block of the analysis (the reference's "
                                                "dask chunks along `grid`); max over ranks, barrier + synchronize on both sides: the "
                                                "compute scaling without the all-gather -- `value` is the gathered metric"},
                     "bytes_sent_per_rank_per_step": ex_bytes, "bytes_per_link_per_step": per_link,
                     "xgmi_budget_ms": {"at_77_GBs_per_direction": 1e3 * per_link / 77e9, "at_153_GBs_per_direction": 1e3 * per_link / 153e9,
                                        "note": "direct route: every rank writes its block to each of its world - 1 peers over its own link "
                                                "to that peer, all links at once; the RCCL ring forwards world - 1 blocks over one link pair"},
                     "note": "exchange alone (no analysis in front of it) and a serial step of this rank's block without any exchange; the "
                             "pipelined step overlaps step i's exchange with step i+1's compute, so ms_per_step ~ max(exchange, compute "
                             "pipelined) -- with the exchange the larger one at N = 8 by the budget above (>= 6x is xGMI-bound)"}
        except Exception as exc:      # (diagnostics only: never let them take the headline down)
            multi = {"error": repr(exc)}

    # ---- one step END TO END from pinned host buffers: H2D of state / obs-space inputs / coordinates,
    #      the step, D2H of the analysis ensemble (never part of `value`)
    e2e_ms = None
    if world == 1 and rank == 0:
        host = [t.cpu().pin_memory() for t in (X, grid_x, obs_x, Yb, d)]
        out_h = torch.empty((1, K_ENS, G), dtype=torch.float32).pin_memory()
        dev_in = [torch.empty_like(t) for t in (X, grid_x, obs_x, Yb, d)]
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, h in zip(dev_in, host):
                a.copy_(h, non_blocking=True)
            o = runner.assimilate(*dev_in)
            out_h.copy_(o, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        e2e_ms = 1e3 * float(np.median(ts[2:]))

    # ---- the other single-GPU configurations of BASELINE.json, timed in this same process: kernel only (burst of five
    #      launches, HIP events), 64 sampled grid points checked against the oracle
    secondary = {}
    if world == 1 and rank == 0 and not args.no_secondary:
        from oracle import letkf_oracle as O
        for name, (k2, stride2, c2, gamma2) in (("c4", (80, 1, 16.5, None)), ("c5", (40, 2, 10.0, 0.5))):
            X2, gx2, ox2, Yb2, d2 = make_case(gpg, k2, stride2, device, seed=43)
            r3 = ShardedLetkf(device, 0, 1, radii=[c2], inf_factor=INF, rbf_gamma=gamma2, method=args.method, native_step=False)
            r3._engine = runner.engine
            xa2 = r3.assimilate(X2, gx2, ox2, Yb2, d2)
            ms2, _ = r3.time_stages(X2, gx2, ox2, Yb2, d2, reps=5)
            pm2, deg2 = r3.last_p_max, r3.mean_degree()
            pts = np.random.RandomState(2).choice(gpg, 64, replace=False)
            st, yb_h, d_h = X2.double().cpu().numpy(), Yb2.double().cpu().numpy(), d2.double().cpu().numpy()
            gxh, oxh = gx2.cpu().numpy(), ox2.cpu().numpy()
            core = O.etkf_weights if gamma2 is None else (
                lambda a, b, inf, g_=gamma2: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, g_), inf))
            ref = np.stack([O.apply_weights(st[:, :, [g]], O.localized_weights(O.abs_distance_1d(gxh[g], oxh), yb_h, d_h,
                                                                              [c2], INF, core=core)[None])[:, :, 0] for g in pts], axis=-1)
            got = xa2[:, :, torch.as_tensor(pts, device=device)].double().cpu().numpy()
            rec = {"workload": "G=%d, k=%d, obs every %d, Gaspari-Cohn radius %g (<=%d local obs)%s, m=1"
                               % (gpg, k2, stride2, c2, pm2, ", RBF kernel gamma %.1f" % gamma2 if gamma2 else ""),
                   "kernel_ms": ms2, "analyses_per_s": gpg / (ms2 * 1e-3), "mean_chebyshev_degree": deg2,
                   "rel_frobenius_error_64_points_vs_oracle": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)),
                   "reference_algorithm_credit_TFLOPs": algorithmic_flops(k2, pm2, 1) * gpg / (ms2 * 1e-3) / 1e12}
            if gamma2 is not None:
                rec["kernel"] = r3.dominant_kernel_name + " (tile lists + the f32 perturbations themselves; pair statistic of sixteen points as one split-f16 MFMA product)" \
                    if r3.dominant_kernel_name.startswith("lketkf") else "letkf_cheb_kernel (one grid point per wavefront)"
                rec["mean_chebyshev_degree_per_tile_max"] = r3.mean_tile_degree()
            if gamma2 is None and deg2:
                tiles2 = r3.engine.tile_route_applies(X2, pm2, r3._tile_extra) and not r3._no_tile_lists
                kk = "tile2" if tiles2 else ((kkind if kkind not in ("point", "tile2") else "tile_split") if (k2 <= 96 and pm2 + 8 <= 96) else "point")
                exf = executed_flops(k2, pm2, 1, deg2, kk)
                usf = useful_flops(k2, pm2, 1, deg2)
                if isinstance(exf, tuple):
                    rec.update(f16_mfma_flops_per_analysis=exf[1],
                               matrix_core_frac=exf[1] * gpg / (ms2 * 1e-3) / 1e12 / PEAK_F16_TFLOPS)
                    exf = exf[0]
                uss = useful_flops_shared(k2, pm2, 1, deg2, u=min(pm2 + 15, 16 * ((pm2 + 8 + 15) // 16)))
                rec.update(useful_flops_per_analysis=uss, useful_frac=uss * gpg / (ms2 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                           useful_basis="useful f32 flops with the Gram matrix counted once per tile of sixteen points (no padding); "
                                        "a per-point algorithm's count (unshared_*) credits the shared Gram sixteen times and can exceed "
                                        "what any kernel executes",
                           unshared_flops_per_analysis=usf, unshared_credit_ratio_to_fp32_peak=usf * gpg / (ms2 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                           executed_f32_equivalent_flops_per_analysis=exf,
                           executed_f32_equivalent_ratio_to_fp32_peak=exf * gpg / (ms2 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                           executed_note="f32-equivalent of the half-precision MFMA triples, padding included: runs on the "
                                         "half-precision matrix pipe, not bounded by 1, not a utilisation (useful_frac is)",
                           kernel={"tile2": "letkf_tile2_kernel (tile lists + split records)",
                                   "tile_split": "letkf_tile_kernel (split half-precision products)",
                                   "tile": "letkf_tile_kernel (f32 products)",
                                   "point": "letkf_cheb_kernel (one grid point per wavefront)"}[kk])
            secondary[name] = rec

        # off the line and more than one state row: a 316 x 316 mesh in row-major order with observations at every
        # second point in both dimensions (localisation over real meshes: gaspari_cohn.py:124-134), and config 2 with eight
        # state rows per grid point (n_var * n_time values, interface/base.py:257-278)
        for name, build in (("c2_mesh_2d", lambda: make_case_2d(316, 316, K_ENS, 2, device, seed=44)),
                            ("c2_m8", lambda: (lambda c: (c[0].repeat(8, 1, 1) * torch.linspace(0.5, 2.0, 8, device=device)[:, None, None],) + c[1:])(
                                make_case(gpg, K_ENS, OBS_STRIDE, device, seed=45)))):
            try:
                X6, g6, o6, Yb6, d6 = build()
                rad6 = 2.5 if name == "c2_mesh_2d" else GC_RADIUS
                rec6 = tile_route_case(runner.engine, X6.contiguous(), g6, o6, Yb6, d6, rad6, INF)
                rec6["workload"] = ("316 x 316 mesh (99 856 points, row-major), k=%d, observations at every 2nd point in both dimensions "
                                    "(24 964), Gaspari-Cohn radius %g on the Euclidean distance, m=1" % (K_ENS, rad6)) if name == "c2_mesh_2d" \
                    else "config 2 with m=8 state rows per grid point (G=%d, k=%d)" % (gpg, K_ENS)
                secondary[name] = rec6
                del X6, g6, o6, Yb6, d6
            except Exception as exc:        # (a secondary figure must not take the bench line down)
                secondary[name] = {"error": repr(exc)}

        # LETKF.estimate_weights (interface/letkf.py:127-146) on the tile route: analysis + the (G, k, k) weights
        try:
            eng = runner.engine
            pmw = int(runner.last_p_max or 20)
            tl = eng.localize_tiles(grid_x, obs_x, [GC_RADIUS], pmw)
            srec = eng.pack_split(Yb, d)
            res = eng.weights_tiles(X, srec, P, tl, INF)
            if res is not None and int(tl.stats[1].item()) == 0:
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    res = eng.weights_tiles(X, srec, P, tl, INF)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                Wd = res[1]
                pts = np.random.RandomState(3).choice(G, 64, replace=False)
                st, yb_h, d_h = X.double().cpu().numpy(), Yb.double().cpu().numpy(), d.double().cpu().numpy()
                gxh, oxh = grid_x.cpu().numpy(), obs_x.cpu().numpy()
                refw = np.stack([O.localized_weights(O.abs_distance_1d(gxh[g], oxh), yb_h, d_h, [GC_RADIUS], INF) for g in pts])
                gotw = Wd[torch.as_tensor(pts, device=device)].double().cpu().numpy()
                wms = float(np.median(ts[1:]))
                secondary["c2_weights"] = {
                    "workload": "G=%d, k=%d: the analysis AND the (G, k, k) weights (estimate_weights), tile route "
                                "(mia_letkf_weights_tiles_f32: letkf_tile2_kernel + letkf_tile2w_kernel)" % (G, K_ENS),
                    "ms": wms, "analyses_per_s": G / (wms * 1e-3), "weights_bytes": int(Wd.numel() * 4),
                    "hbm_floor_ms": Wd.numel() * 4 / 8e12 * 1e3, "declined_points": int(res[3].item()),
                    "rel_frobenius_error_64_points_vs_oracle": float(np.linalg.norm(gotw - refw) / np.linalg.norm(refw))}
                # _apply_weights with those per-point weights (interface/base.py:257-278) on 16 state rows: csrc/apply_local.hip
                try:
                    gen = torch.Generator(device=device)
                    gen.manual_seed(7)
                    m_rows = 16
                    Xr = torch.randn((m_rows, K_ENS, G), generator=gen, device=device)
                    Xr[0] = X[0]
                    xa_r = eng.apply_local_weights(Xr, Wd)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(7):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        xa_r = eng.apply_local_weights(Xr, Wd)
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    ams = float(np.median(ts[1:]))
                    nbytes = 4.0 * (Wd.numel() + 2.0 * Xr.numel())
                    refx = O.apply_weights(Xr[:, :, torch.as_tensor(pts, device=device)].double().cpu().numpy(), gotw)
                    gotx = xa_r[:, :, torch.as_tensor(pts, device=device)].double().cpu().numpy()
                    secondary["c2_apply_weights"] = {
                        "workload": "G=%d, k=%d: xa = mean + (x - mean) W_g for %d state rows per grid point, W from c2_weights "
                                    "(mia_apply_local_weights_f32: apply_local_tile_kernel<3>)" % (G, K_ENS, m_rows),
                        "ms": ams, "compulsory_bytes": nbytes, "GBs": nbytes / ams / 1e6, "hbm_frac": nbytes / ams / 1e6 / PEAK_HBM_GBS,
                        "rel_frobenius_error_64_points_vs_oracle": float(np.linalg.norm(gotx - refx) / np.linalg.norm(refx))}
                    del Xr, xa_r
                except Exception as exc:
                    secondary["c2_apply_weights"] = {"error": repr(exc)}
                del Wd, res
        except Exception as exc:        # (a secondary figure must not take the bench line down)
            secondary["c2_weights"] = {"error": repr(exc)}

    if rank == 0:
        value = G * args.steps / elapsed
        prof = profile_summary(world, launched=kname)
        pdv = (prof or {}).get("derived", {})
        line = {
            "metric": "local analyses/sec (LETKF, 40-member)", "value": value, "unit": "analyses/s",
            "n_gpus": world, "steps": args.steps, "warmup": warm,
            "repeats": repeats,
            "repeats_note": ("the timed region of --steps steps was measured %d times back to back (each bracketed by barrier + "
                             "synchronize); value / ms_per_step are the median region's, spread (min .. max) %.4f .. %.4f ms per step; every region fills "
                             "and drains the pipeline of steps in flight (1.5-2 step latencies, 0.15-0.2 ms, per region: a 20-step region "
                             "is 12-15 %% slower per step than the default 2000-step region, profiles/r03_final_bench_steps20.json)"
                             % (repeats, 1e3 * min(regions) / args.steps, 1e3 * max(regions) / args.steps)) if repeats > 1 else None,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_detail": ("inputs, outputs, accumulation and every vector operation float32; the matrix products run as "
                             "three half-precision MFMAs per f32 product on operands carried as pairs of halves (22-23 "
                             "significant bits; measured error equal to f32 products', DESIGN.md 3.0)")
                            if kkind in ("tile_split", "tile2") else "float32 throughout",
            "config": {"workload": "LETKF config %s: G=%d grid points (%d per GPU), k=%d members, P=%d obs, "
                                   "Gaspari-Cohn radius %g (<=%d local obs), inf %.1f, m=1"
                                   % ("2" if world == 1 else "3-style", G, gpg, K_ENS, P, GC_RADIUS, p_max, INF),
                       "parallelism": "grid-point block shard x%d%s" % (world, " + all-gather of the analysis ensemble over xGMI "
                                                                        "(%s)" % runner.exchange_route if world > 1 else ""),
                       "ranks": world},
            # SURVEY 8(d) / the round contract: achieved = ALGORITHMIC bytes per launch (414 B per analysis at config 2: state in,
            # analysis out, each observation's record and coordinates once) / the dominant kernel's launch duration measured in the
            # timed loop; peak = HBM.  With the eigensolve gone (matrix functions instead: no 9 k^3 term is executed) HBM is the
            # governing algorithmic roof; the ceilings that bind in practice -- vector issue, the matrix pipe -- are listed beside it.
            "roofline": {"bound": "hbm",
                         "achieved": hbm_alg, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_alg / PEAK_HBM_GBS,
                         "frac_basis": "%.0f algorithmic bytes per analysis (SURVEY 8(d): 4 (2 k m + (k + 1) P / G + 2 n_coord (1 + P / G))) x %.0f "
                                       "analyses per launch (%d per step x %.2f steps per launch) / kernel_ms / 8 TB/s"
                                       % (algorithmic_bytes(K_ENS, 1, P / G), gpl, gpg, spl),
                         "algorithmic_bytes_per_launch": algorithmic_bytes(K_ENS, 1, P / G) * gpl,
                         "steps_per_launch": spl, "analyses_per_launch": gpl, "kernel_ms_per_step": kern_ms / spl,
                         "steps_per_launch_note": ("launch coalescing (option step_coalesce) is ON: the launch thread hands the tiles of up to four "
                                                   "steps in flight whose preparation has finished to ONE grid of the fused kernel; mean over "
                                                   "the timed launches; over the whole timed loop %s launches carried %s steps"
                                                   % (co_l1 - co_l0, co_s1 - co_s0)) if co_l1 > co_l0 else
                                                  "one launch per step (launch coalescing, option step_coalesce, is off: the default -- it measured "
                                                  "slower per step, profiles/r05_coalesce.txt)",
                         "step_frac": algorithmic_bytes(K_ENS, 1, P / G) * gpg / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS,
                         "step_frac_note": "the same bytes / ms_per_step / peak: the whole step (index + packing + fused analysis) "
                                           "against the HBM roof",
                         "frac_alone": algorithmic_bytes(K_ENS, 1, P / G) * gpg / (alone_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if alone_ms else None,
                         "frac_alone_note": "one step's launch alone on an idle GPU (kernel_ms_alone)",
                         "kernel": kernel_base_name(kname), "kernel_ms": kern_ms,
                         "kernel_ms_source": ("start / stop HIP events of the kernel's own dispatch (hipExtLaunchKernel, on the analysis "
                                              "stream it is launched on) on every 4th step of the timed loop (%d launches, each opened by the "
                                              "timed step and holding steps_per_launch steps); with %d "
                                              "analysis streams up to that many of these launches share the chip, so a launch lasts "
                                              "longer than it does alone (kernel_ms_alone: letkf_tile2_kernel on lists in memory, burst "
                                              "of launches on an idle GPU)" % (n_timed, runner.analysis_streams))
                                             if loop_kernel_ms else "burst of 5 launches after the timed loop",
                         "kernel_ms_alone": alone_ms,
                         "traffic": pdv.get("hbm_bytes_fetch_doubled"),
                         "traffic_raw": pdv.get("hbm_bytes_raw"),
                         "traffic_source": ("offline: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload, "
                                            "read from profiles/latest_c2.json (%s); gfx950 counts half of wide coalesced streaming "
                                            "reads in FETCH_SIZE (MI355X_MICROARCH.md): `traffic` doubles the fetch side, traffic_raw "
                                            "does not" % (prof or {}).get("source")) if prof else
                                           "none: profiles/latest_c2.json is absent or is a profile of another kernel / workload",
                         "kernel_name_recorded_by_rocprof": (prof or {}).get("kernel_name_recorded"),
                         "kernel_average_ns_rocprof": ((prof or {}).get("kernel_trace") or {}).get("average_ns"),
                         "bound_evidence": bound_evidence(prof),
                         # the ceilings beside the algorithmic one: which unit is how busy (counters: kernel alone, profiles/)
                         "ceilings": {"hbm_algorithmic": hbm_alg / PEAK_HBM_GBS,
                                      "matrix_core_f16": None if f16_tf is None else f16_tf / PEAK_F16_TFLOPS,
                                      "vector_issue_busy": pdv.get("valu_busy_frac"), "matrix_pipe_busy": pdv.get("mfma_busy_frac"),
                                      "wait_any_over_wave_cycles": pdv.get("sq_wait_any_over_wave_cycles"),
                                      "note": "the kernel is bound by instruction issue and latency (vector unit ~50 % busy, a wave "
                                              "issues on a quarter of its cycles), not by bytes or matrix flops: every fraction "
                                              "here is far from 1, the largest is the vector unit's"},
                         "compute_view": {
                             "frac_useful_shared": None if not deg else useful_flops_shared(K_ENS, 20, 1, deg) * gpl / (kern_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                             "frac_useful_shared_basis": "useful f32 flops with the Gram matrix counted ONCE per tile of sixteen points "
                                                         "(k U (U+1) / 16 per analysis, U = 28) + per state row 4 k p + 2 deg p^2 + taper "
                                                         "25 p, no padding / kernel_ms / FP32 peak 157.3 TFLOP/s",
                             "frac_unshared_credit": None if us_tf is None else us_tf / PEAK_FP32_TFLOPS,
                             "frac_unshared_credit_note": "rounds 2-4 quoted this as `frac`: it credits the shared Gram matrix sixteen "
                                                          "times (a per-point algorithm's count) -- kept for continuity, not a roofline",
                             "useful_flops_per_analysis_unshared": us_f,
                             "executed_f32_equivalent": None if ex_tf is None else {
                                 "flops_per_analysis": ex_f, "TFLOPs": ex_tf, "ratio_to_fp32_peak": ex_tf / PEAK_FP32_TFLOPS,
                                 "note": "the f32 products (padding to 16 x 16 x 32 blocks and the shared Gram matrix included) that the "
                                         "kernel's half-precision MFMA triples implement + its vector-unit update; they run on the "
                                         "half-precision matrix pipe, so this ratio is NOT bounded by 1 and is not a utilisation"},
                             "matrix_core": None if f16_tf is None else {
                                 "executed_f16_mfma_flops_per_analysis": f16_f, "achieved": f16_tf, "peak": PEAK_F16_TFLOPS,
                                 "unit": "TFLOP/s", "frac": f16_tf / PEAK_F16_TFLOPS,
                                 "note": "v_mfma_f32_16x16x32_f16, three per f32 product"},
                             "reference_algorithm_credit": {"flops_per_analysis": algorithmic_flops(K_ENS, 20, 1), "TFLOPs": credit_tf,
                                                            "ratio_to_fp32_peak": credit_tf / PEAK_FP32_TFLOPS,
                                                            "note": "SURVEY 8(d) count of the reference's algorithm (9k^3 symmetric-QR "
                                                                    "eigensolve) per analysis: NOT executed by the matrix-function "
                                                                    "route (the ratio exceeds 1), not a utilisation"}},
                         "note": "sixteen grid points per wavefront share one Gram matrix; every contraction is a triple of "
                                 "half-precision MFMAs on operands carried as pairs of halves (f32 accuracy).  The wavefront "
                                 "localises its own tile over the step's bucket index (Gaspari-Cohn, union, ranks, sqrt(rho)), then "
                                 "gathers the union's split records straight into LDS (LDS-DMA) and analyses the tile."},
            "route": {"method": args.method, "mean_chebyshev_degree": deg, "mean_chebyshev_degree_per_tile_max": deg_tile,
                      "degree_note": "per-point mean (what the useful-flop credit uses) and the mean over tiles of the largest degree "
                                     "in the tile (what a tile's wavefront executes)",
                      "declined_points_last_step": runner.last_retries,
                      "eigensolver_route_kernel_ms": eig_ms,
                      "eigensolver_route_kernel_analyses_per_s": (gpg / (eig_ms * 1e-3)) if eig_ms else None},
            "pipeline": {"depth": depth, "serial_ms_per_step": serial_ms,
                         "route": ("index_bucket (+ record packing) -> letkf_tile2f_kernel (the wavefronts localise their own "
                                   "tiles: two launches per step)") if "tile2f" in kname
                                  else "index_bucket (+ record packing) -> localize_tiles -> %s" % kernel_base_name(kname),
                         "analysis_streams": runner.analysis_streams,
                         "preparation_streams": {"n": runner.prep_streams,
                                                 "note": "plain HIP streams taken in turn by the steps in flight; stream counts "
                                                         "A/B under --steps 20: profiles/r05_stream_ab.txt"},
                         "fixed_geometry": fixed_geo,
                         "lists_in_memory_in_flight": lists_flight,
                         "serial_analyses_per_s": (G / (serial_ms * 1e-3)) if serial_ms else None,
                         "note": "depth d > 1: consecutive (independent) steps are software-pipelined over d slots / HIP "
                                 "streams; every step is fully computed, exchanged and validated inside the timed "
                                 "region.  serial_*: the same step run one at a time (what a cycled filter, whose step i+1 "
                                 "depends on step i, gets)"},
            "multi_gpu": multi,
            # stated BEFORE any 8-GPU run exists, from this run's own step time and the xGMI link rate: at N ranks
            # every rank sends its (m k G/N... here gpg-point) block to N - 1 peers, one link each, all links at once
            "multi_gpu_prediction": (lambda t1, blk: {
                "step_ms_1gpu": t1, "block_bytes_per_rank": blk,
                "exchange_ms_at_77_GBs_per_link_direction": 1e3 * blk / 77e9,
                "predicted_speedup_gathered": {str(n): n * t1 / max(t1, 1e3 * blk / 77e9) for n in (2, 4, 8)},
                "predicted_speedup_sharded_output": {str(n): float(n) for n in (2, 4, 8)},
                "note": "weak scaling, %d points per rank: gathered = every rank ends with the whole analysis (the all-gather's %.0f MB "
                        "per link and step bound the step once they take longer than the compute: >= 6x is out of reach of the gathered "
                        "metric on xGMI); sharded output (ShardedLetkf(gather=False), what the reference's dask chunks do) has no "
                        "exchange.  No N > 1 measurement exists yet (no 8-GPU node was available to any round)" % (gpg, blk / 1e6)})(
                    1e3 * elapsed / args.steps if world == 1 else None, 4.0 * K_ENS * gpg) if world == 1 else None,
            "e2e_ms_incl_h2d_d2h": e2e_ms,
            "stages_ms": stage_ms,
            "secondary": secondary,
            "step_driver": ("native: one C call per step (mia_letkf_sharded_step_streams_f32); %d native steps in this process "
                            "(warm-up, timed loop, serial comparison)" % runner.native_steps)
                           if runner.native_steps else "python (engine entries one by one)",
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

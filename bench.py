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
build the observation cell index, Gaspari-Cohn neighbour lists, fused local analysis
(Gram, matrix functions / eigensolve, weights, transform) [, all-gather], enqueued by ONE call
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
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(k, p, m):
    """k^2 p [Gram] + 2kp [Yb d] + 9k^3 [sym-eig, Golub-Van Loan count] + k^3 [W]
    + 5k^2 [w_mean] + 2k^2 m [transform] + 25p [GC]."""
    return k * k * p + 2 * k * p + 9 * k ** 3 + k ** 3 + 5 * k * k + 2 * k * k * m + 25 * p


def executed_flops(k, p, m, deg):
    """What the matfun kernel really executes per analysis (dual route): Gram on 16x16x4 MFMA tiles over
    the padded order (upper tiles), z and output products, Chebyshev coefficients and recurrence."""
    if not deg:
        return None
    tt = (max(p, 1) + 15) // 16
    ks = (k + 3) // 4
    gram = tt * (tt + 1) // 2 * ks * 2048
    n = (p + 3) // 4 * 4
    return gram + m * (2 * k * n * 2 + deg * 2 * n * n) + 4 * (deg + 1) ** 2 + 25 * p


def algorithmic_bytes(k, m, P_over_G, n_coord=1):
    return 4.0 * (2 * k * m + (k + 1) * P_over_G + 2 * n_coord * (1 + P_over_G))


def traffic_from_profiles(world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/*traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes on this same
    workload; bench.py cannot run the profiler on itself).  None when no matching record exists."""
    path = os.path.join(ROOT, "profiles", "latest_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("grid_points") != G_PER_GPU or rec.get("k") != K_ENS:
        return None
    return rec


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


def _cpu_worker(args):
    import torch as _t
    _t.set_num_threads(1)
    from oracle import letkf_oracle as O
    state, grid_x, obs_x, yb, d, pts = args
    t0 = time.perf_counter()
    for g in pts:
        dist = O.abs_distance_1d(grid_x[g], obs_x)
        w = O.localized_weights(dist, yb, d, [GC_RADIUS], INF)
        O.apply_weights(state[:, :, [g]], w[None])
    return time.perf_counter() - t0


def cpu_baseline(n_points_total=16000):
    """The oracle (CPU restatement of the reference's per-grid-point path, torch float64, one
    thread per process, one process per core) on a bounded sample of the same workload."""
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
    chunks = np.array_split(pts, cores)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c)
                               for c in chunks])
    wall = time.perf_counter() - t0
    return {"value": n_points_total / wall, "unit": "analyses/s", "cores": cores, "kind": "port",
            "sample": "%d random grid points of the same G=1e5/P=5e4 problem, per-point python loop "
                      "(localize over all P obs -> mask/scale -> torch fp64 eigh weights -> transform), "
                      "1 thread/process" % n_points_total}


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
    ap.add_argument("--grid-per-gpu", type=int, default=G_PER_GPU)
    ap.add_argument("--pipeline-depth", type=int, default=int(os.environ.get("MIA_PIPELINE_DEPTH", "3")), choices=[1, 2, 3, 4],
                    help="steps in flight (ShardedLetkf.submit): 3 (default) = steps i+1 and i+2 are enqueued before step i "
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
                          max_in_flight=max(2, args.pipeline_depth))

    def step():
        return runner.assimilate(X, grid_x, obs_x, Yb, d)

    def run(n_steps, depth):
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
                out = step()
            else:
                pend.append(runner.submit(X, grid_x, obs_x, Yb, d))
                if len(pend) == depth:
                    out = pend.popleft().result()
        while pend:
            out = pend.popleft().result()
        return out

    def timed(n_steps, depth):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(n_steps, depth)
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
    run(max(args.warmup, 50), depth)
    runner.kernel_timings.clear()
    elapsed, out = timed(args.steps, depth)
    loop_kernel_ms = runner.kernel_ms()      # the dominant kernel inside the timed loop (every 4th step)
    n_timed = len(runner.kernel_timings)
    serial_ms = None
    if depth != 1:           # secondary figure: the unpipelined step (latency of one step incl. its read-back)
        n_ser = max(10, args.steps // 8)
        el1, _ = timed(n_ser, 1)
        serial_ms = 1e3 * el1 / n_ser
    assert out.shape == (1, K_ENS, G) and bool(torch.isfinite(out).all())
    assert runner.last_flags_ok(), "kernel flagged grid points"

    # ---- dominant kernel alone, HIP events on the launch stream (torch's current stream)
    alone_ms, stage_ms = runner.time_stages(X, grid_x, obs_x, Yb, d, reps=max(5, min(args.steps, 20)))
    kern_ms = loop_kernel_ms if loop_kernel_ms else alone_ms
    p_max = runner.last_p_max
    flops = algorithmic_flops(K_ENS, 20, 1) * gpg
    achieved = flops / (kern_ms * 1e-3) / 1e12
    hbm_alg = algorithmic_bytes(K_ENS, 1, P / G) * gpg / (kern_ms * 1e-3) / 1e9

    # secondary figure: the fused Jacobi-eigensolver route on the same shard (kernel only)
    eig_ms = None
    if args.method != "eig":
        r2 = ShardedLetkf(device, rank, world, radii=[GC_RADIUS], inf_factor=INF, method="eig")
        r2._engine = runner.engine
        eig_ms, _ = r2.time_stages(X, grid_x, obs_x, Yb, d, reps=5)
    deg = runner.mean_degree()

    if rank == 0:
        value = G * args.steps / elapsed
        line = {
            "metric": "local analyses/sec (LETKF, 40-member)", "value": value, "unit": "analyses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LETKF config %s: G=%d grid points (%d per GPU), k=%d members, P=%d obs, "
                                   "Gaspari-Cohn radius %g (<=%d local obs), inf %.1f, m=1"
                                   % ("2" if world == 1 else "3-style", G, gpg, K_ENS, P, GC_RADIUS, p_max, INF),
                       "parallelism": "grid-point block shard x%d%s" % (world, " + RCCL all-gather" if world > 1 else "")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": (traffic_from_profiles(world) or {}).get("hbm_bytes_fetch_doubled"),
                         "traffic_detail": traffic_from_profiles(world),
                         "kernel": runner.dominant_kernel_name, "kernel_ms": kern_ms,
                         "kernel_ms_source": ("HIP events recorded by the library on the analysis stream around the kernel of "
                                              "every 4th step of the timed loop (%d launches, mia_letkf_step_timing_events)" % n_timed)
                                             if loop_kernel_ms else "burst of 5 launches after the timed loop",
                         "kernel_ms_alone": alone_ms,
                         "algorithmic_flops_per_analysis": algorithmic_flops(K_ENS, 20, 1),
                         "hbm_algorithmic_GBs": hbm_alg, "hbm_frac": hbm_alg / PEAK_HBM_GBS,
                         "executed_flops_per_analysis": executed_flops(K_ENS, 20, 1, deg),
                         "executed_frac": executed_flops(K_ENS, 20, 1, deg) * gpg / (kern_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                         "note": "fp32 peak: vector = f32-MFMA = 157.3 TFLOP/s; `achieved`/`frac` credit the SURVEY 8(d) "
                                 "flop count of the reference algorithm (9k^3 symmetric-QR eigensolve) whatever the method; "
                                 "the matfun route executes far fewer flops (executed_*: 16x16x4 MFMA Gram tiles incl. "
                                 "padding + Chebyshev recurrence of the measured mean degree), so frac can exceed 1"},
            "route": {"method": args.method, "mean_chebyshev_degree": deg, "declined_points_last_step": runner.last_retries,
                      "eigensolver_route_kernel_ms": eig_ms,
                      "eigensolver_route_kernel_analyses_per_s": (gpg / (eig_ms * 1e-3)) if eig_ms else None},
            "pipeline": {"depth": depth, "serial_ms_per_step": serial_ms,
                         "note": "depth d > 1: consecutive (independent) steps are software-pipelined over d slots / HIP "
                                 "streams; every step is fully computed, exchanged and validated inside the timed "
                                 "region.  serial_ms_per_step: the same step run one at a time (its latency)"},
            "stages_ms": stage_ms,
            "step_driver": ("native: one C call per step (mia_letkf_sharded_step_streams_f32); %d native steps in this process "
                            "(warm-up, timed loop, serial comparison)" % runner.native_steps)
                           if runner.native_steps else "python (engine entries one by one)",
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Experiment builds (-DMIA_EXPERIMENTS): the C2 analysis kernel alone under MIA_TILE2_PRIO (wave priority by phase) and
MIA_TILE2_TRIM (steps taken off the table's Chebyshev degree): kernel time over 40 launches, error against the oracle at 64 points.
   MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS python tools/tile2_knobs.py"""
import os, sys, subprocess
if len(sys.argv) == 1:
    envs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in os.environ.get("KNOBS", "").split(";")] if os.environ.get("KNOBS") else \
        [{}, {"MIA_TILE2_PRIO": "1"}, {"MIA_TILE2_PRIO": "2"}, {"MIA_TILE2_TRIM": "1"}, {"MIA_TILE2_TRIM": "2"}, {"MIA_TILE2_TRIM": "3"}, {}]
    for env in envs:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "run"], env=e, capture_output=True, text=True)
        print(env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
        if out.returncode != 0 or "core dump" in (out.stdout + out.stderr):
            sys.exit("a run failed: stopping (no further GPU step after a fault)")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
rec = bench.tile_route_case(eng, X, gx, ox, Yb, d, 10.0, 1.1, burst=40, n_check=64)
print("kernel %.4f ms  degree %.2f  error %.2e  declined %d" % (rec["kernel_ms"], rec["mean_chebyshev_degree"], rec["rel_frobenius_error_vs_oracle"], rec["declined_points"]))

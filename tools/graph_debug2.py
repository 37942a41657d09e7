import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
dev = torch.device("cuda:0")
G = int(sys.argv[1]); stage = sys.argv[2]
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
outs = []
for i in range(4):
    o = r.assimilate(X, gx, ox, Yb, d); torch.cuda.synchronize(); print("step", i, "replays", r.graph_replays, float(o.abs().max()), flush=True)
    outs.append(o.clone())
print("equal", torch.equal(outs[0], outs[-1]), flush=True)
if stage == "steps": sys.exit(0)
print("flags ok", r.last_flags_ok(), flush=True)
km, st = r.time_stages(X, gx, ox, Yb, d, reps=3); torch.cuda.synchronize(); print("time_stages", st, flush=True)
if stage == "stages": sys.exit(0)
r2 = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, method="eig"); r2._engine = r.engine
print(r2.time_stages(X, gx, ox, Yb, d, reps=2)); torch.cuda.synchronize()
print("deg", r.mean_degree())

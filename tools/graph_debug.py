import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(4096, 40, 2, dev)
nb0 = eng.localize(gx, ox, [10.0]); ref = eng.analysis(X, Yb, d, nb0, 1.1)
torch.cuda.synchronize()
which = sys.argv[1]
g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        if which == "loc":
            nb = eng.localize(gx, ox, [10.0], assume_p_max=20)
        elif which == "pack":
            rec = eng.pack_obs(Yb, d, torch.float32)
        elif which == "ana":
            out = eng.analysis(X, Yb, d, nb0, 1.1, defer_retry=True)
        elif which == "fill":
            z = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
print("captured", which, flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", i, "ok", flush=True)
if which == "loc":
    print(nb.stats.tolist(), torch.equal(nb.cnt, nb0.cnt), torch.equal(nb.idx[:, :20], nb0.idx[:, :20]))
if which == "ana":
    print(torch.equal(out[0], ref))

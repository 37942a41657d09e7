#!/usr/bin/env python3
"""Cost of the chunked compute / all-gather overlap route on ONE GPU (single-rank RCCL group: the collective
degenerates to a device copy, so this measures the chunking overhead only -- launch gaps of the smaller
analysis kernels, the per-chunk events, the final permute copy)."""
import os, socket, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from bench import make_case, K_ENS, OBS_STRIDE, GC_RADIUS, INF, G_PER_GPU
from torch_assimilate_amd.sharded import ShardedLetkf

dev = torch.device("cuda:0")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
G = G_PER_GPU
args = make_case(G, K_ENS, OBS_STRIDE, dev)

def run(fn, steps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps

plain = ShardedLetkf(dev, 0, 1, radii=[GC_RADIUS], inf_factor=INF, native_step=False)
print("python plain     %.4f ms/step" % run(lambda: plain.assimilate(*args)))
nat = ShardedLetkf(dev, 0, 1, radii=[GC_RADIUS], inf_factor=INF)
print("native plain     %.4f ms/step" % run(lambda: nat.assimilate(*args)))
ref = nat.assimilate(*args)
for C in (1, 2, 4, 8):
    r = ShardedLetkf(dev, 0, 1, radii=[GC_RADIUS], inf_factor=INF, comm_chunks=C)
    r._force_comm = True
    t = run(lambda: r.assimilate(*args))
    ok = torch.equal(r.assimilate(*args), ref)
    print("native RCCL C=%d  %.4f ms/step (1-rank communicator, equal=%s, native steps %d)" % (C, t, ok, r.native_steps))
    r.close()
for C in (1, 2, 4, 8):
    r = ShardedLetkf(dev, 0, 1, radii=[GC_RADIUS], inf_factor=INF, comm_chunks=C, native_step=False)
    print("overlapped C=%d   %.4f ms/step" % (C, run(lambda: r._assimilate_overlapped(*args, G, 0, G))))
dist.destroy_process_group()

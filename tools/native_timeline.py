#!/usr/bin/env python3
"""A few steps of the native step driver through a one-rank RCCL communicator, for `rocprofv3 --kernel-trace`:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 tools/native_timeline.py 4"""
import os, socket, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from bench import make_case, K_ENS, OBS_STRIDE, GC_RADIUS, INF, G_PER_GPU
from torch_assimilate_amd.sharded import ShardedLetkf

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
args = make_case(G_PER_GPU, K_ENS, OBS_STRIDE, dev)
r = ShardedLetkf(dev, 0, 1, radii=[GC_RADIUS], inf_factor=INF, comm_chunks=C)
r._force_comm = C > 0
for _ in range(steps):
    r.assimilate(*args)
torch.cuda.synchronize()
r.close()
dist.destroy_process_group()

#!/usr/bin/env python3
"""What the packets between two kernels of one HIP stream cost on this stack (MI355X, ROCm 7.2): a stream of ~80 us kernels
with (a) nothing between them, (b) an event record, (c) an event record + a wait for an (already complete) event of another
stream, (d) as (c) plus a 32-byte device-to-host copy on a second stream behind the recorded event -- the packet pattern of
the pipelined step (csrc/sharded_step.hip).  Prints microseconds per iteration; (a) is the kernel itself."""
import time
import torch

dev = torch.device("cuda:0")
n = 1536
a = torch.randn(n, n, device=dev)
b = torch.randn(n, n, device=dev)
out = torch.empty(n, n, device=dev)
main, side, prep = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
host = torch.empty(8, dtype=torch.int32).pin_memory()
small = torch.zeros(8, dtype=torch.int32, device=dev)
tiny = torch.zeros(64, device=dev)
ITERS = 400


def run(mode):
    evs = [torch.cuda.Event() for _ in range(16)]
    pes = [torch.cuda.Event() for _ in range(16)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(ITERS):
        if mode in ("prepwait", "full"):
            with torch.cuda.stream(prep):
                tiny.add_(1.0)
                pes[i & 15].record(prep)
            main.wait_event(pes[i & 15])
        with torch.cuda.stream(main):
            torch.mm(a, b, out=out)
        if mode in ("record", "recwait", "copy", "full"):
            evs[i & 15].record(main)
        if mode in ("recwait", "copy", "full"):
            side.wait_event(evs[i & 15])
        if mode in ("copy", "full"):
            with torch.cuda.stream(side):
                host.copy_(small, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ITERS * 1e6


for mode in ("plain", "record", "recwait", "copy", "prepwait", "full", "plain"):
    run(mode)
    print("%-9s %7.1f us per iteration" % (mode, min(run(mode) for _ in range(3))), flush=True)

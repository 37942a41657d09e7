"""mia_apply_local_weights_f32 (csrc/apply_local.hip) over the number of state rows: time per call, effective bandwidth."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G, k = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 40
gen = torch.Generator(device=dev); gen.manual_seed(1)
W = torch.randn((G, k, k), generator=gen, device=dev) / k ** 0.5
for m in [int(a) for a in sys.argv[3:]] or [1, 4, 16, 32, 64, 96, 112, 128, 160, 256]:
    X = torch.randn((m, k, G), generator=gen, device=dev)
    eng.apply_local_weights(X, W); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = eng.apply_local_weights(X, W)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ref = torch.einsum("mig,gij->mjg", X[:2, :, :2000] - X[:2, :, :2000].mean(dim=1, keepdim=True), W[:2000]) + X[:2, :, :2000].mean(dim=1, keepdim=True)
    err = float(torch.linalg.norm(out[:2, :, :2000] - ref) / torch.linalg.norm(ref))
    bytes_ = 4.0 * (G * k * k + 2.0 * m * k * G)
    print("m = %3d: %.3f ms, %.0f GB/s of compulsory traffic (W once + x in + xa out), error %.1e" % (m, ms, bytes_ / ms / 1e6, err), flush=True)
    # ... and with ONE weight matrix for all points (the global ETKF's transform)
    eng.apply_weights(X, W[0]); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        outg = eng.apply_weights(X, W[0])
    e1.record(); torch.cuda.synchronize()
    msg = e0.elapsed_time(e1) / 5
    refg = torch.einsum("mig,ij->mjg", X[:2, :, :2000] - X[:2, :, :2000].mean(dim=1, keepdim=True), W[0]) + X[:2, :, :2000].mean(dim=1, keepdim=True)
    errg = float(torch.linalg.norm(outg[:2, :, :2000] - refg) / torch.linalg.norm(refg))
    print("         one W for all points: %.3f ms, %.0f GB/s (x in + xa out), error %.1e" % (msg, 8.0 * m * k * G / msg / 1e6, errg), flush=True)
    del outg
    del X, out

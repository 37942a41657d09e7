#!/bin/bash
# builds and runs tools/mfma_rate.hip on the GPU box (plain HIP executable, no torch)
set -e
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_rate tools/mfma_rate.hip 2>/dev/null
timeout -k 10 120 /tmp/mfma_rate | tee gpurun_out/mfma_rate.txt

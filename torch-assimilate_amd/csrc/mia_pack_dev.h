// Observation records: [k][P] perturbations (+ d[P]) -> obs-major [P][kp], kp = round_up(k + 1, 4)
// (what the per-point mask-gather `arg[..., luse]`, pytassim/localization/wrapper.py:94-97, reads).
#pragma once
#include "mia_common.h"

namespace mia {

// One workgroup (any multiple of 32 threads) transposes 32 observations (all kp entries) through a 32 x 33 LDS tile.
template <typename T>
__device__ inline void pack_obs_tile(const T* __restrict__ Yb, const T* __restrict__ d, int k, int64_t P, int kp,
                                     T* __restrict__ rec, int64_t block, T (*tile)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ny = blockDim.x >> 5;  // 32 x ny
  const int64_t j0 = block * 32;
  for (int i0 = 0; i0 < kp; i0 += 32) {
    for (int r = ty; r < 32; r += ny) {
      const int i = i0 + r; const int64_t j = j0 + tx;
      T v = T(0);
      if (j < P) { if (i < k) v = Yb[(int64_t)i * P + j]; else if (i == k) v = d[j]; }
      tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += ny) {
      const int64_t j = j0 + r; const int i = i0 + tx;
      if (j < P && i < kp) rec[j * kp + i] = tile[tx][r];
    }
    __syncthreads();
  }
}

// float32 packing job that rides along with the first index-build kernel (independent work, one launch less)
struct PackJob {
  const float* Yb; const float* d; float* rec; int k; int kp;
};

}  // namespace mia

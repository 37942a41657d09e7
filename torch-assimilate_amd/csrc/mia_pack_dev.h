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

// One WAVEFRONT packs 64 observations.  Lane j reads entry j of every row -- 256-byte row segments, sixteen rows requested
// before any is consumed -- into an LDS image [64][kp + 1] (odd stride: no bank conflicts either way); the 64 records are
// then ONE contiguous piece of the output, written with 16-byte stores.  (pack_obs_tile moves 32 observations per trip of
// two barrier-separated phases with two loads in flight per lane; beside a bulk kernel it took ~40 us for 17 MB.)
// lds: 64 * (kp + 1) elements.  Single-wave workgroups only.
template <typename T>
__device__ inline void pack_obs_wave(const T* __restrict__ Yb, const T* __restrict__ d, int k, int64_t P, int kp,
                                     T* __restrict__ rec, int64_t block, T* lds) {
  const int lane = threadIdx.x & 63;
  const int64_t j0 = block * 64;
  const int64_t jc = j0 + lane < P ? j0 + lane : P - 1;
  const int ls = kp + 1;
  for (int i0 = 0; i0 < kp; i0 += 16) {
    T v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u;
      const T* src = i < k ? Yb + (int64_t)i * P + jc : d + jc;        // (rows beyond the innovation: read d, store 0)
      v[u] = *src;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u;
      if (i < kp) lds[lane * ls + i] = i <= k ? v[u] : T(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int nvalid = P - j0 < 64 ? (int)(P - j0) : 64;
  const int nq = nvalid * (kp >> 2);                      // 4-element pieces of the block's records (kp is a multiple of 4)
  T* out = rec + j0 * kp;
  int r = (lane * 4) / kp, c = lane * 4 - r * kp;
  const int dr = 256 / kp, dc = 256 - dr * kp;           // 64 lanes x 4 elements further on
  for (int q = lane; q < nq; q += 64) {
    const T* s_ = lds + r * ls + c;
    const T a0 = s_[0], a1 = s_[1], a2 = s_[2], a3 = s_[3];
    typedef T v4t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<v4t*>(out + (int64_t)q * 4) = v4t{a0, a1, a2, a3};      // (records start 16-byte aligned, kp % 4 = 0)
    r += dr; c += dc;
    if (c >= kp) { c -= kp; ++r; }
  }
}

// float32 packing job that rides along with the first index-build kernel (independent work, one launch less)
struct PackJob {
  const float* Yb; const float* d; float* rec; int k; int kp;
};

}  // namespace mia

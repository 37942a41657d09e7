// Shared helpers for the gfx950 LETKF kernels (device + host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mia_letkf.h"

#define MIA_HIP_TRY(expr)                      \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return (int)_e;      \
  } while (0)

#define MIA_LAUNCH_CHECK()                     \
  do {                                         \
    hipError_t _e = hipGetLastError();         \
    if (_e != hipSuccess) return (int)_e;      \
  } while (0)

namespace mia {

constexpr int kWave = 64;  // CDNA4 wavefront

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Gaspari-Cohn 5th-order taper, compact support r < 2 (Gaspari & Cohn 1999, eq. 4.10).
// Branch structure of pytassim/localization/gaspari_cohn.py:127-133 (strict `<`,
// NaN -> 0).  Horner form of _f1 (:78-84) and _f2 (:87-95).
template <typename T>
__host__ __device__ inline T gc_taper(T r) {
  if (r < T(1)) {
    return (((T(-0.25) * r + T(0.5)) * r + T(0.625)) * r - T(5.0 / 3.0)) * r * r + T(1);
  }
  if (r < T(2)) {
    return ((((r * T(1.0 / 12.0) - T(0.5)) * r + T(0.625)) * r + T(5.0 / 3.0)) * r - T(5)) * r +
           T(4) - T(2.0 / 3.0) / r;
  }
  return T(0);
}

// 64-lane sum / max via DPP-friendly butterfly (result in every lane)
template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace mia

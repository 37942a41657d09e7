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

// Full-wave reductions on the DPP path (cross-lane moves inside the VALU, ~10 instructions, no LDS crossbar
// round trips): __shfl_xor compiles to ds_bpermute_b32, whose ~100-cycle latency six times in a row was a
// visible share of the small per-point kernels.  Result is wave-uniform (broadcast through readlane).
__device__ inline float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ inline float wave_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ inline float wave_max_dpp(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

}  // namespace mia

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

// largest dynamic LDS request a launch may make: 160 KB per CU on gfx950, less what the kernels declare statically
// (a request of 159.6 KB passed a plain 160 KB check and then failed at launch with hipErrorInvalidValue)
// Wave priority of the preparation kernels (observation index, neighbour / tile lists): short chains of dependent memory round
// trips that run BESIDE an earlier step's analysis kernel.  0 = leave the arbiter alone (A/B: tools/ab_prio.sh)
#ifndef MIA_PREP_PRIO
#define MIA_PREP_PRIO 0
#endif
#define MIA_PREP_PRIORITY() __builtin_amdgcn_s_setprio(MIA_PREP_PRIO)
constexpr size_t kMaxDynamicLds = 160 * 1024 - 1024;

// device copy of a kernel expression (mia_kernel_op_t program), passed by value in the launch parameters
template <typename T>
struct KernelProgram { int n; unsigned char op[MIA_KERNEL_MAX_OPS]; T val[MIA_KERNEL_MAX_OPS]; };

// a well-formed program: known opcodes, operand stack never below the arity of an operator, never deeper than
// MIA_KERNEL_MAX_DEPTH, exactly one value left at the end
static inline int kernel_program_check(const mia_kernel_op_t* prog, int n_ops) {
  if (!prog) return MIA_ERR_NULL;
  if (n_ops < 1 || n_ops > MIA_KERNEL_MAX_OPS) return MIA_ERR_SIZE;
  int depth = 0;
  for (int i = 0; i < n_ops; ++i) {
    const int op = prog[i].op;
    if (op < MIA_KOP_DOT || op > MIA_KOP_SIN) return MIA_ERR_SIZE;
    if (op <= MIA_KOP_DIAG) { if (++depth > MIA_KERNEL_MAX_DEPTH) return MIA_ERR_UNSUPPORTED; }
    else if (op <= MIA_KOP_POW) { if (depth < 2) return MIA_ERR_SIZE; --depth; }
    else if (depth < 1) return MIA_ERR_SIZE;
  }
  return depth == 1 ? MIA_OK : MIA_ERR_SIZE;
}

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

// Gaspari-Cohn taper with form factor infinity, C_0(z, inf, c) (Gaspari & Cohn 1999), compact support r < 2:
// pytassim/localization/gaspari_cohn.py:139-254 (GaspariCohnInf).  The reference assigns _f4, _f3, _f2, _f1 in this
// order under the strict conditions r < 2, 1.5, 1, 0.5 (:244-251), i.e. the LAST matching branch wins; NaN -> 0.
// rinv = 1 / r (only used for r >= 0.5).  Horner forms of _f1 (:176-183), _f2 (:185-193), _f3 (:195-204), _f4 (:206-215).
template <typename T>
__host__ __device__ inline T gc_inf_taper_rinv(T r, T rinv) {
  if (r < T(0.5)) return (((T(-28.0 / 33.0) * r + T(8.0 / 11.0)) * r + T(20.0 / 11.0)) * r - T(80.0 / 33.0)) * r * r + T(1);
  if (r < T(1))
    return ((((T(20.0 / 33.0) * r - T(16.0 / 11.0)) * r * r + T(100.0 / 33.0)) * r - T(45.0 / 11.0)) * r) + T(51.0 / 22.0) -
           T(7.0 / 44.0) * rinv;
  if (r < T(1.5))
    return (((((T(-4.0 / 11.0) * r + T(16.0 / 11.0)) * r - T(10.0 / 11.0)) * r - T(100.0 / 33.0)) * r + T(5)) * r) -
           T(61.0 / 22.0) + T(115.0 / 132.0) * rinv;
  if (r < T(2))
    return (((((T(4.0 / 33.0) * r - T(8.0 / 11.0)) * r + T(10.0 / 11.0)) * r + T(80.0 / 33.0)) * r - T(80.0 / 11.0)) * r) +
           T(64.0 / 11.0) - T(32.0 / 33.0) * rinv;
  return T(0);
}
template <typename T>
__host__ __device__ inline T gc_inf_taper(T r) { return gc_inf_taper_rinv<T>(r, r < T(0.5) ? T(0) : T(1) / r); }

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
// maximum over the wave of values that are >= +0 or NaN: their bit patterns order like the values (NaN above all), so
// the integer maximum folds into the DPP instruction (fmaxf adds a canonicalising v_max per step)
__device__ inline float wave_max_nonneg_dpp(float v) {
  unsigned u = __float_as_uint(v);
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false); u = u > t ? u : t;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)u, 0), b = (unsigned)__builtin_amdgcn_readlane((int)u, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)u, 32), d = (unsigned)__builtin_amdgcn_readlane((int)u, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return __uint_as_float(ab > cd ? ab : cd);
}

}  // namespace mia

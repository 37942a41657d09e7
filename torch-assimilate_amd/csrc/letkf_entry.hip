// C entry points of the per-grid-point analysis and their dispatch to the eigensolver kernels (letkf_sys.hip: float32,
// order <= 64, systolic Jacobi; letkf_wave.hip: runtime order, float32 / float64) and the matrix-function kernels
// (letkf_cheb.hip, letkf_tile.hip); observation-record packing.  (The first-generation LDS-resident kernel that used to
// live here -- 11.8 ms per 1e5 analyses at C2, kept for A/B runs through round 1 -- is gone.)
//
// What every route computes per grid point:
//   gather + sqrt(rho)-scale of the local obs columns      interface/wrapper.py:91-97
//   Gram matrix                                            core/utils.py:172 via etkf.py:68
//   symmetric eigensolve, clamp >= 0, +(k-1)/inf           core/utils.py:57-60, etkf.py:67
//   w_mean = Pa (Yb d^T), W = V diag(sqrt((k-1)/l)) V^T    core/etkf.py:70-76
//   xa = mean + X' (w_mean 1^T + W)                        interface/base.py:257-278
// primal route (p_max > k): the k x k matrix C = Yl Yl^T, the reference's operation sequence; dual route (p_max <= k): the
// p x p matrix S = Yl^T Yl, every function of A = C + reg I applied through
//   f(A) = f(reg) I + Yl V diag((f(l+reg) - f(reg)) / l) V^T Yl^T
// with cancellation-free divided differences; p == 0 collapses to the reference's prior branch sqrt(inf) I (etkf.py:91-95).
#include "mia_common.h"
#include "mia_jacobi.h"
#include "mia_localize_dev.h"
#include "mia_kernels.h"
#include "mia_pack_dev.h"

#include <cstdlib>

namespace mia {

template <typename T>
int wave_analysis_launch(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const T* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap,
                         int p_max, T inf_factor, int kernel_mode, T gamma, T* Xa, int64_t ldo, int64_t o0,
                         T* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream,
                         const mia_kernel_op_t* prog = nullptr, int n_ops = 0);

int wave_analysis_launch_f32_in_f64(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                                    const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                    float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                    float* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream);

int sys_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                        const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                        float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                        float* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream);

// [k][P] (+ d[P]) -> obs-major records [P][kp]: one observation's k perturbations and its
// innovation become one contiguous, coalescable 4*kp-byte record for the gather
template <typename T>
__global__ __launch_bounds__(256) void pack_obs_kernel(const T* Yb, const T* d, int k, int64_t P, int kp, T* rec) {
  __shared__ T tile[32][33];
  pack_obs_tile<T>(Yb, d, k, P, kp, rec, (int64_t)blockIdx.x, tile);
}

template <typename T>
static int pack_impl(const T* Yb, const T* d, int k, int64_t P, T* rec, hipStream_t stream) {
  if (k < 2 || P < 0) return MIA_ERR_SIZE;
  if (P == 0) return MIA_OK;
  if (!Yb || !d || !rec) return MIA_ERR_NULL;
  if (((uintptr_t)rec) & 15) return MIA_ERR_ALIGN;
  const int kp = (k + 1 + 3) & ~3;
  if ((P + 31) / 32 > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  pack_obs_kernel<T><<<dim3((unsigned)((P + 31) / 32)), dim3(256), 0, stream>>>(Yb, d, k, P, kp, rec);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <typename T>
static int analysis_packed_impl(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1, const T* rec,
                                int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                int p_cap, int p_max, T inf_factor, int kernel_mode, T gamma, T* Xa, int64_t ldo,
                                int64_t o0, T* W_opt, int32_t* flags_opt, hipStream_t stream, int only_flagged = 0,
                                const mia_kernel_op_t* prog = nullptr, int n_ops = 0) {
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2 || P < 0 || p_cap < 1 || p_max < 0) return MIA_ERR_SIZE;
  if (!(inf_factor > T(0))) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!X || !Xa || !nbr_cnt || !nbr_idx || !nbr_w) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  if (p_max > p_cap) p_max = p_cap;
  if (P > 0 && !rec) return MIA_ERR_NULL;
  const int kp = (k + 1 + 3) & ~3;
  if (kernel_mode == 2)   // kernel expression: runtime-order kernel only
    return wave_analysis_launch<T>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                   kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream, prog, n_ops);
  if constexpr (sizeof(T) == 4) {
    // redo of declined points (only_flagged): float64 arithmetic on the float32 data where the block fits the LDS -- a spectrum
    // too wide for the matrix-function route is also what a float32 eigensolver resolves worst (tools/stress_tile.py)
    if (only_flagged && kernel_mode != 2) {
      const int rc = wave_analysis_launch_f32_in_f64(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                                     kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream);
      if (rc != MIA_ERR_UNSUPPORTED) return rc;
    }
    // float32, order <= 64: the systolic-Jacobi kernel (letkf_sys.hip)
    const int rc = sys_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                       kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream);
    if (rc != MIA_ERR_UNSUPPORTED) return rc;
  }
  // runtime-order kernel (letkf_wave.hip): float64 and orders > 64
  return wave_analysis_launch<T>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                 kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream);
}

template <typename T>
static int analysis_impl(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1, const T* Yb,
                         const T* d, int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx,
                         const double* nbr_w, int p_cap, int p_max, T inf_factor, int kernel_mode, T gamma,
                         T* Xa, int64_t ldo, int64_t o0, T* W_opt, int32_t* flags_opt, void* ws,
                         size_t ws_bytes, hipStream_t stream) {
  if (k < 2 || P < 0) return MIA_ERR_SIZE;
  if (P > 0 && (!Yb || !d || !ws)) return MIA_ERR_NULL;
  size_t need = 0;
  int rc = mia_letkf_analysis_workspace_bytes(k, P, (int)sizeof(T), &need);
  if (rc != MIA_OK) return rc;
  if (ws_bytes < need) return MIA_ERR_WORKSPACE;
  if (P > 0 && (((uintptr_t)ws) & 255)) return MIA_ERR_ALIGN;
  rc = pack_impl<T>(Yb, d, k, P, (T*)ws, stream);
  if (rc != MIA_OK) return rc;
  return analysis_packed_impl<T>(X, ldx, m, k, g0, g1, (const T*)ws, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                                 inf_factor, kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, stream);
}

}  // namespace mia

using namespace mia;

extern "C" int mia_letkf_analysis_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (k < 2 || P < 0 || (elem_bytes != 4 && elem_bytes != 8)) return MIA_ERR_SIZE;
  const size_t kp = (size_t)((k + 1 + 3) & ~3);
  *bytes = align_up((size_t)P * kp * (size_t)elem_bytes + 256, 256);
  return MIA_OK;
}

extern "C" int mia_letkf_analysis_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                      const float* Yb, const float* d, int64_t P, const int32_t* nbr_cnt,
                                      const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                      float inf_factor, float* Xa, int64_t ldo, int64_t o0, float* W_opt,
                                      int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return analysis_impl<float>(X, ldx, m, k, g0, g1, Yb, d, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                              inf_factor, 0, 0.0f, Xa, ldo, o0, W_opt, flags_opt, ws, ws_bytes,
                              (hipStream_t)stream);
}

extern "C" int mia_letkf_analysis_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                      const double* Yb, const double* d, int64_t P, const int32_t* nbr_cnt,
                                      const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                      double inf_factor, double* Xa, int64_t ldo, int64_t o0, double* W_opt,
                                      int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return analysis_impl<double>(X, ldx, m, k, g0, g1, Yb, d, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                               inf_factor, 0, 0.0, Xa, ldo, o0, W_opt, flags_opt, ws, ws_bytes,
                               (hipStream_t)stream);
}

extern "C" int mia_lketkf_rbf_analysis_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const float* Yb, const float* d, int64_t P, const int32_t* nbr_cnt,
                                           const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                           float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                           float* W_opt, int32_t* flags_opt, void* ws, size_t ws_bytes,
                                           void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  if (!(gamma > 0.0f)) return MIA_ERR_SIZE;
  return analysis_impl<float>(X, ldx, m, k, g0, g1, Yb, d, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                              inf_factor, 1, gamma, Xa, ldo, o0, W_opt, flags_opt, ws, ws_bytes,
                              (hipStream_t)stream);
}

extern "C" int mia_lketkf_rbf_analysis_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const double* Yb, const double* d, int64_t P, const int32_t* nbr_cnt,
                                           const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                           double inf_factor, double gamma, double* Xa, int64_t ldo, int64_t o0,
                                           double* W_opt, int32_t* flags_opt, void* ws, size_t ws_bytes,
                                           void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  if (!(gamma > 0.0)) return MIA_ERR_SIZE;
  return analysis_impl<double>(X, ldx, m, k, g0, g1, Yb, d, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                               inf_factor, 1, gamma, Xa, ldo, o0, W_opt, flags_opt, ws, ws_bytes,
                               (hipStream_t)stream);
}

extern "C" int mia_letkf_pack_obs_f32(const float* Yb, const float* d, int k, int64_t P, float* rec, void* stream) {
  (void)hipGetLastError();
  return pack_impl<float>(Yb, d, k, P, rec, (hipStream_t)stream);
}
extern "C" int mia_letkf_pack_obs_f64(const double* Yb, const double* d, int k, int64_t P, double* rec, void* stream) {
  (void)hipGetLastError();
  return pack_impl<double>(Yb, d, k, P, rec, (hipStream_t)stream);
}
extern "C" int mia_letkf_analysis_packed_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                             const float* rec, int64_t P, const int32_t* nbr_cnt,
                                             const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                             float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                             float* W_opt, int32_t* flags_opt, void* stream) {
  (void)hipGetLastError();
  return analysis_packed_impl<float>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                     gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, W_opt, flags_opt, (hipStream_t)stream);
}
extern "C" int mia_letkf_analysis_packed_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                             const double* rec, int64_t P, const int32_t* nbr_cnt,
                                             const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                             double inf_factor, double gamma, double* Xa, int64_t ldo, int64_t o0,
                                             double* W_opt, int32_t* flags_opt, void* stream) {
  (void)hipGetLastError();
  return analysis_packed_impl<double>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                      gamma > 0.0 ? 1 : 0, gamma, Xa, ldo, o0, W_opt, flags_opt, (hipStream_t)stream);
}

extern "C" int mia_lketkf_kernel_analysis_packed_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                                     const float* rec, int64_t P, const int32_t* nbr_cnt,
                                                     const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                                     float inf_factor, const mia_kernel_op_t* prog, int n_ops,
                                                     float* Xa, int64_t ldo, int64_t o0, float* W_opt,
                                                     int32_t* flags_opt, void* stream) {
  (void)hipGetLastError();
  const int rc = kernel_program_check(prog, n_ops);
  if (rc != MIA_OK) return rc;
  return analysis_packed_impl<float>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                     2, 0.0f, Xa, ldo, o0, W_opt, flags_opt, (hipStream_t)stream, 0, prog, n_ops);
}
extern "C" int mia_lketkf_kernel_analysis_packed_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                                     const double* rec, int64_t P, const int32_t* nbr_cnt,
                                                     const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                                     double inf_factor, const mia_kernel_op_t* prog, int n_ops,
                                                     double* Xa, int64_t ldo, int64_t o0, double* W_opt,
                                                     int32_t* flags_opt, void* stream) {
  (void)hipGetLastError();
  const int rc = kernel_program_check(prog, n_ops);
  if (rc != MIA_OK) return rc;
  return analysis_packed_impl<double>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                      2, 0.0, Xa, ldo, o0, W_opt, flags_opt, (hipStream_t)stream, 0, prog, n_ops);
}

extern "C" int mia_letkf_analysis_matfun_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                             const float* rec, int64_t P, const int32_t* nbr_cnt,
                                             const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                             float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                             int32_t* flags, int32_t* retry_count, void* stream) {
  (void)hipGetLastError();
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2 || P < 0 || p_cap < 1 || p_max < 0) return MIA_ERR_SIZE;
  if (!(inf_factor > 0.0f)) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!X || !Xa || !nbr_cnt || !nbr_idx || !nbr_w || !flags || !retry_count) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  if (p_max > p_cap) p_max = p_cap;
  if (P > 0 && !rec) return MIA_ERR_NULL;
  return cheb_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                              gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, flags, retry_count, nullptr, nullptr,
                              (hipStream_t)stream);
}
extern "C" int mia_letkf_weights_matfun_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                            const float* rec, int64_t P, const int32_t* nbr_cnt,
                                            const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                            float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0, float* W,
                                            int32_t* flags, int32_t* retry_count, void* stream) {
  (void)hipGetLastError();
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2 || P < 0 || p_cap < 1 || p_max < 0) return MIA_ERR_SIZE;
  if (!(inf_factor > 0.0f)) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!X || !Xa || !W || !nbr_cnt || !nbr_idx || !nbr_w || !flags || !retry_count) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  if (p_max > p_cap) p_max = p_cap;
  if (P > 0 && !rec) return MIA_ERR_NULL;
  return cheb_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                              gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, flags, retry_count, nullptr, nullptr,
                              (hipStream_t)stream, 0, 0, nullptr, W);
}
extern "C" int mia_letkf_weights_retry_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const float* rec, int64_t P, const int32_t* nbr_cnt,
                                           const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                           float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0, float* W,
                                           int32_t* flags, void* stream) {
  (void)hipGetLastError();
  if (!flags || !W) return MIA_ERR_NULL;
  return analysis_packed_impl<float>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                     gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, W, flags, (hipStream_t)stream, 1);
}
extern "C" int mia_letkf_analysis_retry_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                            const float* rec, int64_t P, const int32_t* nbr_cnt,
                                            const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                            float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                            int32_t* flags, void* stream) {
  (void)hipGetLastError();
  if (!flags) return MIA_ERR_NULL;
  return analysis_packed_impl<float>(X, ldx, m, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                     gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, nullptr, flags, (hipStream_t)stream, 1);
}

extern "C" int mia_letkf_analysis_matfun_fused_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                                   const float* rec, int64_t P, const double* grid_xyz, int n_coord,
                                                   const int32_t* coord_group, const double* gc_c, int n_r,
                                                   double gc_eps, void* index_ws, size_t index_ws_bytes,
                                                   int p_max_assumed, float inf_factor, float gamma, float* Xa,
                                                   int64_t ldo, int64_t o0, int32_t* flags, int32_t* retry_count,
                                                   int32_t* stats, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2 || P < 1 || p_max_assumed < 0) return MIA_ERR_SIZE;
  if (!(inf_factor > 0.0f)) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (!stats) return MIA_ERR_NULL;
  MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (ng == 0) return MIA_OK;
  if (!X || !Xa || !rec || !flags || !retry_count || !index_ws) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  size_t need = 0;
  int rc = mia_letkf_localize_workspace_bytes(P, n_coord, &need);
  if (rc != MIA_OK) return rc;
  if (index_ws_bytes < need) return MIA_ERR_WORKSPACE;
  ScanParams sp;
  rc = make_scan_params(&sp, grid_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, index_ws);
  if (rc != MIA_OK) return rc;
  return cheb_analysis_launch(X, ldx, m, k, g0, ng, rec, nullptr, nullptr, nullptr, p_max_assumed > 0 ? p_max_assumed : 1,
                              p_max_assumed, inf_factor, gamma > 0.0f ? 1 : 0, gamma, Xa, ldo, o0, flags, retry_count,
                              &sp, stats, stream);
}

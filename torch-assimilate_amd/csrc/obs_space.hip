// Observation-space preparation on the device: the step immediately upstream of the local analysis
// (SURVEY.md 8f rank 1).  Reference: AssimilationInterface._get_obs_space_variables
// (pytassim/interface/base.py:359-379): per observation subset, ensemble mean / perturbations of H(x)
// (split_mean_perts), innovation y - mean, both multiplied by R^-1/2 (Observation.mul_rcinv,
// pytassim/observation.py:241-295), subsets stacked along the observation axis (_stack_obs).
//
//  * uncorrelated R (observation.py:241-245, 276-278): value * (1 / sqrt(var)).  One fused kernel that can
//    also emit the packed obs-major records the analysis kernels gather from (no separate pack pass).
//  * correlated R (observation.py:247-275): value @ inv(cholesky(R).T), i.e. out = V L^-T with R = L L^T.
//    The reference inverts the factor; here the k+1 value rows are appended below R and the blocked
//    right-looking Cholesky sweeps over the augmented matrix: the panel solve that produces L[i, blk] for a
//    row i of R produces (V L^-T)[r, blk] for an appended row r, and the trailing update is the same rank-NB
//    update for both.  No inverse is formed.
#include "mia_common.h"
#include "mia_pack_dev.h"

namespace mia {

// one workgroup = 32 observations x all members; means through an 8-way split of the member loop
template <typename T>
__global__ __launch_bounds__(256) void obs_space_uncorr_kernel(const T* __restrict__ hx, int64_t ldh,
                                                               const T* __restrict__ y, const T* __restrict__ var,
                                                               int k, int64_t P, int kp, T* __restrict__ Yb, int64_t ldy,
                                                               T* __restrict__ d, T* __restrict__ rec) {
  __shared__ T tile[32][33];
  __shared__ T part[8][33];
  __shared__ T mean_s[32], rinv_s[32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t j0 = (int64_t)blockIdx.x * 32, j = j0 + tx;
  T s = T(0);
  if (j < P) for (int i = ty; i < k; i += 8) s += hx[(int64_t)i * ldh + j];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
    T t = T(0);
    for (int q = 0; q < 8; ++q) t += part[q][tx];
    mean_s[tx] = t / T(k);
    rinv_s[tx] = j < P ? T(1) / sqrt(var[j]) : T(0);        // observation.py:243-244: 1 / sqrt(covariance)
  }
  __syncthreads();
  const T mean = mean_s[tx], rinv = rinv_s[tx];
  for (int i0 = 0; i0 < kp; i0 += 32) {
    for (int r = ty; r < 32; r += 8) {
      const int i = i0 + r;
      T v = T(0);
      if (j < P) {
        if (i < k) { v = (hx[(int64_t)i * ldh + j] - mean) * rinv; if (Yb) Yb[(int64_t)i * ldy + j] = v; }
        else if (i == k) { v = (y[j] - mean) * rinv; if (d) d[j] = v; }
      }
      tile[r][tx] = v;
    }
    __syncthreads();
    if (rec) {
      for (int r = ty; r < 32; r += 8) {
        const int64_t jj = j0 + r; const int ii = i0 + tx;
        if (jj < P && ii < kp) rec[jj * kp + ii] = tile[tx][r];
      }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const T* Yb, const T* d, int k, int64_t P, int kp, T* rec) {
  __shared__ T tile[32][33];
  pack_obs_tile<T>(Yb, d, k, P, kp, rec, (int64_t)blockIdx.x, tile);
}

// ---------------------------------------------------------------- correlated R
constexpr int NB = 32;

// rows P .. P+k of the augmented matrix: centred ensemble rows and the innovation row
template <typename T>
__global__ __launch_bounds__(256) void obs_space_center_kernel(const T* __restrict__ hx, int64_t ldh,
                                                               const T* __restrict__ y, int k, int64_t P,
                                                               T* __restrict__ V /* [k+1][P] */) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= P) return;
  T s = T(0);
  for (int i = 0; i < k; ++i) s += hx[(int64_t)i * ldh + j];
  const T mean = s / T(k);
  for (int i = 0; i < k; ++i) V[(int64_t)i * P + j] = hx[(int64_t)i * ldh + j] - mean;
  V[(int64_t)k * P + j] = y[j] - mean;
}

// unblocked Cholesky of the nb x nb diagonal block at (c0, c0), one wave, block kept in LDS.
// A non-positive pivot sets *info = c0 + j + 1 (LAPACK convention) and leaves NaN behind.
template <typename T>
__global__ __launch_bounds__(64) void chol_diag_kernel(T* __restrict__ M, int64_t ld, int64_t c0, int nb, int32_t* info) {
  __shared__ T a[NB][NB + 1];
  const int r = threadIdx.x;
  if (r < nb) for (int c = 0; c <= r; ++c) a[r][c] = M[(c0 + r) * ld + c0 + c];
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (r == j) {
      const T piv = a[j][j];
      if (!(piv > T(0)) && info) atomicCAS(info, 0, (int)(c0 + j + 1));
      a[j][j] = sqrt(piv);
    }
    __syncthreads();
    if (r > j && r < nb) a[r][j] /= a[j][j];
    __syncthreads();
    if (r > j && r < nb) for (int c = j + 1; c <= r; ++c) a[r][c] -= a[r][j] * a[c][j];
    __syncthreads();
  }
  if (r < nb) for (int c = 0; c < nb; ++c) M[(c0 + r) * ld + c0 + c] = c <= r ? a[r][c] : T(0);
}

// rows i >= c0 + nb (rows of R below the block and the appended value rows): x L_bb^T = M[i, blk]
template <typename T>
__global__ __launch_bounds__(256) void chol_panel_kernel(T* __restrict__ M, int64_t ld, int64_t c0, int nb, int64_t rows) {
  __shared__ T l[NB][NB + 1];
  for (int it = threadIdx.x; it < nb * nb; it += blockDim.x) {
    const int r = it / nb, c = it - r * nb;
    l[r][c] = M[(c0 + r) * ld + c0 + c];
  }
  __syncthreads();
  const int64_t i = c0 + nb + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows) return;
  T x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (c < nb) {
      T v = M[i * ld + c0 + c];
#pragma unroll
      for (int q = 0; q < NB; ++q) if (q < c) v -= x[q] * l[c][q];
      x[c] = v / l[c][c];
    }
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) if (c < nb) M[i * ld + c0 + c] = x[c];
}

// trailing update M[i][j] -= sum_q M[i][blk q] * M[j][blk q] for i, j >= c1 = c0 + nb, j < P; rows of R only
// need j <= i, the appended rows (i >= P) the whole row.  32 x 32 output tile per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void chol_update_kernel(T* __restrict__ M, int64_t ld, int64_t c0, int nb, int64_t P,
                                                          int64_t rows) {
  __shared__ T li[32][NB + 1], lj[32][NB + 1];
  const int64_t c1 = c0 + nb;
  const int64_t i0 = c1 + (int64_t)blockIdx.y * 32, j0 = c1 + (int64_t)blockIdx.x * 32;
  if (i0 + 31 < P && j0 > i0 + 31) return;               // tile of R rows only, entirely above the diagonal
  for (int it = threadIdx.x; it < 32 * nb; it += 256) {
    const int r = it / nb, q = it - r * nb;
    li[r][q] = (i0 + r < rows) ? M[(i0 + r) * ld + c0 + q] : T(0);
    lj[r][q] = (j0 + r < P) ? M[(j0 + r) * ld + c0 + q] : T(0);
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = i0 + r, j = j0 + tx;
    if (i >= rows || j >= P || (i < P && j > i)) continue;
    T acc = T(0);
    for (int q = 0; q < nb; ++q) acc += li[r][q] * lj[tx][q];
    M[i * ld + j] -= acc;
  }
}

template <typename T>
static int obs_space_corr_impl(const T* hx, int64_t ldh, const T* y, const T* cov, int k, int64_t P, T* Yb, int64_t ldy,
                               T* d, T* rec, int32_t* info, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (k < 1 || P < 0 || ldh < P || (Yb && ldy < P)) return MIA_ERR_SIZE;
  if (P == 0) return MIA_OK;
  if (!hx || !y || !cov || !ws) return MIA_ERR_NULL;
  if ((uintptr_t)ws % 256) return MIA_ERR_ALIGN;
  const int64_t rows = P + k + 1;
  if (ws_bytes < (size_t)rows * P * sizeof(T)) return MIA_ERR_WORKSPACE;
  if (P > 46000) return MIA_ERR_UNSUPPORTED;             // (grid.y of the update kernel, and P^2 storage)
  T* M = (T*)ws;
  if (info) MIA_HIP_TRY(hipMemsetAsync(info, 0, sizeof(int32_t), stream));
  MIA_HIP_TRY(hipMemcpyAsync(M, cov, (size_t)P * P * sizeof(T), hipMemcpyDeviceToDevice, stream));
  T* V = M + (size_t)P * P;
  obs_space_center_kernel<T><<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream>>>(hx, ldh, y, k, P, V);
  MIA_LAUNCH_CHECK();
  for (int64_t c0 = 0; c0 < P; c0 += NB) {
    const int nb = (int)(P - c0 < NB ? P - c0 : NB);
    chol_diag_kernel<T><<<dim3(1), dim3(64), 0, stream>>>(M, P, c0, nb, info);
    MIA_LAUNCH_CHECK();
    const int64_t below = rows - (c0 + nb);
    chol_panel_kernel<T><<<dim3((unsigned)((below + 255) / 256)), dim3(256), 0, stream>>>(M, P, c0, nb, rows);
    MIA_LAUNCH_CHECK();
    const int64_t tc = (P - (c0 + nb) + 31) / 32, tr = (below + 31) / 32;
    if (tc > 0) {
      chol_update_kernel<T><<<dim3((unsigned)tc, (unsigned)tr), dim3(256), 0, stream>>>(M, P, c0, nb, P, rows);
      MIA_LAUNCH_CHECK();
    }
  }
  if (Yb) MIA_HIP_TRY(hipMemcpy2DAsync(Yb, (size_t)ldy * sizeof(T), V, (size_t)P * sizeof(T), (size_t)P * sizeof(T), k,
                                       hipMemcpyDeviceToDevice, stream));
  if (d) MIA_HIP_TRY(hipMemcpyAsync(d, V + (size_t)k * P, (size_t)P * sizeof(T), hipMemcpyDeviceToDevice, stream));
  if (rec) {
    const int kp = (k + 1 + 3) & ~3;
    pack_rows_kernel<T><<<dim3((unsigned)((P + 31) / 32)), dim3(256), 0, stream>>>(V, V + (size_t)k * P, k, P, kp, rec);
    MIA_LAUNCH_CHECK();
  }
  return MIA_OK;
}

template <typename T>
static int obs_space_uncorr_impl(const T* hx, int64_t ldh, const T* y, const T* var, int k, int64_t P, T* Yb,
                                 int64_t ldy, T* d, T* rec, hipStream_t stream) {
  if (k < 1 || P < 0 || ldh < P || (Yb && ldy < P)) return MIA_ERR_SIZE;
  if (P == 0) return MIA_OK;
  if (!hx || !y || !var) return MIA_ERR_NULL;
  if (rec && ((uintptr_t)rec & 15)) return MIA_ERR_ALIGN;
  if ((P + 31) / 32 > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  const int kp = (k + 1 + 3) & ~3;
  obs_space_uncorr_kernel<T><<<dim3((unsigned)((P + 31) / 32)), dim3(256), 0, stream>>>(hx, ldh, y, var, k, P, kp, Yb, ldy,
                                                                                       d, rec);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_obs_space_uncorr_f32(const float* hx, int64_t ldh, const float* y, const float* var, int k, int64_t P,
                                        float* Yb_opt, int64_t ldy, float* d_opt, float* rec_opt, void* stream) {
  (void)hipGetLastError();
  return obs_space_uncorr_impl<float>(hx, ldh, y, var, k, P, Yb_opt, ldy, d_opt, rec_opt, (hipStream_t)stream);
}
extern "C" int mia_obs_space_uncorr_f64(const double* hx, int64_t ldh, const double* y, const double* var, int k, int64_t P,
                                        double* Yb_opt, int64_t ldy, double* d_opt, double* rec_opt, void* stream) {
  (void)hipGetLastError();
  return obs_space_uncorr_impl<double>(hx, ldh, y, var, k, P, Yb_opt, ldy, d_opt, rec_opt, (hipStream_t)stream);
}
extern "C" int mia_obs_space_corr_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (k < 1 || P < 0 || (elem_bytes != 4 && elem_bytes != 8)) return MIA_ERR_SIZE;
  *bytes = mia::align_up((size_t)(P + k + 1) * (size_t)(P > 0 ? P : 1) * elem_bytes, 256);
  return MIA_OK;
}
extern "C" int mia_obs_space_corr_f32(const float* hx, int64_t ldh, const float* y, const float* cov, int k, int64_t P,
                                      float* Yb_opt, int64_t ldy, float* d_opt, float* rec_opt, int32_t* info_opt,
                                      void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return obs_space_corr_impl<float>(hx, ldh, y, cov, k, P, Yb_opt, ldy, d_opt, rec_opt, info_opt, ws, ws_bytes,
                                    (hipStream_t)stream);
}
extern "C" int mia_obs_space_corr_f64(const double* hx, int64_t ldh, const double* y, const double* cov, int k, int64_t P,
                                      double* Yb_opt, int64_t ldy, double* d_opt, double* rec_opt, int32_t* info_opt,
                                      void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return obs_space_corr_impl<double>(hx, ldh, y, cov, k, P, Yb_opt, ldy, d_opt, rec_opt, info_opt, ws, ws_bytes,
                                     (hipStream_t)stream);
}

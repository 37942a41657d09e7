// Localised IEnKS weight update and the ensemble transform with per-grid-point weights, for gfx950.
//
// the other per-grid-point algorithm of the package.  One workgroup owns one grid
// point and performs IEnKSTransformModule / IEnKSBundleModule._update_weights (pytassim/core/ienks.py:108-126)
// on the localised block, i.e. what LocalizedIEnKSTransform.inner_loop (interface/lienks.py:75-118) runs once
// per grid point through wrapper_localization with args_to_skip=(0,) (the weights are not masked):
//
//   w_mean  = mean_j(W - I),  Wp = W - w_mean 1^T                       _split_weights      ienks.py:46-54
//   Wp^-1,  w_prec = (k-1) (Wp Wp^T)^-1 = (k-1) Wp^-T Wp^-1             _decompose_weights  ienks.py:56-67
//   D = Wp^-1 Yl   (transform)   |   D = Yl / eps   (bundle)            _get_dh_dw          ienks.py:69-75, 167-173
//   grad = (k-1) w_mean - D d_l^T                                       _get_gradient       ienks.py:77-87
//   Pn = (1-tau) w_prec + tau (D D^T + (k-1) I) = V L V^T               _update_covariance  ienks.py:89-103
//   w_cov = V L^-1 V^T,   Wp' = V ((k-1)/L)^1/2 V^T
//   W' = (w_mean - tau w_cov grad) 1^T + Wp'                            _update_weights / forward :121-141
//
// The reference decomposes with torch.svd twice (core/utils.py:122).  Here the inverse of the (general, square)
// Wp is an in-place Gauss-Jordan elimination with partial pivoting in LDS, and the SVD of the symmetric positive
// definite Pn is the symmetric Jacobi eigensolver of mia_jacobi_sym.h (U = V for such a matrix).  An empty local
// block returns the weights unchanged (ienks.py:135).
#include <cstdlib>
#include <type_traits>
#include "mia_common.h"
#include "mia_options.h"
#include "mia_jacobi.h"
#include "mia_jacobi_sym.h"
#include "mia_kernels.h"

namespace mia {

template <typename T>
struct IenksParams {
  const T* Win; int64_t w_stride;     // [n][k][k], or one [k][k] for all points when w_stride == 0
  int k; int64_t ng;
  const T* rec; int kp;
  const int32_t* cnt; const int32_t* idx; const double* w; int p_cap; int p_max;
  T tau, inv_eps;                      // inv_eps == 0: transform variant
  T* Wout; int32_t* flags;
  int n, lda, rows, need_inv, max_sweeps; T rot_tol2, stop_tol2;
  int nd;                              // > 0: dual route (tau == 1, p_max <= k): order of the p x p eigenproblem
  int only_flagged;                    // redo only the points the matfun route declined (MIA_FLAG_RETRY in flags)
};

// NT = 256 in production.  One-wave workgroups (NT = 64, MIA_IENKS_NARROW=1) were measured SLOWER for k = 40 (transform,
// tau = 1: 19.9 vs 14.8 ms per 1e5 grid points; tau < 1: 57 vs 39 ms): the update is bound by its LDS traffic (every
// operand of the elimination, the k x k products and the Jacobi comes from LDS), not by its barriers
template <typename T, int NT>
__global__ __launch_bounds__(NT) void ienks_update_kernel(IenksParams<T> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int k = P.k, kp = P.kp, n = P.n, lda = P.lda, pm = P.p_max;
  T* A = reinterpret_cast<T*>(smem_raw);     // [n][lda]  W -> Wp -> Wp^-1, later the eigenvectors V
  T* S = A + (size_t)n * lda;                // [n][lda]  Pn (canonical upper storage), later Wp'
  T* Yt = S + (size_t)n * lda;               // [rows][kp] obs-major local block: yb[0..k), d, pad
  const bool bundle = P.inv_eps != T(0);
  T* Dt = bundle ? Yt : Yt + (size_t)P.rows * kp;   // [rows][kp] obs-major D (scaled in place in the bundle variant)
  T* wm = Yt + (bundle ? 1 : 2) * (size_t)P.rows * kp;   // [n] w_mean
  T* gr = wm + n;                            // [n] gradient, later V^T grad / L
  T* cs = gr + n;                            // [n] rotations
  T* lam = cs + n;                           // [n]
  T* cg = lam + n;                           // [n] w_cov grad
  T* lw = cg + n;                            // [pm + 2]
  int* lidx = reinterpret_cast<int*>(lw + ((pm + 2 + 1) & ~1));   // [pm + 2]
  int* ibuf = lidx + ((pm + 2 + 1) & ~1);                         // [4]: pivot row, flag
  unsigned short* dec = reinterpret_cast<unsigned short*>(ibuf + 4);
  const int nb = n >> 1, ndec = nb * (nb - 1) / 2;
  int* piv = reinterpret_cast<int*>(dec + ((ndec + 7) & ~7));     // [n]
  T* red = reinterpret_cast<T*>(piv + ((n + 1) & ~1));            // [NT / 64 * 2] pivot search

  for (int it = tid; it < ndec; it += NT) {
    int bj = 1;
    while ((bj + 1) * bj / 2 <= it) ++bj;
    dec[it] = (unsigned short)((it - bj * (bj - 1) / 2) | (bj << 8));
  }
  if (tid == 0) ibuf[1] = 0;
  const int64_t pt = blockIdx.x;
  if (P.only_flagged && !(P.flags[pt] & MIA_FLAG_RETRY)) return;
  const T km1 = T(k - 1);
  const int cnt = P.cnt[pt];
  const T* win = P.Win + pt * P.w_stride;
  T* wout = P.Wout + pt * (int64_t)k * k;
  int flag = 0;
  if (cnt > pm || cnt > P.p_cap) {   // loud failure: never update with a truncated list
    if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
    const T nanv = T(__builtin_nanf(""));
    for (int it = tid; it < k * k; it += NT) wout[it] = nanv;
    return;
  }
  if (cnt == 0) {                    // no local observation: weights returned as they came (ienks.py:135)
    for (int it = tid; it < k * k; it += NT) wout[it] = win[it];
    if (P.flags && tid == 0) P.flags[pt] = 0;
    return;
  }
  // ---- W into LDS, local lists
  for (int it = tid; it < k * k; it += NT) { const int i = it / k, j = it - i * k; A[i * lda + j] = win[it]; }
  for (int j = tid; j < cnt; j += NT) { lidx[j] = P.idx[pt * P.p_cap + j]; lw[j] = T(P.w[pt * P.p_cap + j]); }
  __syncthreads();
  // ---- w_mean and Wp (in place); gather + sqrt(rho) scale of the local block (wrapper.py:91-97)
  for (int i = tid; i < n; i += NT) {
    T acc = T(0);
    if (i < k) { for (int j = 0; j < k; ++j) acc += A[i * lda + j]; acc = (acc - T(1)) / T(k); }
    wm[i] = acc;
  }
  {
    constexpr int VW = 16 / sizeof(T);
    const int kpv = kp / VW;
    using VT = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
    for (int it = tid; it < cnt * kpv; it += NT) {
      const int j = it / kpv, c = it - j * kpv;
      VT v = reinterpret_cast<const VT*>(P.rec + (int64_t)lidx[j] * kp)[c];
      const T wj = lw[j];
      if constexpr (sizeof(T) == 4) { v.x *= wj; v.y *= wj; v.z *= wj; v.w *= wj; }
      else { v.x *= wj; v.y *= wj; }
      reinterpret_cast<VT*>(Yt + (size_t)j * kp)[c] = v;
    }
  }
  __syncthreads();
  for (int it = tid; it < k * k; it += NT) { const int i = it / k, j = it - i * k; A[i * lda + j] -= wm[i]; }
  __syncthreads();
  // ---- Wp^-1: in-place Gauss-Jordan with partial pivoting (rows swapped physically, columns unswapped at the end)
  if (P.need_inv) {
    for (int c = 0; c < k; ++c) {
      // pivot search over rows c..k-1 of column c
      T best = T(-1); int brow = c;
      for (int r = c + tid; r < k; r += NT) { const T a = t_abs(A[r * lda + c]); if (a > best) { best = a; brow = r; } }
      for (int o = 32; o > 0; o >>= 1) {
        const T ob = __shfl_xor(best, o, 64); const int orow = __shfl_xor(brow, o, 64);
        if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
      }
      if ((tid & 63) == 0) { red[2 * (tid >> 6)] = best; red[2 * (tid >> 6) + 1] = T(brow); }
      __syncthreads();
      if (tid == 0) {
        T b = red[0]; int br = (int)red[1];
        for (int wv = 1; wv < NT / 64; ++wv) if (red[2 * wv] > b) { b = red[2 * wv]; br = (int)red[2 * wv + 1]; }
        ibuf[0] = br; piv[c] = br;
        if (!(b > T(0))) ibuf[1] = 1;      // singular (or NaN) column: flagged, result non-finite like the reference's 1/s
      }
      __syncthreads();
      const int pr = ibuf[0];
      if (pr != c) for (int j = tid; j < k; j += NT) { const T t0 = A[c * lda + j]; A[c * lda + j] = A[pr * lda + j]; A[pr * lda + j] = t0; }
      __syncthreads();
      const T pinv = T(1) / A[c * lda + c];
      __syncthreads();
      for (int j = tid; j < k; j += NT) A[c * lda + j] = (j == c ? T(1) : A[c * lda + j]) * pinv;
      // column c of the other rows is the multiplier: kept in registers/LDS scratch (cs reused) before it is cleared
      for (int i = tid; i < k; i += NT) cs[i] = (i == c) ? T(0) : A[i * lda + c];
      __syncthreads();
      for (int it = tid; it < k * k; it += NT) {
        const int i = it / k, j = it - i * k;
        if (i != c) { const T f = cs[i]; A[i * lda + j] = (j == c ? T(0) : A[i * lda + j]) - f * A[c * lda + j]; }
      }
      __syncthreads();
    }
    for (int c = k - 1; c >= 0; --c) {
      const int pr = piv[c];
      if (pr != c) for (int i = tid; i < k; i += NT) { const T t0 = A[i * lda + c]; A[i * lda + c] = A[i * lda + pr]; A[i * lda + pr] = t0; }
      __syncthreads();
    }
    if (tid == 0 && ibuf[1]) flag |= MIA_FLAG_NONFINITE;
  }
  // ---- D (obs-major): transform D = Wp^-1 Yl, bundle D = Yl / eps (in place, Dt aliases Yt)
  if (bundle) {
    for (int it = tid; it < cnt * k; it += NT) { const int b = it / k, i = it - b * k; Yt[(size_t)b * kp + i] *= P.inv_eps; }
  } else {
    for (int it = tid; it < cnt * k; it += NT) {
      const int b = it / k, i = it - b * k;
      T acc = T(0);
      for (int j = 0; j < k; ++j) acc += A[i * lda + j] * Yt[(size_t)b * kp + j];
      Dt[(size_t)b * kp + i] = acc;
    }
  }
  __syncthreads();
  // ---- gradient
  for (int i = tid; i < n; i += NT) {
    T acc = T(0);
    if (i < k) { for (int b = 0; b < cnt; ++b) acc += Dt[(size_t)b * kp + i] * Yt[(size_t)b * kp + k]; acc = km1 * wm[i] - acc; }
    gr[i] = acc;
  }
  int jstat;
  if (P.nd > 0) {
    // ---- dual route (tau = 1): Pn = (k-1) I + D D^T has rank-p structure, so the p x p matrix D^T D = V L V^T is
    //      decomposed instead and every function of Pn is applied as
    //      f(Pn) = f(k-1) I + M diag((f(k-1+l) - f(k-1)) / l) M^T,  M = D V,
    //      with the divided differences in closed, cancellation-free form (u = sqrt(k-1+l), a = sqrt(k-1)):
    //      Pn^-1: -1 / ((k-1)(k-1+l));   sqrt(k-1) Pn^-1/2: -1 / (u (u + a))      (an eighth of the k x k Jacobi work)
    const int nd = P.nd;
    for (int it = tid; it < (nd - cnt) * kp; it += NT) Dt[(size_t)cnt * kp + it] = T(0);   // padding rows of D^T
    __syncthreads();
    for (int it = tid; it < nd * nd; it += NT) {
      const int a = it / nd, b = it - a * nd;
      A[a * lda + b] = (a == b) ? T(1) : T(0);
      if (a > b) continue;
      T acc = T(0);
      for (int i = 0; i < k; ++i) acc += Dt[(size_t)a * kp + i] * Dt[(size_t)b * kp + i];
      S[a * lda + b] = acc;
    }
    __syncthreads();
    const bool conv = jacobi_sym<T, NT>(S, A, cs, dec, nd, nd, lda, km1, P.rot_tol2, P.stop_tol2, P.max_sweeps, jstat);
    if (!conv) flag |= MIA_FLAG_NOCONV;
    const T ar = t_sqrt(km1);
    for (int r = tid; r < nd; r += NT) {
      T l = S[r * lda + r];
      l = l > T(0) ? l : T(0);
      const T u = t_sqrt(km1 + l);
      lam[r] = (r < cnt) ? T(-1) / (u * (u + ar)) : T(0);            // sqrt(k-1) Pn^-1/2 - I
      cs[r] = (r < cnt) ? T(-1) / (km1 * (km1 + l)) : T(0);          // Pn^-1 - I / (k-1)
    }
    __syncthreads();
    // M = D V  (k x nd), kept in the S buffer
    for (int it = tid; it < k * nd; it += NT) {
      const int i = it / nd, r = it - i * nd;
      T acc = T(0);
      for (int b = 0; b < cnt; ++b) acc += Dt[(size_t)b * kp + i] * A[b * lda + r];
      S[i * lda + r] = acc;
    }
    __syncthreads();
    for (int r = tid; r < nd; r += NT) {       // cs <- diag(gM) M^T grad
      T acc = T(0);
      for (int i = 0; i < k; ++i) acc += S[i * lda + r] * gr[i];
      cs[r] *= acc;
    }
    __syncthreads();
    for (int i = tid; i < k; i += NT) {
      T acc = gr[i] / km1;
      for (int r = 0; r < nd; ++r) acc += S[i * lda + r] * cs[r];
      cg[i] = wm[i] - P.tau * acc;              // updated w_mean
    }
    __syncthreads();
    for (int it = tid; it < k * k; it += NT) {
      const int i = it / k, j = it - i * k;
      T acc = cg[i] + (i == j ? T(1) : T(0));
      for (int r = 0; r < nd; ++r) acc += lam[r] * S[i * lda + r] * S[j * lda + r];
      if (!(t_abs(acc) <= T(1e30))) flag |= MIA_FLAG_NONFINITE;
      wout[it] = acc;
    }
  } else {
  // ---- the updated precision (canonical upper triangle of S)
  {
    const T one_m_tau = T(1) - P.tau;
    for (int it = tid; it < n * n; it += NT) {
      const int a = it / n, b = it - a * n;
      if (a > b) continue;
      T acc = T(0);
      if (b < k) {
        for (int j = 0; j < cnt; ++j) acc += Dt[(size_t)j * kp + a] * Dt[(size_t)j * kp + b];
        if (a == b) acc += km1;
        acc *= P.tau;
        if (one_m_tau != T(0)) {      // (1 - tau) (k-1) Wp^-T Wp^-1
          T pr = T(0);
          for (int j = 0; j < k; ++j) pr += A[j * lda + a] * A[j * lda + b];
          acc += one_m_tau * km1 * pr;
        }
      } else if (a == b) acc = km1;   // padding row of an odd ensemble size: decoupled
      S[a * lda + b] = acc;
    }
  }
  __syncthreads();
  // ---- eigendecomposition of Pn; V accumulates in A (Wp^-1 is no longer needed)
  for (int it = tid; it < n * n; it += NT) { const int a = it / n, b = it - a * n; A[a * lda + b] = (a == b) ? T(1) : T(0); }
  __syncthreads();
  const bool conv = jacobi_sym<T, NT>(S, A, cs, dec, n, n, lda, T(0), P.rot_tol2, P.stop_tol2, P.max_sweeps, jstat);
  if (!conv) flag |= MIA_FLAG_NOCONV;
  for (int r = tid; r < n; r += NT) {
    const T l = S[r * lda + r];
    lam[r] = l;
    T acc = T(0);
    for (int b = 0; b < k; ++b) acc += A[b * lda + r] * gr[b];
    cs[r] = (r < k) ? acc / l : T(0);                  // L^-1 V^T grad
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    T acc = T(0);
    for (int r = 0; r < k; ++r) acc += A[i * lda + r] * cs[r];
    cg[i] = wm[i] - P.tau * acc;                        // updated w_mean
  }
  for (int r = tid; r < n; r += NT) lam[r] = (r < k) ? t_sqrt(km1 / lam[r]) : T(0);
  __syncthreads();
  for (int it = tid; it < k * k; it += NT) {
    const int i = it / k, j = it - i * k;
    T acc = cg[i];
    for (int r = 0; r < k; ++r) acc += lam[r] * A[i * lda + r] * A[j * lda + r];
    if (!(t_abs(acc) <= T(1e30))) flag |= MIA_FLAG_NONFINITE;
    wout[it] = acc;
  }
  }
  if (P.flags) {
    if (tid == 0) ibuf[2] = 0;
    __syncthreads();
    if (flag) atomicOr(&ibuf[2], flag);
    __syncthreads();
    if (tid == 0) P.flags[pt] = ibuf[2] | (jstat << 8);
  }
}

template <typename T>
static int ienks_update_impl(const T* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1, const T* rec, int64_t Pn,
                             const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                             T tau, T epsilon, T* W_out, int32_t* flags_opt, hipStream_t stream, int only_flagged = 0) {
  if (g1 < g0 || g0 < 0 || k < 2 || Pn < 0 || p_cap < 1 || p_max < 0) return MIA_ERR_SIZE;
  if (!(tau >= T(0)) || !(tau <= T(1))) return MIA_ERR_SIZE;              // bound_tensor(0, 1), interface/ienks.py:84
  if (w_stride != 0 && w_stride != (int64_t)k * k) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!W_in || !W_out || !nbr_cnt || !nbr_idx || !nbr_w) return MIA_ERR_NULL;
  if (Pn > 0 && !rec) return MIA_ERR_NULL;
  if (p_max > p_cap) p_max = p_cap;
  if (ng > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  IenksParams<T> ap;
  ap.only_flagged = only_flagged;
  if (only_flagged && !flags_opt) return MIA_ERR_NULL;
  ap.Win = W_in; ap.w_stride = w_stride; ap.k = k; ap.ng = ng; ap.rec = rec; ap.kp = (k + 1 + 3) & ~3;
  ap.cnt = nbr_cnt; ap.idx = nbr_idx; ap.w = nbr_w; ap.p_cap = p_cap; ap.p_max = p_max;
  ap.tau = tau; ap.inv_eps = epsilon > T(0) ? T(1) / epsilon : T(0);
  ap.Wout = W_out; ap.flags = flags_opt;
  ap.n = (k + 1) & ~1; ap.lda = ap.n + 1;
  ap.nd = (tau == T(1) && p_max <= k && !MIA_EXP_FLAG("MIA_IENKS_PRIMAL")) ? (((p_max > 0 ? p_max : 1) + 1) & ~1) : 0;
  ap.rows = ap.nd > 0 ? ap.nd : (p_max > 0 ? p_max : 1);
  if (ap.n > 510) return MIA_ERR_UNSUPPORTED;
  // the inverse of Wp is needed for the transform variant's D and, when tau < 1, for w_prec
  ap.need_inv = (ap.inv_eps == T(0) || tau < T(1)) ? 1 : 0;
  ap.max_sweeps = sizeof(T) == 4 ? 16 : 24;
  const T stop_tol = sizeof(T) == 4 ? T(2.4e-7) : T(9e-16);
  const T rot_tol = stop_tol * T(0.5);
  ap.stop_tol2 = stop_tol * stop_tol; ap.rot_tol2 = rot_tol * rot_tol;
  const int nbk = ap.n / 2;
  size_t e = 2 * (size_t)ap.n * ap.lda + (ap.inv_eps != T(0) ? 1 : 2) * (size_t)ap.rows * ap.kp + 5 * (size_t)ap.n +
             (size_t)((p_max + 3) & ~1);
  size_t lds = e * sizeof(T) + (size_t)((p_max + 3) & ~1) * sizeof(int) + 4 * sizeof(int) +
               (size_t)((nbk * (nbk - 1) / 2 + 7) & ~7) * sizeof(unsigned short) + (size_t)((ap.n + 1) & ~1) * sizeof(int) +
               (size_t)(256 / 64 * 2) * sizeof(T);
  lds = align_up(lds, 16);
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  const bool small = ap.n <= 44 && MIA_EXP_FLAG("MIA_IENKS_NARROW");
  if (small) {
    auto kern = ienks_update_kernel<T, 64>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)ng), dim3(64), lds, stream>>>(ap);
  } else {
    auto kern = ienks_update_kernel<T, 256>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)ng), dim3(256), lds, stream>>>(ap);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Ensemble transform with per-grid-point weights, _apply_weights with 3-D weights (interface/base.py:257-278):
// xa[v][j][g] = mean_g + sum_i (x[v][i][g] - mean_g) W[g][i][j].  HBM-bound on W (4 k^2 bytes per point against
// 8 k m for the state): one wavefront per grid point streams its W row by row (lane j owns column j, rows are
// contiguous), the m state columns of the point sit in LDS.
template <typename T>
struct LocalApplyParams { const T* X; int64_t ldx; int m, k; int64_t g0, ng; const T* W; T* Xa; int64_t ldo, o0; };

template <typename T>
__global__ __launch_bounds__(256) void apply_local_weights_kernel(LocalApplyParams<T> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = P.k;
  T* xp = reinterpret_cast<T*>(smem_raw) + (size_t)wave * (k + 1);   // [k] perturbations, [k] = mean
  const int64_t pt = (int64_t)blockIdx.x * 4 + wave;
  if (pt >= P.ng) return;
  const T* w = P.W + pt * (int64_t)k * k;
  for (int mi = 0; mi < P.m; ++mi) {
    const T* xrow = P.X + (int64_t)mi * k * P.ldx + P.g0 + pt;
    T part = T(0);
    for (int i = lane; i < k; i += 64) { const T v = xrow[(int64_t)i * P.ldx]; xp[i] = v; part += v; }
    const T mean = wave_sum(part) / T(k);
    __builtin_amdgcn_wave_barrier();
    T* orow = P.Xa + (int64_t)mi * k * P.ldo + P.o0 + pt;
    for (int j0 = 0; j0 < k; j0 += 64) {
      const int j = j0 + lane;
      if (j < k) {
        T acc = T(0);
#pragma unroll 4
        for (int i = 0; i < k; ++i) acc += (xp[i] - mean) * w[(int64_t)i * k + j];
        orow[(int64_t)j * P.ldo] = mean + acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
static int apply_local_weights_impl(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1, const T* W, T* Xa,
                                    int64_t ldo, int64_t o0, hipStream_t stream) {
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!X || !W || !Xa) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  const int64_t nb = (ng + 3) / 4;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  const size_t lds = 4 * (size_t)(k + 1) * sizeof(T);
  if (lds > 64 * 1024) return MIA_ERR_UNSUPPORTED;
  LocalApplyParams<T> ap{X, ldx, m, k, g0, ng, W, Xa, ldo, o0};
  apply_local_weights_kernel<T><<<dim3((unsigned)nb), dim3(256), lds, stream>>>(ap);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_lienks_update_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1, const float* rec,
                                     int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                     int p_cap, int p_max, float tau, float epsilon, float* W_out, int32_t* flags_opt,
                                     void* stream) {
  (void)hipGetLastError();
  return ienks_update_impl<float>(W_in, w_stride, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, tau, epsilon,
                                  W_out, flags_opt, (hipStream_t)stream);
}
// tau = 1 through the eigensolver-free weights kernel (letkf_cheb.hip); the general kernel above redoes what it declines
extern "C" int mia_lienks_update_matfun_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                                            const float* rec, int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx,
                                            const double* nbr_w, int p_cap, int p_max, float epsilon, float* W_out,
                                            int32_t* flags, int32_t* retry_count, void* stream) {
  (void)hipGetLastError();
  if (g1 < g0 || g0 < 0 || k < 2 || P < 0 || p_cap < 1 || p_max < 0) return MIA_ERR_SIZE;
  if (w_stride != 0 && w_stride != (int64_t)k * k) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!W_in || !W_out || !nbr_cnt || !nbr_idx || !nbr_w || !flags || !retry_count) return MIA_ERR_NULL;
  if (P > 0 && !rec) return MIA_ERR_NULL;
  if (p_max > p_cap) p_max = p_cap;
  const IenksOpts opt{epsilon > 0.0f ? 2 : 1, W_in, w_stride, epsilon > 0.0f ? 1.0f / epsilon : 1.0f};
  // (X = W_out as a harmless non-NULL placeholder: no state row is transformed, m = 0)
  return cheb_analysis_launch(W_out, ng, 0, k, 0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, 1.0f, 0, 0.0f, W_out, ng, 0,
                              flags, retry_count, nullptr, nullptr, (hipStream_t)stream, 0, 0, nullptr, W_out, &opt);
}
extern "C" int mia_lienks_update_retry_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                                           const float* rec, int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx,
                                           const double* nbr_w, int p_cap, int p_max, float tau, float epsilon,
                                           float* W_out, int32_t* flags, void* stream) {
  (void)hipGetLastError();
  return ienks_update_impl<float>(W_in, w_stride, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, tau, epsilon,
                                  W_out, flags, (hipStream_t)stream, 1);
}

extern "C" int mia_lienks_update_f64(const double* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1, const double* rec,
                                     int64_t P, const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                     int p_cap, int p_max, double tau, double epsilon, double* W_out, int32_t* flags_opt,
                                     void* stream) {
  (void)hipGetLastError();
  return ienks_update_impl<double>(W_in, w_stride, k, g0, g1, rec, P, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, tau, epsilon,
                                   W_out, flags_opt, (hipStream_t)stream);
}
extern "C" int mia_apply_local_weights_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const float* W, float* Xa, int64_t ldo, int64_t o0, void* stream) {
  (void)hipGetLastError();
  // tiles of sixteen points (apply_local.hip) where the shape allows: the same argument checks first
  if (g1 > g0 && g0 >= 0 && m >= 1 && k >= 2 && X && W && Xa && ldx >= g1 && ldo >= o0 + (g1 - g0)) {
    const int rc = apply_local_tile_launch(X, ldx, m, k, g0, g1 - g0, W, Xa, ldo, o0, (hipStream_t)stream);
    if (rc != MIA_ERR_UNSUPPORTED) return rc;
  }
  return apply_local_weights_impl<float>(X, ldx, m, k, g0, g1, W, Xa, ldo, o0, (hipStream_t)stream);
}
extern "C" int mia_apply_local_weights_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const double* W, double* Xa, int64_t ldo, int64_t o0, void* stream) {
  (void)hipGetLastError();
  return apply_local_weights_impl<double>(X, ldx, m, k, g0, g1, W, Xa, ldo, o0, (hipStream_t)stream);
}

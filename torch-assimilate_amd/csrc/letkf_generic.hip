// Generic LDS-resident local analysis for gfx950 (any k / p_max that fits the 160 KB LDS,
// float32 and float64).  One workgroup walks a contiguous run of grid points; for each
// point the whole local block lives in LDS:
//
//   gather + sqrt(rho)-scale of the local obs columns      interface/wrapper.py:91-97
//   Gram matrix                                            core/utils.py:172 via etkf.py:68
//   symmetric eigensolve, clamp >= 0, +(k-1)/inf           core/utils.py:57-60, etkf.py:67
//   w_mean = Pa (Yb d^T), W = V diag(sqrt((k-1)/l)) V^T    core/etkf.py:70-76
//   xa = mean + X' (w_mean 1^T + W)                        interface/base.py:257-278
//
// Two algebraically identical routes, chosen per launch:
//  * primal (p_max > k): eigensolve of the k x k matrix C = Yl Yl^T, exactly the
//    reference's operation sequence;
//  * dual (p_max <= k): C has rank <= p, so the p x p matrix S = Yl^T Yl is decomposed
//    instead (S = V L V^T) and every function of A = C + reg*I is applied through
//       f(A) = f(reg) I + Yl V diag((f(l+reg) - f(reg)) / l) V^T Yl^T,
//    with the divided differences written in their cancellation-free closed forms.
//    For p == 0 this collapses to the reference's prior branch sqrt(inf)*I (etkf.py:91-95).
//
// The eigensolver is a parallel-order (round-robin tournament) cyclic Jacobi with the
// rotation threshold scaled by (|a_pp| + reg): what enters the result is f(l + reg), so
// off-diagonal mass below eps*(l + reg) is already below rounding of the reference result.
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

int sys_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                        const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                        float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                        float* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream);

template <typename T>
struct AnaParams {
  const T* X; int64_t ldx; int m; int k;
  int64_t g0, ng;
  const T* rec; int kp;              // packed obs records [P][kp]: yb[0..k), d, pad
  const int32_t* cnt; const int32_t* idx; const double* w; int p_cap; int p_max;
  T reg;                             // (k-1)/inf
  T* Xa; int64_t ldo, o0;
  T* W; int32_t* flags;
  int dual; int nmax; int lda; int ldy; int pts_per_block;
  int max_sweeps; T tol;
  int kernel_mode;                   // 0 = linear (ETKF), 1 = RBF (KETKF, primal only)
  T gamma;
};

template <typename T, int NT>
__global__ __launch_bounds__(NT) void letkf_generic_kernel(AnaParams<T> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x;
  const int k = P.k, pm = P.p_max, nmax = P.nmax, lda = P.lda, ldy = P.ldy;
  T* Yl = sm;                        // [k][ldy]   local, sqrt(rho)-scaled obs perturbations
  T* dl = Yl + (size_t)k * ldy;      // [pm + 2]   local scaled innovations
  T* A = dl + (pm + 2);              // [nmax][lda]
  T* V = A + (size_t)nmax * lda;     // [nmax][lda]
  T* cs = V + (size_t)nmax * lda;    // [nmax]
  T* gW = cs + nmax;                 // [nmax] per-mode factor of the square-root term
  T* gM = gW + nmax;                 // [nmax] per-mode factor of the mean term
  T* av = gM + nmax;                 // [nmax]
  T* uv = av + nmax;                 // [nmax]
  T* wbar = uv + nmax;               // [k]
  T* xp = wbar + k;                  // [k]
  T* zb = xp + k;                    // [nmax]
  T* qb = zb + nmax;                 // [nmax]
  T* sb = qb + nmax;                 // [nmax]
  T* red = sb + nmax;                // [4]
  int* iflag = reinterpret_cast<int*>(red + 4);  // [4]
  T* Mq = reinterpret_cast<T*>(iflag + 4);       // [k][lda] only for W output on the dual route (primal: M = V)
  int* lidx = reinterpret_cast<int*>(Mq + ((P.W && P.dual) ? (size_t)k * lda : 0));  // [pm + 1]
  T* lw = reinterpret_cast<T*>(lidx + ((pm + 2) & ~1));                  // [pm + 1]

  const T km1 = T(k - 1);
  const T reg = P.reg;
  const T f0 = P.dual ? t_sqrt(km1 / reg) : T(0);

  const int64_t pt_begin = (int64_t)blockIdx.x * P.pts_per_block;
  int64_t pt_end = pt_begin + P.pts_per_block;
  if (pt_end > P.ng) pt_end = P.ng;

  for (int64_t pt = pt_begin; pt < pt_end; ++pt) {
    const int64_t g = P.g0 + pt;
    int cnt = P.cnt[pt];
    int flag = 0;
    __syncthreads();  // previous point's LDS no longer in use
    if (cnt > pm || cnt > P.p_cap) {
      // loud failure: never analyse with a truncated list
      if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
      const T nanv = T(__builtin_nanf(""));
      for (int it = tid; it < P.m * k; it += NT) P.Xa[(int64_t)it * P.ldo + P.o0 + pt] = nanv;
      if (P.W) for (int it = tid; it < k * k; it += NT) P.W[pt * (int64_t)k * k + it] = nanv;
      continue;
    }
    // ---- neighbour list -> LDS
    for (int j = tid; j < cnt; j += NT) {
      lidx[j] = P.idx[pt * P.p_cap + j];
      lw[j] = T(P.w[pt * P.p_cap + j]);
    }
    __syncthreads();
    // ---- gather + scale (wrapper.py:91-97); records are contiguous in the member index
    for (int it = tid; it < cnt * (k + 1); it += NT) {
      const int j = it / (k + 1), i = it - j * (k + 1);
      const T v = P.rec[(int64_t)lidx[j] * P.kp + i] * lw[j];
      if (i < k) Yl[i * ldy + j] = v; else dl[j] = v;
    }
    const int ntrue = P.dual ? cnt : k;          // modes that can carry signal
    const int n = (ntrue + 1) & ~1;              // even order for the tournament
    if (P.dual) {
      // zero pad column when cnt is odd
      if (n > cnt) { for (int i = tid; i < k; i += NT) Yl[i * ldy + cnt] = T(0); if (tid == 0) dl[cnt] = T(0); }
    }
    __syncthreads();
    // ---- Gram matrix + identity
    if (P.dual) {
      for (int it = tid; it < n * n; it += NT) {
        const int a = it / n, b = it - a * n;
        if (a <= b) {
          T acc = T(0);
          for (int i = 0; i < k; ++i) acc += Yl[i * ldy + a] * Yl[i * ldy + b];
          A[a * lda + b] = acc; A[b * lda + a] = acc;
        }
        V[a * lda + b] = (a == b) ? T(1) : T(0);
      }
    } else if (P.kernel_mode == 0) {
      for (int it = tid; it < n * n; it += NT) {
        const int a = it / n, b = it - a * n;
        if (a <= b) {
          T acc = T(0);
          if (b < k) for (int j = 0; j < cnt; ++j) acc += Yl[a * ldy + j] * Yl[b * ldy + j];
          A[a * lda + b] = acc; A[b * lda + a] = acc;
        }
        V[a * lda + b] = (a == b) ? T(1) : T(0);
      }
    } else {
      // RBF Gram matrix exp(-gamma*|y_a - y_b|^2) (kernels/rbf.py:75-81,110-111)
      for (int it = tid; it < n * n; it += NT) {
        const int a = it / n, b = it - a * n;
        if (a <= b) {
          T acc = T(0);
          if (b < k) {
            for (int j = 0; j < cnt; ++j) { const T df = Yl[a * ldy + j] - Yl[b * ldy + j]; acc += df * df; }
            acc = t_exp(-P.gamma * acc);
          }
          A[a * lda + b] = acc; A[b * lda + a] = acc;
        }
        V[a * lda + b] = (a == b) ? T(1) : T(0);
      }
    }
    __syncthreads();
    // ---- right-hand side of the mean weights
    //   dual  : b = dl (p)           primal linear : b = Yl dl (k)
    //   primal RBF : b = centred k(Yl, dl)   (core/ketkf.py:86-89)
    if (!P.dual) {
      if (P.kernel_mode == 0) {
        for (int i = tid; i < k; i += NT) {
          T acc = T(0);
          for (int j = 0; j < cnt; ++j) acc += Yl[i * ldy + j] * dl[j];
          zb[i] = acc;
        }
        if (n > k && tid == 0) zb[k] = T(0);
        __syncthreads();
      } else {
        // row means of K (k_partial_mean before its own centring), core/ketkf.py:77
        for (int i = tid; i < k; i += NT) {
          T acc = T(0);
          for (int j = 0; j < k; ++j) acc += A[i * lda + j];
          uv[i] = acc / T(k);
          T ko = T(0);
          for (int j = 0; j < cnt; ++j) { const T df = Yl[i * ldy + j] - dl[j]; ko += df * df; }
          zb[i] = t_exp(-P.gamma * ko);
        }
        __syncthreads();
        if (tid == 0) {
          T gm = T(0), om = T(0);
          for (int i = 0; i < k; ++i) { gm += uv[i]; om += zb[i]; }
          red[0] = gm / T(k); red[1] = om / T(k);
        }
        __syncthreads();
        // K_c[a][b] = K[a][b] - colmean[b] - (rowmean[a] - grand)   (ketkf.py:78-81);
        // K symmetric -> colmean == rowmean
        for (int it = tid; it < k * k; it += NT) {
          const int a = it / k, b = it - a * k;
          A[a * lda + b] = A[a * lda + b] - uv[b] - (uv[a] - red[0]);
        }
        // k_obs centred (ketkf.py:87-88)
        for (int i = tid; i < k; i += NT) zb[i] = zb[i] - red[1] - (uv[i] - red[0]);
        if (n > k && tid == 0) zb[k] = T(0);
        __syncthreads();
      }
    }
    // ---- symmetric eigensolve
    const bool conv = jacobi_lds<T, NT>(A, V, cs, iflag, n, n, lda, reg, P.tol, P.max_sweeps);
    if (!conv) flag |= MIA_FLAG_NOCONV;
    // ---- per-mode factors (clamp >= 0 then + reg: core/utils.py:58-59)
    for (int r = tid; r < n; r += NT) {
      T lam = A[r * lda + r];
      lam = lam > T(0) ? lam : T(0);
      const T le = lam + reg;
      T acc = T(0);
      if (P.dual) {
        const T sl = t_sqrt(le), sr = t_sqrt(reg);
        gW[r] = (r < ntrue) ? -t_sqrt(km1) / (sl * sr * (sr + sl)) : T(0);
        for (int b = 0; b < cnt; ++b) acc += V[b * lda + r] * dl[b];
      } else {
        gW[r] = (r < ntrue) ? t_sqrt(km1 / le) : T(0);
        for (int b = 0; b < k; ++b) acc += V[b * lda + r] * zb[b];
      }
      gM[r] = (r < ntrue) ? T(1) / le : T(0);
      av[r] = acc * gM[r];
    }
    __syncthreads();
    // u = V (gM o V^T b)
    for (int b = tid; b < n; b += NT) {
      T acc = T(0);
      for (int r = 0; r < n; ++r) acc += V[b * lda + r] * av[r];
      uv[b] = acc;
    }
    __syncthreads();
    // w_mean: dual Yl u, primal u
    for (int i = tid; i < k; i += NT) {
      T acc;
      if (P.dual) { acc = T(0); for (int b = 0; b < cnt; ++b) acc += Yl[i * ldy + b] * uv[b]; }
      else acc = uv[i];
      wbar[i] = acc;
    }
    __syncthreads();
    // ---- ensemble transform, one state row at a time
    for (int mi = 0; mi < P.m; ++mi) {
      const T* xrow = P.X + (int64_t)mi * k * P.ldx + g;
      for (int i = tid; i < k; i += NT) xp[i] = xrow[(int64_t)i * P.ldx];
      __syncthreads();
      if (tid == 0) {
        T s = T(0);
        for (int i = 0; i < k; ++i) s += xp[i];
        red[2] = s / T(k);
      }
      __syncthreads();
      const T xm = red[2];
      for (int i = tid; i < k; i += NT) xp[i] -= xm;
      __syncthreads();
      if (tid == 0) {
        T s = T(0);
        for (int i = 0; i < k; ++i) s += xp[i] * wbar[i];
        red[3] = s;
      }
      // z = X' B   (dual: B = Yl, primal: B = I)
      for (int b = tid; b < n; b += NT) {
        T acc = T(0);
        if (P.dual) { if (b < cnt) for (int i = 0; i < k; ++i) acc += xp[i] * Yl[i * ldy + b]; }
        else acc = b < k ? xp[b] : T(0);
        zb[b] = acc;
      }
      __syncthreads();
      for (int r = tid; r < n; r += NT) {
        T acc = T(0);
        for (int b = 0; b < n; ++b) acc += zb[b] * V[b * lda + r];
        qb[r] = acc * gW[r];
      }
      __syncthreads();
      for (int b = tid; b < n; b += NT) {
        T acc = T(0);
        for (int r = 0; r < n; ++r) acc += qb[r] * V[b * lda + r];
        sb[b] = acc;
      }
      __syncthreads();
      const T mterm = xm + red[3];
      T* orow = P.Xa + (int64_t)mi * k * P.ldo + P.o0 + pt;
      for (int j = tid; j < k; j += NT) {
        T acc;
        if (P.dual) { acc = f0 * xp[j]; for (int b = 0; b < cnt; ++b) acc += sb[b] * Yl[j * ldy + b]; }
        else acc = sb[j];
        const T out = mterm + acc;
        if (!(out == out) || t_abs(out) > T(1e30)) flag |= MIA_FLAG_NONFINITE;
        orow[(int64_t)j * P.ldo] = out;
      }
      __syncthreads();
    }
    // ---- optional weights output: w_mean_i + f0*delta_ij + sum_r gW_r M_ir M_jr,  M = B V
    if (P.W) {
      const T* Mm = V;
      if (P.dual) {
        for (int it = tid; it < k * n; it += NT) {
          const int i = it / n, r = it - i * n;
          T acc = T(0);
          for (int b = 0; b < cnt; ++b) acc += Yl[i * ldy + b] * V[b * lda + r];
          Mq[i * lda + r] = acc;
        }
        Mm = Mq;
        __syncthreads();
      }
      T* wout = P.W + pt * (int64_t)k * k;
      for (int it = tid; it < k * k; it += NT) {
        const int i = it / k, j = it - i * k;
        T acc = wbar[i] + (i == j ? f0 : T(0));
        for (int r = 0; r < n; ++r) acc += gW[r] * Mm[i * lda + r] * Mm[j * lda + r];
        wout[it] = acc;
      }
    }
    if (P.flags) {
      // any thread may have seen a non-finite value
      if (tid == 0) iflag[2] = 0;
      __syncthreads();
      if (flag) atomicOr(&iflag[2], flag);
      __syncthreads();
      if (tid == 0) P.flags[pt] = iflag[2];
    }
  }
}

// [k][P] (+ d[P]) -> obs-major records [P][kp]: one observation's k perturbations and its
// innovation become one contiguous, coalescable 4*kp-byte record for the gather
template <typename T>
__global__ __launch_bounds__(256) void pack_obs_kernel(const T* Yb, const T* d, int k, int64_t P, int kp, T* rec) {
  __shared__ T tile[32][33];
  pack_obs_tile<T>(Yb, d, k, P, kp, rec, (int64_t)blockIdx.x, tile);
}

template <typename T>
static size_t generic_lds_bytes(int k, int p_max, int nmax, int lda, int ldy, bool want_mq) {
  size_t e = (size_t)k * ldy + (p_max + 2) + 2 * (size_t)nmax * lda + 5 * (size_t)nmax + 2 * (size_t)k +
             3 * (size_t)nmax + 4;
  size_t b = e * sizeof(T) + 4 * sizeof(int);
  if (want_mq) b += (size_t)k * lda * sizeof(T);
  b += (size_t)((p_max + 2) & ~1) * sizeof(int) + (size_t)(p_max + 2) * sizeof(T);
  return align_up(b, 16);
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
  const char* which = getenv("MIA_KERNEL");   // experiments: "generic" | "wave" | default (systolic, then wave)
  if constexpr (sizeof(T) == 4) {
    if (!which || which[0] == 's') {
      const int rc = sys_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                         kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream);
      if (rc != MIA_ERR_UNSUPPORTED) return rc;
    }
  }
  if (!which || which[0] != 'g')   // runtime-order kernel (letkf_wave.hip): float64 and orders > 64
    return wave_analysis_launch<T>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor,
                                   kernel_mode, gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream);
  if (only_flagged) return MIA_ERR_UNSUPPORTED;
  AnaParams<T> ap;
  ap.X = X; ap.ldx = ldx; ap.m = m; ap.k = k; ap.g0 = g0; ap.ng = ng; ap.rec = rec; ap.kp = kp;
  ap.cnt = nbr_cnt; ap.idx = nbr_idx; ap.w = nbr_w; ap.p_cap = p_cap; ap.p_max = p_max;
  ap.reg = T(k - 1) / inf_factor;
  ap.Xa = Xa; ap.ldo = ldo; ap.o0 = o0; ap.W = W_opt; ap.flags = flags_opt;
  ap.kernel_mode = kernel_mode; ap.gamma = gamma;
  ap.dual = (kernel_mode == 0 && p_max <= k) ? 1 : 0;
  const int ntrue = ap.dual ? p_max : k;
  ap.nmax = (ntrue + 1) & ~1;
  if (ap.nmax < 2) ap.nmax = 2;
  ap.lda = ap.nmax | 1;
  ap.ldy = (p_max + 1) | 1;
  ap.max_sweeps = sizeof(T) == 4 ? 16 : 24;
  ap.tol = sizeof(T) == 4 ? T(2.4e-7) : T(9e-16);
  const size_t lds = generic_lds_bytes<T>(k, p_max, ap.nmax, ap.lda, ap.ldy, W_opt != nullptr && ap.dual);
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  const bool big = ap.nmax > 32;
  // enough workgroups to fill 256 CUs several times over, but contiguous runs per workgroup
  int ppb = (int)((ng + 8191) / 8192);
  if (ppb < 1) ppb = 1;
  if (ppb > 16) ppb = 16;
  ap.pts_per_block = ppb;
  const int64_t nblk = (ng + ppb - 1) / ppb;
  if (nblk > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  if (big) {
    auto kern = letkf_generic_kernel<T, 256>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nblk), dim3(256), lds, stream>>>(ap);
  } else {
    auto kern = letkf_generic_kernel<T, 64>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nblk), dim3(64), lds, stream>>>(ap);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
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

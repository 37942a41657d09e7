// Fused per-grid-point LETKF analysis for gfx950, second generation ("wave" kernel).
//
// Same mathematics and reference citations as letkf_entry.hip (mask/scale wrapper.py:91-97,
// Gram utils.py:172, eigensolve + clamp + shift utils.py:57-60 / etkf.py:67, w_mean and W
// etkf.py:70-76, transform base.py:257-278; RBF route ketkf.py:65-94), restructured around the
// measured bottleneck of the first kernel (instruction issue: ~1.1e5 wave-instructions per
// analysis, most of them integer modulo arithmetic and a poorly mapped eigenvector update):
//
//  * one 64-lane wavefront owns one grid point (NT = 64; NT = 256 for orders > 44), the local
//    block lives in LDS in the obs-major layout of the packed records, so the gather is a
//    scaled float4 copy;
//  * the symmetric matrix is kept ONCE (canonical upper-triangle addressing), so a Jacobi round
//    updates n/2 diagonal 2x2 blocks (done by the lanes that also derive the rotations) and
//    (n/2)(n/2-1)/2 off-diagonal blocks - half the work of a full-storage update;
//  * tournament indices come from add/compare/select (no division, no tables), the block ->
//    (pair, pair) decode is a table built once per workgroup;
//  * rounds in which no pair exceeds the threshold skip the block/eigenvector update
//    (one barrier + reduction decides, wave-uniformly);
//  * eigenvector rows are mapped (pair = lane % pairs, row = lane / pairs) so each lane keeps
//    one rotation for all its rows.
#include <cstdlib>
#include <type_traits>
#include "mia_common.h"
#include "mia_options.h"
#include "mia_jacobi.h"
#include "mia_jacobi_sym.h"
#include "mia_kernel_prog.h"

namespace mia {

// T = type of the arithmetic, TI = type of the arrays (state, records, analysis, weights).  TI = float with T = double is the
// redo of declined points (mia_letkf_analysis_retry_f32): float32 data, float64 eigensolve -- points whose spectrum is too wide
// for the matrix-function route are also the ones a float32 eigensolver resolves worst (lambda_max / reg ~ 1e4: 1.4e-5 on the
// analysis in tools/stress_tile.py with the float32 kernel).
template <typename T, typename TI = T>
struct WaveParams {
  const TI* X; int64_t ldx; int m; int k;
  int64_t g0, ng;
  const TI* rec; int kp;
  const int32_t* cnt; const int32_t* idx; const double* w; int p_cap; int p_max;
  T reg; TI* Xa; int64_t ldo, o0; TI* W; int32_t* flags;
  int dual; int nmax; int lda; int rows; int pts_per_block; int max_sweeps; T rot_tol2, stop_tol2;
  int kernel_mode; T gamma;   // 0 linear (ETKF), 1 RBF(gamma), 2 kernel expression `prog`
  int only_flagged;
  KernelProgram<T> prog;
};

template <typename T> struct Vec4 { T x, y, z, w; };

// 16x16x4 MFMA in the element type of the kernel (gfx950 has both; identical operand / result lane layout)
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
__device__ inline mfma_f32x4 mfma16(float a, float b, mfma_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ inline mfma_f64x4 mfma16(double a, double b, mfma_f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <typename T> struct MfmaAcc { using type = mfma_f32x4; };
template <> struct MfmaAcc<double> { using type = mfma_f64x4; };

// the three pair statistics of columns a and b of the obs-major local block (rows = local observations)
template <typename T>
__device__ inline T kprog_pair(const KernelProgram<T>& kp, const T* Yt, int kpad, int cnt, int a, int b, bool same) {
  T dt = T(0), sq = T(0), l1 = T(0);
  for (int j = 0; j < cnt; ++j) {
    const T xa = Yt[(size_t)j * kpad + a], xb = Yt[(size_t)j * kpad + b];
    const T df = xa - xb;
    dt += xa * xb; sq += df * df; l1 += t_abs(df);
  }
  return kprog_eval(kp, dt, sq, l1, same);
}

template <typename T, int NT, typename TI = T>
__global__ __launch_bounds__(NT) void letkf_wave_kernel(WaveParams<T, TI> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int k = P.k, kp = P.kp, pm = P.p_max, nmax = P.nmax, lda = P.lda, rows = P.rows;
  T* Yt = reinterpret_cast<T*>(smem_raw);   // [rows][kp]  obs-major local block: yb[0..k), d, pad
  T* S = Yt + (size_t)rows * kp;            // [nmax][lda]  canonical upper storage
  T* V = S + (size_t)nmax * lda;            // [nmax][lda]
  T* cs = V + (size_t)nmax * lda;           // [nmax]
  T* gW = cs + nmax;                        // [nmax]
  T* gM = gW + nmax;                        // [nmax]
  T* av = gM + nmax;                        // [nmax]
  T* uv = av + nmax;                        // [nmax]
  T* zb = uv + nmax;                        // [nmax]
  T* qb = zb + nmax;                        // [nmax]
  T* sb = qb + nmax;                        // [nmax]
  T* xp = sb + nmax;                        // [k]
  T* wbar = xp + k;                         // [k]
  T* red = wbar + k;                        // [8]
  T* lw = red + 8;                          // [pm + 2]
  int* lidx = reinterpret_cast<int*>(lw + ((pm + 2 + 1) & ~1));   // [pm + 2]
  int* iflag = lidx + ((pm + 2 + 1) & ~1);                        // [4]
  unsigned short* dec = reinterpret_cast<unsigned short*>(iflag + 4);  // [nbmax*(nbmax-1)/2]
  const int nbmax = nmax >> 1;
  const int ndec = nbmax * (nbmax - 1) / 2;
  T* Mq = reinterpret_cast<T*>(dec + ((ndec + 7) & ~7));          // [k][lda] (W on the dual route)

  // block (bi < bj) enumeration bj-major: valid for every order n <= nmax
  for (int it = tid; it < ndec; it += NT) {
    int bj = 1;
    while ((bj + 1) * bj / 2 <= it) ++bj;
    const int bi = it - bj * (bj - 1) / 2;
    dec[it] = (unsigned short)(bi | (bj << 8));
  }

  const T km1 = T(k - 1);
  const T reg = P.reg;
  const T f0 = P.dual ? t_sqrt(km1 / reg) : T(0);

  const int64_t pt_begin = (int64_t)blockIdx.x * P.pts_per_block;
  int64_t pt_end = pt_begin + P.pts_per_block;
  if (pt_end > P.ng) pt_end = P.ng;

  for (int64_t pt = pt_begin; pt < pt_end; ++pt) {
    const int64_t g = P.g0 + pt;
    const int cnt = P.cnt[pt];
    int flag = 0;
    __syncthreads();
    if (P.only_flagged && !(P.flags[pt] & MIA_FLAG_RETRY)) continue;
    if (cnt > pm || cnt > P.p_cap) {   // loud failure: never analyse with a truncated list
      if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
      const TI nanv = TI(__builtin_nanf(""));
      for (int it = tid; it < P.m * k; it += NT) P.Xa[(int64_t)it * P.ldo + P.o0 + pt] = nanv;
      if (P.W) for (int it = tid; it < k * k; it += NT) P.W[pt * (int64_t)k * k + it] = nanv;
      continue;
    }
    for (int j = tid; j < cnt; j += NT) {
      lidx[j] = P.idx[pt * P.p_cap + j];
      lw[j] = T(P.w[pt * P.p_cap + j]);
    }
    __syncthreads();
    // Primal route with a linear or RBF core: the member Gram is STREAMED from the packed records on the matrix cores
    // (see the same construction in letkf_cheb.hip), so no local block is staged in LDS and the number of local
    // observations is not bounded by it.  (A kernel expression may need |x - y|_1, which is no product: it keeps the block.)
    const bool streamp = !P.dual && P.kernel_mode != 2;
    // ---- gather + sqrt(rho) scale: records are 16-byte aligned rows of kp elements
    if constexpr (!std::is_same<T, TI>::value) {      // float32 records into a float64 block
      const int kpv = kp / 4;
      for (int it = tid; it < (streamp ? 0 : cnt * kpv); it += NT) {
        const int j = it / kpv, c = it - j * kpv;
        const float4 v = reinterpret_cast<const float4*>(P.rec + (int64_t)lidx[j] * kp)[c];
        const T wj = lw[j];
        T* o = Yt + (size_t)j * kp + 4 * c;
        o[0] = T(v.x) * wj; o[1] = T(v.y) * wj; o[2] = T(v.z) * wj; o[3] = T(v.w) * wj;
      }
    } else {
      constexpr int VW = 16 / sizeof(T);
      const int kpv = kp / VW;
      using VT = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
      for (int it = tid; it < (streamp ? 0 : cnt * kpv); it += NT) {
        const int j = it / kpv, c = it - j * kpv;
        VT v = reinterpret_cast<const VT*>(P.rec + (int64_t)lidx[j] * kp)[c];
        const T wj = lw[j];
        if constexpr (sizeof(T) == 4) { v.x *= wj; v.y *= wj; v.z *= wj; v.w *= wj; }
        else { v.x *= wj; v.y *= wj; }
        reinterpret_cast<VT*>(Yt + (size_t)j * kp)[c] = v;
      }
    }
    const int ntrue = P.dual ? cnt : k;
    const int n = (ntrue + 1) & ~1;
    if (P.dual && n > cnt) for (int i = tid; i < kp; i += NT) Yt[(size_t)cnt * kp + i] = T(0);
    __syncthreads();
    // ---- Gram matrix (canonical upper triangle) + identity
    for (int it = tid; it < n * n; it += NT) {
      const int a = it / n, b = it - a * n;
      V[a * lda + b] = (a == b) ? T(1) : T(0);
      if (a > b) continue;
      if (streamp) { S[a * lda + b] = T(0); continue; }     // (filled by the streamed tiles below; padding stays zero)
      T acc = T(0);
      if (P.dual) {
        const T* ya = Yt + (size_t)a * kp; const T* yb = Yt + (size_t)b * kp;
        for (int i = 0; i < k; ++i) acc += ya[i] * yb[i];
      } else if (b < k) {
        if (P.kernel_mode == 0) {
          for (int j = 0; j < cnt; ++j) acc += Yt[(size_t)j * kp + a] * Yt[(size_t)j * kp + b];
        } else if (P.kernel_mode == 1) {   // RBF Gram exp(-gamma |y_a - y_b|^2)  (kernels/rbf.py:75-81,110-111)
          for (int j = 0; j < cnt; ++j) { const T df = Yt[(size_t)j * kp + a] - Yt[(size_t)j * kp + b]; acc += df * df; }
          acc = t_exp(-P.gamma * acc);
        } else {   // any other reference kernel or composition (kernels/*.py, base_kernels.py)
          acc = kprog_pair(P.prog, Yt, kp, cnt, a, b, a == b);
        }
      }
      S[a * lda + b] = acc;
    }
    __syncthreads();
    if (streamp) {
      // extended Gram [Yl; d_l][Yl; d_l]^T by 16x16 tiles (ta <= tb) spread over the waves: lane (lr, h) feeds
      // rec[idx(4 s + h)][16 t + lr] * sqrt(rho); column k delivers b = Yl d, entry (k, k) |d_l|^2
      using AccT = typename MfmaAcc<T>::type;
      const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, h = lane >> 4;
      const int tt = (k + 1 + 15) >> 4, ntile = tt * (tt + 1) / 2;
      const int ksteps = (cnt + 3) >> 2;
      for (int i = tid; i < n; i += NT) zb[i] = T(0);
      __syncthreads();
      for (int tile = wv; tile < ntile; tile += NT / 64) {
        int tb_ = 0;
        while ((tb_ + 1) * (tb_ + 2) / 2 <= tile) ++tb_;
        const int ta_ = tile - tb_ * (tb_ + 1) / 2;
        const int ra = 16 * ta_ + lr, rb = 16 * tb_ + lr;
        AccT acc = {T(0), T(0), T(0), T(0)};
        for (int s_ = 0; s_ < ksteps; ++s_) {
          const int j = 4 * s_ + h;
          T av = T(0), bv = T(0);
          if (j < cnt) {
            const T wj = lw[j];
            const TI* rj = P.rec + (int64_t)lidx[j] * kp;
            av = ra <= k ? T(rj[ra]) * wj : T(0);
            bv = rb <= k ? T(rj[rb]) * wj : T(0);
          }
          acc = mfma16(av, bv, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // result rows: float32 tile row 4 h + q, float64 tile row 4 q + h (probed on gfx950); column lr in both
          const int a = 16 * ta_ + (sizeof(T) == 8 ? 4 * q + h : 4 * h + q), b = 16 * tb_ + lr;
          if (b == k && a < k) zb[a] = acc[q];
          if (a == k && b == k) red[2] = acc[q];
          if (a <= b && b < k) S[a * lda + b] = acc[q];
        }
      }
      __syncthreads();
      if (P.kernel_mode == 1) {   // RBF: exp(-gamma (C_aa + C_bb - 2 C_ab)); kernel vector from C_aa + |d|^2 - 2 b_a
        for (int i = tid; i < k; i += NT) uv[i] = S[i * lda + i];
        __syncthreads();
        for (int it = tid; it < k * k; it += NT) {
          const int a = it / k, b = it - a * k;
          if (a <= b) { const T sq = uv[a] + uv[b] - T(2) * S[a * lda + b]; S[a * lda + b] = t_exp(-P.gamma * (sq > T(0) ? sq : T(0))); }
        }
        for (int i = tid; i < k; i += NT) { const T sq = uv[i] + red[2] - T(2) * zb[i]; zb[i] = t_exp(-P.gamma * (sq > T(0) ? sq : T(0))); }
        __syncthreads();
      }
    }
    // ---- right-hand side of the mean weights (primal only; dual uses d directly)
    if (!P.dual) {
      if (P.kernel_mode == 0) {
        for (int i = tid; i < (streamp ? 0 : n); i += NT) {
          T acc = T(0);
          if (i < k) for (int j = 0; j < cnt; ++j) acc += Yt[(size_t)j * kp + i] * Yt[(size_t)j * kp + k];
          zb[i] = acc;
        }
        __syncthreads();
      } else {
        // double centring of K and centring of k(Yb, d)   (core/ketkf.py:77-89)
        for (int i = tid; i < k; i += NT) {
          T acc = T(0);
          for (int j = 0; j < k; ++j) acc += sym(S, lda, i, j);
          uv[i] = acc / T(k);
          if (P.kernel_mode == 1) {
            // (zb = k(Yb, d) came with the streamed Gram)
          } else {
            zb[i] = kprog_pair(P.prog, Yt, kp, cnt, i, k, false);   // k(Yb, d): never "the same sample" (diag.py:65-66)
          }
        }
        __syncthreads();
        if (tid == 0) {
          T gm = T(0), om = T(0);
          for (int i = 0; i < k; ++i) { gm += uv[i]; om += zb[i]; }
          red[0] = gm / T(k); red[1] = om / T(k);
        }
        __syncthreads();
        for (int it = tid; it < k * k; it += NT) {
          const int a = it / k, b = it - a * k;
          if (a <= b) S[a * lda + b] = S[a * lda + b] - uv[b] - (uv[a] - red[0]);
        }
        for (int i = tid; i < n; i += NT) zb[i] = i < k ? zb[i] - red[1] - (uv[i] - red[0]) : T(0);
        __syncthreads();
      }
    }
    // ---- symmetric eigensolve.  S = V (D + E) V^T on return, E = what is left off the diagonal
    //      (relative size <= stop_tol); every matrix function below is evaluated as
    //      f(D + E) = f(D) + F o E + O(E^2),  F_pq = (f(d_p) - f(d_q)) / (d_p - d_q)
    //      (Daleckii-Krein), with the divided differences in cancellation-free closed form.
    int jstat;
    const bool conv = jacobi_sym<T, NT>(S, V, cs, dec, n, n, lda, reg, P.rot_tol2, P.stop_tol2, P.max_sweeps, jstat);
    if (!conv) flag |= MIA_FLAG_NOCONV;
    // ---- per-mode values (clamp >= 0 then + reg: core/utils.py:58-59).  u = sqrt(l + reg)
    const T ar = t_sqrt(reg);
    for (int r = tid; r < n; r += NT) {
      T lam = S[r * lda + r];
      lam = lam > T(0) ? lam : T(0);
      const T le = lam + reg;
      const T u = t_sqrt(le);
      T acc = T(0);
      if (P.dual) {
        gW[r] = (r < ntrue) ? -t_sqrt(km1) / (u * ar * (ar + u)) : T(0);
        for (int b = 0; b < cnt; ++b) acc += V[b * lda + r] * Yt[(size_t)b * kp + k];
      } else {
        gW[r] = (r < ntrue) ? t_sqrt(km1) / u : T(0);
        for (int b = 0; b < k; ++b) acc += V[b * lda + r] * zb[b];
      }
      gM[r] = (r < ntrue) ? T(1) / le : T(0);
      qb[r] = u;        // kept for the divided differences
      sb[r] = acc;      // a = V^T rhs
    }
    __syncthreads();
    // mean term: (D + E + reg)^-1 a  ~=  gM o (a - E (gM o a))        [F_pq = -gM_p gM_q]
    for (int r = tid; r < n; r += NT) {
      T acc = T(0);
      for (int b = 0; b < n; ++b) if (b != r) acc += sym(S, lda, r, b) * gM[b] * sb[b];
      av[r] = gM[r] * (sb[r] - acc);
    }
    __syncthreads();
    // square-root term: off-diagonals of S become F o E in place, gW stays the diagonal
    {
      const T cdual = t_sqrt(km1) / ar;
      for (int it = tid; it < n * n; it += NT) {
        const int a = it / n, b = it - a * n;
        if (a < b) {
          const T ua = qb[a], ub = qb[b];
          T F;
          if (P.dual) F = cdual * (ar + ua + ub) / ((ua + ub) * ua * ub * (ar + ua) * (ar + ub));
          else F = -t_sqrt(km1) / (ua * ub * (ua + ub));
          if (b >= ntrue) F = T(0);
          S[a * lda + b] *= F;
        }
      }
    }
    for (int b = tid; b < n; b += NT) {   // u = V av
      T acc = T(0);
      for (int r = 0; r < n; ++r) acc += V[b * lda + r] * av[r];
      uv[b] = acc;
    }
    __syncthreads();
    if (P.W) {   // w_mean explicitly only for the weights output
      for (int i = tid; i < k; i += NT) {
        T acc;
        if (P.dual) { acc = T(0); for (int b = 0; b < cnt; ++b) acc += Yt[(size_t)b * kp + i] * uv[b]; }
        else acc = uv[i];
        wbar[i] = acc;
      }
    }
    // ---- ensemble transform, one state row at a time
    for (int mi = 0; mi < P.m; ++mi) {
      const TI* xrow = P.X + (int64_t)mi * k * P.ldx + g;
      for (int i = tid; i < k; i += NT) xp[i] = T(xrow[(int64_t)i * P.ldx]);
      __syncthreads();
      // every lane sums the k members itself (LDS broadcast reads): no serial section, no barrier
      T xm = T(0);
      for (int i = 0; i < k; ++i) xm += xp[i];
      xm /= T(k);
      // z = X' B  (dual: B = Yl -> z_b = sum_i x'_i Yl[i][b]; primal: B = I)
      for (int b = tid; b < n; b += NT) {
        T acc = T(0);
        if (P.dual) {
          if (b < cnt) { const T* yb = Yt + (size_t)b * kp; for (int i = 0; i < k; ++i) acc += (xp[i] - xm) * yb[i]; }
        } else acc = b < k ? xp[b] - xm : T(0);
        zb[b] = acc;
      }
      __syncthreads();
      for (int r = tid; r < n; r += NT) {   // zv = V^T z
        T acc = T(0);
        for (int b = 0; b < n; ++b) acc += zb[b] * V[b * lda + r];
        av[r] = acc;
      }
      __syncthreads();
      for (int r = tid; r < n; r += NT) {   // q = (diag(gW) + F o E) zv
        T acc = gW[r] * av[r];
        for (int b = 0; b < n; ++b) if (b != r) acc += sym(S, lda, r, b) * av[b];
        qb[r] = acc;
      }
      __syncthreads();
      T zu = T(0);   // X' w_mean = z . u  (every lane, broadcast reads)
      for (int b = 0; b < n; ++b) zu += zb[b] * uv[b];
      for (int b = tid; b < n; b += NT) {
        T acc = T(0);
        for (int r = 0; r < n; ++r) acc += qb[r] * V[b * lda + r];
        sb[b] = acc;
      }
      __syncthreads();
      const T mterm = xm + zu;
      TI* orow = P.Xa + (int64_t)mi * k * P.ldo + P.o0 + pt;
      for (int j = tid; j < k; j += NT) {
        T acc;
        if (P.dual) { acc = f0 * (xp[j] - xm); for (int b = 0; b < cnt; ++b) acc += sb[b] * Yt[(size_t)b * kp + j]; }
        else acc = sb[j];
        const T out = mterm + acc;
        if (!(t_abs(out) <= T(1e30))) flag |= MIA_FLAG_NONFINITE;
        orow[(int64_t)j * P.ldo] = TI(out);
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
          for (int b = 0; b < cnt; ++b) acc += Yt[(size_t)b * kp + i] * V[b * lda + r];
          Mq[i * lda + r] = acc;
        }
        Mm = Mq;
        __syncthreads();
      }
      TI* wout = P.W + pt * (int64_t)k * k;
      for (int it = tid; it < k * k; it += NT) {
        const int i = it / k, j = it - i * k;
        T acc = wbar[i] + (i == j ? f0 : T(0));
        for (int r = 0; r < n; ++r) acc += gW[r] * Mm[i * lda + r] * Mm[j * lda + r];
        wout[it] = TI(acc);
      }
    }
    if (P.flags) {
      if (tid == 0) iflag[2] = 0;
      __syncthreads();
      if (flag) atomicOr(&iflag[2], flag);
      __syncthreads();
      if (tid == 0) P.flags[pt] = iflag[2] | (jstat << 8);   // bits 8-15 sweeps, 16+ rotating rounds
    }
  }
}

template <typename T>
static size_t wave_lds_bytes(int k, int kp, int p_max, int nmax, int lda, int rows, bool want_mq) {
  size_t e = (size_t)rows * kp + 2 * (size_t)nmax * lda + 8 * (size_t)nmax + 2 * (size_t)k + 8 + ((p_max + 3) & ~1);
  size_t b = e * sizeof(T);
  b += (size_t)((p_max + 3) & ~1) * sizeof(int) + 4 * sizeof(int);
  const int nb = nmax / 2;
  b += (size_t)((nb * (nb - 1) / 2 + 7) & ~7) * sizeof(unsigned short);
  if (want_mq) b += (size_t)k * lda * sizeof(T);
  return align_up(b, 16);
}

template <typename T, typename TI>
static int wave_analysis_launch_io(const TI* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const TI* rec,
                                   const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap,
                                   int p_max, T inf_factor, int kernel_mode, T gamma, TI* Xa, int64_t ldo, int64_t o0,
                                   TI* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream,
                                   const mia_kernel_op_t* prog, int n_ops) {
  WaveParams<T, TI> ap;
  ap.prog.n = 0;
  if (kernel_mode == 2) {
    const int rc = kernel_program_check(prog, n_ops);
    if (rc != MIA_OK) return rc;
    ap.prog.n = n_ops;
    for (int i = 0; i < n_ops; ++i) { ap.prog.op[i] = (unsigned char)prog[i].op; ap.prog.val[i] = T(prog[i].value); }
  }
  ap.only_flagged = only_flagged;
  if (only_flagged && !flags_opt) return MIA_ERR_NULL;
  ap.X = X; ap.ldx = ldx; ap.m = m; ap.k = k; ap.g0 = g0; ap.ng = ng; ap.rec = rec;
  ap.kp = (k + 1 + 3) & ~3;
  ap.cnt = nbr_cnt; ap.idx = nbr_idx; ap.w = nbr_w; ap.p_cap = p_cap; ap.p_max = p_max;
  ap.reg = T(k - 1) / inf_factor;
  ap.Xa = Xa; ap.ldo = ldo; ap.o0 = o0; ap.W = W_opt; ap.flags = flags_opt;
  ap.kernel_mode = kernel_mode; ap.gamma = gamma;
  ap.dual = (kernel_mode == 0 && p_max <= k) ? 1 : 0;
  const int ntrue = ap.dual ? p_max : k;
  ap.nmax = (ntrue + 1) & ~1;
  if (ap.nmax < 2) ap.nmax = 2;
  if (ap.nmax > 510) return MIA_ERR_UNSUPPORTED;   // 8-bit pair ids in the block table
  ap.lda = ap.nmax + 1;
  ap.rows = ap.dual ? ap.nmax : (kernel_mode == 2 ? (p_max > 0 ? p_max : 1) : 0);   // linear / RBF primal: streamed, no block
  ap.max_sweeps = sizeof(T) == 4 ? 16 : 24;
  // stop at sqrt(eps): the first-order correction leaves O(stop_tol^2) = O(eps).  The weights
  // output W uses the diagonal part only, so it asks for full convergence.
  T stop_tol = sizeof(T) == 4 ? T(2.4e-4) : T(1.5e-8);
  if (W_opt) stop_tol = sizeof(T) == 4 ? T(2.4e-7) : T(9e-16);
  MIA_EXP_SET(stop_tol, "MIA_JACOBI_STOP_TOL", (T)atof);
  const T rot_tol = stop_tol * T(0.5);
  ap.stop_tol2 = stop_tol * stop_tol;
  ap.rot_tol2 = rot_tol * rot_tol;
  const size_t lds = wave_lds_bytes<T>(k, ap.kp, p_max, ap.nmax, ap.lda, ap.rows, W_opt != nullptr && ap.dual);
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  const bool big = ap.nmax > 44;
  int ppb = (int)((ng + 16383) / 16384);
  if (ppb < 4) ppb = 4;
  if (ppb > 32) ppb = 32;
  ap.pts_per_block = ppb;
  const int64_t nblk = (ng + ppb - 1) / ppb;
  if (nblk > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  if (big) {
    auto kern = letkf_wave_kernel<T, 256, TI>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nblk), dim3(256), lds, stream>>>(ap);
  } else {
    auto kern = letkf_wave_kernel<T, 64, TI>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nblk), dim3(64), lds, stream>>>(ap);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <typename T>
int wave_analysis_launch(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const T* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap,
                         int p_max, T inf_factor, int kernel_mode, T gamma, T* Xa, int64_t ldo, int64_t o0,
                         T* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream,
                         const mia_kernel_op_t* prog, int n_ops) {
  return wave_analysis_launch_io<T, T>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max, inf_factor, kernel_mode,
                                       gamma, Xa, ldo, o0, W_opt, flags_opt, only_flagged, stream, prog, n_ops);
}

// float32 arrays, float64 arithmetic (ETKF / RBF cores; the redo of declined points).  MIA_ERR_UNSUPPORTED when the float64
// block does not fit the LDS: the caller keeps the float32 kernels then.
int wave_analysis_launch_f32_in_f64(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                                    const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                    float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                    float* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream) {
  return wave_analysis_launch_io<double, float>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, p_cap, p_max,
                                                (double)inf_factor, kernel_mode, (double)gamma, Xa, ldo, o0, W_opt, flags_opt,
                                                only_flagged, stream, nullptr, 0);
}

template int wave_analysis_launch<float>(const float*, int64_t, int, int, int64_t, int64_t, const float*,
                                         const int32_t*, const int32_t*, const double*, int, int, float, int, float,
                                         float*, int64_t, int64_t, float*, int32_t*, int, hipStream_t,
                                         const mia_kernel_op_t*, int);
template int wave_analysis_launch<double>(const double*, int64_t, int, int, int64_t, int64_t, const double*,
                                          const int32_t*, const int32_t*, const double*, int, int, double, int, double,
                                          double*, int64_t, int64_t, double*, int32_t*, int, hipStream_t,
                                          const mia_kernel_op_t*, int);

}  // namespace mia

// Threshold parallel-order (round-robin tournament) cyclic Jacobi eigensolver for a symmetric matrix kept ONCE in
// LDS (canonical upper-triangle addressing), shared by the runtime-order analysis kernel (letkf_wave.hip) and the
// localised IEnKS weight update (ienks.hip).  Stands in for torch.symeig / torch.svd of a symmetric matrix
// (pytassim/core/utils.py:57, :122).
#pragma once
#include "mia_common.h"
#include "mia_jacobi.h"

namespace mia {

__device__ inline int wrap_up(int v, int n1) { return v >= n1 ? v - n1 : v; }
__device__ inline int wrap_dn(int v, int n1) { return v < 0 ? v + n1 : v; }

// pair i of round r in the round-robin tournament on n players (n even): player n-1 stays,
// the others rotate; every unordered pair meets exactly once in n-1 rounds.
__device__ inline void pair_of(int r, int i, int n1, int& p, int& q) {
  p = wrap_up(r + i, n1);
  q = (i == 0) ? n1 : wrap_dn(r - i, n1);
}

template <typename T>
__device__ inline T& sym(T* S, int lda, int x, int y) {   // canonical (upper) element of a symmetric matrix
  const int lo = x < y ? x : y, hi = x < y ? y : x;
  return S[lo * lda + hi];
}

// rotation (c, s) annihilating a_pq; returns t = tan(theta).  float: hardware rcp/rsq/sqrt
// (1 ulp) followed by one normalisation step so that c^2 + s^2 = 1 to rounding; double: IEEE ops.
__device__ inline float rot_params(float app, float aqq, float apq, float& c, float& s) {
  const float tau = (aqq - app) * 0.5f * __builtin_amdgcn_rcpf(apq);
  const float at = __builtin_fabsf(tau);
  float t = __builtin_amdgcn_rcpf(at + __builtin_amdgcn_sqrtf(1.0f + tau * tau));
  t = tau < 0.0f ? -t : t;
  c = __builtin_amdgcn_rsqf(1.0f + t * t);
  s = t * c;
  const float corr = 1.5f - 0.5f * (c * c + s * s);
  c *= corr; s *= corr;
  return t;
}
__device__ inline double rot_params(double app, double aqq, double apq, double& c, double& s) {
  const double tau = (aqq - app) / (2.0 * apq);
  const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  c = 1.0 / sqrt(1.0 + t * t);
  s = t * c;
  return t;
}

// Threshold parallel-order Jacobi, symmetric matrix in canonical upper storage.
// V (nv rows x n) accumulates rotations: S_in = V S_out V^T.  Pairs whose relative off-diagonal
// |a_pq| / sqrt((|a_pp|+shift)(|a_qq|+shift)) is below rot_tol are left alone; sweeping stops as
// soon as every off-diagonal is below stop_tol (the caller corrects for what is left to first
// order, so stop_tol ~ sqrt(eps) already gives eps-level results).  stat: bits 0-7 sweeps,
// bits 8+ rounds that rotated.  Returns false when max_sweeps was hit first.
template <typename T, int NT>
__device__ bool jacobi_sym(T* S, T* V, T* cs, const unsigned short* dec, int n, int nv, int lda, T shift,
                           T rot_tol2, T stop_tol2, int max_sweeps, int& stat) {
  stat = 0;
  if (n < 2) return true;
  const int tid = threadIdx.x;
  const int nb = n >> 1, n1 = n - 1;
  const int noff = nb * (nb - 1) / 2;
  // eigenvector mapping: lane -> (pair vj, first row vr0), rows advance by vstep per pass
  const int vstep = NT / nb;
  const int vj = tid % nb, vr0 = tid / nb;
  const bool vact = vr0 < vstep;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // ---- stopping rule: largest relative off-diagonal element
    int big = 0;
    for (int it = tid; it < n * n; it += NT) {
      const int a = it / n, b = it - a * n;
      if (a < b) {
        const T e = S[a * lda + b];
        big |= (e * e > stop_tol2 * (t_abs(S[a * lda + a]) + shift) * (t_abs(S[b * lda + b]) + shift)) ? 1 : 0;
      }
    }
    if (!__syncthreads_or(big)) return true;
    stat += 1;
    for (int r = 0; r < n1; ++r) {
      // ---- step 1: rotation of every pair + its diagonal block
      int rot = 0;
      for (int i = tid; i < nb; i += NT) {
        int p, q;
        pair_of(r, i, n1, p, q);
        const T app = S[p * lda + p], aqq = S[q * lda + q];
        T& rpq = sym(S, lda, p, q);
        const T apq = rpq;
        T c = T(1), s = T(0);
        if (apq * apq > rot_tol2 * (t_abs(app) + shift) * (t_abs(aqq) + shift)) {
          const T t = rot_params(app, aqq, apq, c, s);
          S[p * lda + p] = app - t * apq;
          S[q * lda + q] = aqq + t * apq;
          rpq = T(0);
          rot = 1;
        }
        cs[2 * i] = c; cs[2 * i + 1] = s;
      }
      if (!__syncthreads_or(rot)) continue;   // nothing to rotate in this round (uniform)
      stat += 1 << 8;
      // ---- step 2a: off-diagonal 2x2 blocks  B <- R1^T B R2
      for (int it = tid; it < noff; it += NT) {
        const int d = dec[it];
        const int bi = d & 0xff, bj = d >> 8;
        const T c1 = cs[2 * bi], s1 = cs[2 * bi + 1], c2 = cs[2 * bj], s2 = cs[2 * bj + 1];
        if (s1 == T(0) && s2 == T(0)) continue;
        int p1, q1, p2, q2;
        pair_of(r, bi, n1, p1, q1);
        pair_of(r, bj, n1, p2, q2);
        T& e00 = sym(S, lda, p1, p2); T& e01 = sym(S, lda, p1, q2);
        T& e10 = sym(S, lda, q1, p2); T& e11 = sym(S, lda, q1, q2);
        const T a00 = e00, a01 = e01, a10 = e10, a11 = e11;
        const T b00 = c1 * a00 - s1 * a10, b01 = c1 * a01 - s1 * a11;
        const T b10 = s1 * a00 + c1 * a10, b11 = s1 * a01 + c1 * a11;
        e00 = c2 * b00 - s2 * b01; e01 = s2 * b00 + c2 * b01;
        e10 = c2 * b10 - s2 * b11; e11 = s2 * b10 + c2 * b11;
      }
      // ---- step 2b: eigenvectors  V <- V J
      if (vact) {
        const T c2 = cs[2 * vj], s2 = cs[2 * vj + 1];
        if (s2 != T(0)) {
          int p2, q2;
          pair_of(r, vj, n1, p2, q2);
          for (int row = vr0; row < nv; row += vstep) {
            const T v0 = V[row * lda + p2], v1 = V[row * lda + q2];
            V[row * lda + p2] = c2 * v0 - s2 * v1;
            V[row * lda + q2] = s2 * v0 + c2 * v1;
          }
        }
      }
      __syncthreads();
    }
  }
  return false;
}

}  // namespace mia

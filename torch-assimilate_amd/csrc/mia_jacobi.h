// Workgroup-cooperative symmetric eigensolver in LDS (parallel-order cyclic Jacobi), shared by
// the local (LETKF) and global (ETKF) kernels.  Stands in for torch.symeig at
// pytassim/core/utils.py:57.
#pragma once
#include "mia_common.h"

namespace mia {

template <typename T> __device__ inline T t_sqrt(T x);
template <> __device__ inline float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ inline double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ inline T t_abs(T x) { return x < T(0) ? -x : x; }
template <typename T> __device__ inline T t_exp(T x);
template <> __device__ inline float t_exp<float>(float x) { return expf(x); }
template <> __device__ inline double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ inline T t_pow(T x, T y);
template <> __device__ inline float t_pow<float>(float x, float y) { return powf(x, y); }
template <> __device__ inline double t_pow<double>(double x, double y) { return pow(x, y); }
template <typename T> __device__ inline T t_tanh(T x);
template <> __device__ inline float t_tanh<float>(float x) { return tanhf(x); }
template <> __device__ inline double t_tanh<double>(double x) { return tanh(x); }
template <typename T> __device__ inline T t_sin(T x);
template <> __device__ inline float t_sin<float>(float x) { return sinf(x); }
template <> __device__ inline double t_sin<double>(double x) { return sin(x); }

// Parallel cyclic Jacobi on the n x n symmetric matrix A (LDS, leading dim lda), n even.
// V (nv rows) accumulates the rotations: A_in = V A_out V^T.  Returns true when a sweep
// needed no rotation.  Must be called by all NT threads of the workgroup.
template <typename T, int NT>
__device__ bool jacobi_lds(T* A, T* V, T* cs, int* iflag, int n, int nv, int lda, T shift,
                           T tol, int max_sweeps) {
  const int tid = threadIdx.x;
  const int nb = n >> 1;
  if (n < 2) return true;
  bool converged = false;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    if (tid == 0) iflag[sweep & 1] = 0;
    // (the barrier inside the first round orders this store before any flag update)
    for (int r = 0; r < n - 1; ++r) {
      // --- rotation parameters, one pair per thread
      __syncthreads();
      for (int i = tid; i < nb; i += NT) {
        int p, q;
        if (i == 0) { p = r; q = n - 1; }
        else { p = (r + i) % (n - 1); q = (r - i + (n - 1)) % (n - 1); }
        if (p > q) { int t = p; p = q; q = t; }
        const T app = A[p * lda + p], aqq = A[q * lda + q], apq = A[p * lda + q];
        T c = T(1), s = T(0);
        const T thr = tol * t_sqrt((t_abs(app) + shift) * (t_abs(aqq) + shift));
        if (t_abs(apq) > thr) {
          const T tau = (aqq - app) / (T(2) * apq);
          const T t = (tau >= T(0) ? T(1) : T(-1)) / (t_abs(tau) + t_sqrt(T(1) + tau * tau));
          c = T(1) / t_sqrt(T(1) + t * t);
          s = t * c;
          iflag[sweep & 1] = 1;
        }
        cs[2 * i] = c; cs[2 * i + 1] = s;
      }
      __syncthreads();
      // --- A <- J^T A J, one 2x2 block per work item
      for (int it = tid; it < nb * nb; it += NT) {
        const int bi = it / nb, bj = it - bi * nb;
        int p1, q1, p2, q2;
        if (bi == 0) { p1 = r; q1 = n - 1; } else { p1 = (r + bi) % (n - 1); q1 = (r - bi + (n - 1)) % (n - 1); }
        if (bj == 0) { p2 = r; q2 = n - 1; } else { p2 = (r + bj) % (n - 1); q2 = (r - bj + (n - 1)) % (n - 1); }
        if (p1 > q1) { int t = p1; p1 = q1; q1 = t; }
        if (p2 > q2) { int t = p2; p2 = q2; q2 = t; }
        const T c1 = cs[2 * bi], s1 = cs[2 * bi + 1], c2 = cs[2 * bj], s2 = cs[2 * bj + 1];
        const T a00 = A[p1 * lda + p2], a01 = A[p1 * lda + q2];
        const T a10 = A[q1 * lda + p2], a11 = A[q1 * lda + q2];
        // rows: (p1, q1) <- (c1*p1 - s1*q1, s1*p1 + c1*q1)
        const T b00 = c1 * a00 - s1 * a10, b01 = c1 * a01 - s1 * a11;
        const T b10 = s1 * a00 + c1 * a10, b11 = s1 * a01 + c1 * a11;
        // cols: (p2, q2) <- (c2*p2 - s2*q2, s2*p2 + c2*q2)
        T r00 = c2 * b00 - s2 * b01, r01 = s2 * b00 + c2 * b01;
        T r10 = c2 * b10 - s2 * b11, r11 = s2 * b10 + c2 * b11;
        if (bi == bj && s1 != T(0)) { r01 = T(0); r10 = T(0); }
        A[p1 * lda + p2] = r00; A[p1 * lda + q2] = r01;
        A[q1 * lda + p2] = r10; A[q1 * lda + q2] = r11;
      }
      // --- V <- V J
      for (int it = tid; it < nb * nv; it += NT) {
        const int row = it / nb, bj = it - row * nb;
        int p2, q2;
        if (bj == 0) { p2 = r; q2 = n - 1; } else { p2 = (r + bj) % (n - 1); q2 = (r - bj + (n - 1)) % (n - 1); }
        if (p2 > q2) { int t = p2; p2 = q2; q2 = t; }
        const T c2 = cs[2 * bj], s2 = cs[2 * bj + 1];
        const T v0 = V[row * lda + p2], v1 = V[row * lda + q2];
        V[row * lda + p2] = c2 * v0 - s2 * v1;
        V[row * lda + q2] = s2 * v0 + c2 * v1;
      }
    }
    __syncthreads();
    if (iflag[sweep & 1] == 0) { converged = true; break; }
  }
  __syncthreads();
  return converged;
}

}  // namespace mia

// Third-generation fused LETKF analysis kernel for gfx950 (float32): "systolic" Jacobi.
//
// Mathematics and reference citations as in letkf_wave.hip / letkf_entry.hip.  What changes
// is the eigensolver's data movement, because rocprofv3 showed the second-generation kernel to
// be VALU-issue bound on index arithmetic (22.5k VALU instructions per analysis, ~2/3 of them
// integer):
//
//  * the matrix order is a template parameter NMAX (the launch picks the smallest bucket that
//    holds p_max or k; smaller local problems are zero padded), so every stride is a constant;
//  * S and V are kept in TOURNAMENT-SLOT order (Brent & Luk): pair i of a round always sits in
//    slots (2i, 2i+1).  A lane therefore owns the same 2x2 block of S (or the same column pair
//    of V) in every round: it reads FIXED LDS addresses, rotates, and writes the result to the
//    FIXED addresses where the round-robin permutation wants it for the next round.  The
//    per-round index arithmetic of the earlier kernels disappears altogether;
//  * all reads of a round precede all writes; with one wavefront per workgroup (NT = 64) the
//    in-order LDS queue makes that safe without any s_barrier, larger orders use NT = 256 and
//    three barriers per round;
//  * after n-1 rounds the permutation is the identity again, so eigenvalues/eigenvectors come
//    out in natural order at every sweep boundary (the only place the loop can stop).
#include <cstdlib>
#include "mia_common.h"
#include "mia_kernels.h"
#include "mia_options.h"

namespace mia {

struct SysParams {
  const float* X; int64_t ldx; int m; int k;
  int64_t g0, ng;
  const float* rec; int kp;
  const int32_t* cnt; const int32_t* idx; const double* w; int p_cap; int p_max;
  float reg; float* Xa; int64_t ldo, o0; float* W; int32_t* flags;
  int dual; int rows; int pts_per_block; int max_sweeps; float rot_tol2, stop_tol2;
  int kernel_mode; float gamma;
  int only_flagged;   // process only points whose flags[] carry MIA_FLAG_RETRY
};

// round-robin successor of a slot (pair i = slots 2i, 2i+1; slot 0 never moves)
__host__ __device__ constexpr int slot_next(int s, int nb) {
  if (nb < 2 || s == 0) return s;
  if (s == 1) return 2;
  if ((s & 1) == 0) return (s == 2 * nb - 2) ? 2 * nb - 1 : s + 2;
  return s - 2;
}

template <int LDA>
__device__ inline int canon(int a, int b) { return a < b ? a * LDA + b : b * LDA + a; }

__device__ inline void rot_params_f(float app, float aqq, float apq, float& c, float& s, float& t) {
  const float tau = (aqq - app) * 0.5f * __builtin_amdgcn_rcpf(apq);
  t = __builtin_amdgcn_rcpf(__builtin_fabsf(tau) + __builtin_amdgcn_sqrtf(1.0f + tau * tau));
  t = tau < 0.0f ? -t : t;
  c = __builtin_amdgcn_rsqf(1.0f + t * t);
  s = t * c;
  const float corr = 1.5f - 0.5f * (c * c + s * s);   // c^2 + s^2 = 1 to rounding
  c *= corr; s *= corr;
}

#define MIA_LDS_ORDER() asm volatile("" ::: "memory")

using f32x4 = __attribute__((ext_vector_type(4))) float;

// strictly-upper element (a < b) number `it` in b-major order: it = b(b-1)/2 + a
__device__ inline void tri_decode(int it, int& a, int& b) {
  b = (int)((1.0f + __builtin_amdgcn_sqrtf(1.0f + 8.0f * (float)it)) * 0.5f);
  while (b * (b - 1) / 2 > it) --b;
  while ((b + 1) * b / 2 <= it) ++b;
  a = it - b * (b - 1) / 2;
}

// D = A B^T on the matrix cores, 16x16 output tiles distributed over the waves of the workgroup, K = 4 * k4.
// a_at(row, kk) / b_at(col, kk) deliver the operands (zero outside the matrices), store(row, col, value) the result.
// v_mfma_f32_16x16x4_f32: lane l feeds A[16 ti + (l & 15)][4 s + (l >> 4)] and B[16 tj + (l & 15)][4 s + (l >> 4)] and
// receives D[16 ti + 4 (l >> 4) + q][16 tj + (l & 15)], q = 0..3.
template <typename FA, typename FB, typename FD>
__device__ __forceinline__ void mfma_abt(int rows, int cols, int k4, int wave, int nwave, int lane, FA a_at, FB b_at, FD store) {
  const int lr = lane & 15, h = lane >> 4;
  const int tr = (rows + 15) >> 4, tc = (cols + 15) >> 4;
  for (int tile = wave; tile < tr * tc; tile += nwave) {
    const int ti = tile / tc, tj = tile - ti * tc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s_ = 0; s_ < k4; ++s_)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_at(16 * ti + lr, 4 * s_ + h), b_at(16 * tj + lr, 4 * s_ + h), acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) store(16 * ti + 4 * h + q, 16 * tj + lr, acc[q]);
  }
}

template <int NMAX, int NT>
__global__ __launch_bounds__(NT, (NT == 64 ? (NMAX >= 32 ? 3 : 4) : 2)) void letkf_sys_kernel(SysParams P) {
  // row stride: a multiple of 4 (float4 rows) that is NOT a multiple of 8, so that consecutive rows
  // start in different LDS banks (stride 64 put every row of a column on one bank)
  constexpr int NB = NMAX / 2, LDA = (NMAX % 8 == 0) ? NMAX + 4 : NMAX, NOFF = NB * (NB - 1) / 2;
  constexpr int SP = NOFF == 0 ? 1 : (NOFF + NT - 1) / NT;
  constexpr int VSTEP = NT / NB, VP = (NMAX + VSTEP - 1) / VSTEP;
  constexpr int NTRI = NMAX * (NMAX - 1) / 2, TP = (NTRI + NT - 1) / NT;
  constexpr int N4 = NMAX / 4;
  constexpr int UNR = N4 <= 6 ? N4 : 2;   // full unrolling of the matvec loops only for small orders (register pressure)
  constexpr bool MULTIWAVE = NT > 64;
  constexpr int NWAVE = NT / 64;
  constexpr int TT = (NMAX + 15) / 16, NTILE = TT * (TT + 1) / 2;
  static_assert(NMAX % 4 == 0 && NB <= NT, "order must be a multiple of 4 and fit the workgroup");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = P.k, kp = P.kp, pm = P.p_max;
  float* S = reinterpret_cast<float*>(smem_raw);    // [NMAX][LDA] canonical upper / slot order in the sweeps
  // rotations of the round, one private copy per wavefront: every wave derives all NB rotations itself
  // (reads only), so no barrier is needed between deriving and using them
  float2* cs2 = reinterpret_cast<float2*>(S + NMAX * LDA) + (size_t)wave * NB;   // [NWAVE][NB]
  float* gWs = reinterpret_cast<float*>(reinterpret_cast<float2*>(S + NMAX * LDA) + (size_t)NWAVE * NB);  // [NMAX]
  float* tv = gWs + NMAX;                           // [NMAX] scratch vectors, all float4-readable
  float* uq = tv + NMAX;
  float* hq = uq + NMAX;
  float* zb = hq + NMAX;
  float* qb = zb + NMAX;
  float* sb = qb + NMAX;
  float* uvs = sb + NMAX;
  float* red = uvs + NMAX;                          // [8]
  int* iflag = reinterpret_cast<int*>(red + 8);     // [4]
  float* xp = reinterpret_cast<float*>(iflag + 4);  // [kp] centred state row (float4-readable)
  float* wbar = xp + kp;                            // [kp]
  float* zall = wbar + kp;                          // [m][NMAX]  z = X' B for every state row
  float* Yt = zall + (size_t)P.m * NMAX;            // [rows][kp] obs-major: yb[0..k), d, pad
  // The local block is dead once the Gram matrix, the right-hand side and z exist, and V is born only
  // then: they share one region unless the weights output needs both at once (P.W).
  float* V = P.W ? Yt + (size_t)P.rows * kp : Yt;   // [NMAX][LDA] rows natural, columns slot order
  const size_t ureg = P.W ? (size_t)P.rows * kp + NMAX * LDA
                          : ((size_t)P.rows * kp > (size_t)NMAX * LDA ? (size_t)P.rows * kp : (size_t)NMAX * LDA);
  float* lw = Yt + ureg;                            // [pm + 2]
  int* lidx = reinterpret_cast<int*>(lw + ((pm + 3) & ~1));   // [pm + 2]
  float* Mq = reinterpret_cast<float*>(lidx + ((pm + 3) & ~1));  // [k][LDA] (W on the dual route)
  float* Nq = Mq + (P.dual ? (size_t)k * LDA : 0);               // [max(k, NMAX)][LDA] (W output)

  // ------------------------------------------------------------------ fixed lane roles
  int rd[SP], wr00[SP], wr01[SP], wr10[SP], wr11[SP], cbi[SP], cbj[SP];
#pragma unroll
  for (int sp = 0; sp < SP; ++sp) {
    const int it = tid + sp * NT;
    int bi, bj;
    tri_decode(it, bi, bj);
    cbi[sp] = bi; cbj[sp] = bj;
    rd[sp] = (2 * bi) * LDA + 2 * bj;
    const int r0 = slot_next(2 * bi, NB), r1 = slot_next(2 * bi + 1, NB);
    const int c0 = slot_next(2 * bj, NB), c1 = slot_next(2 * bj + 1, NB);
    wr00[sp] = canon<LDA>(r0, c0); wr01[sp] = canon<LDA>(r0, c1);
    wr10[sp] = canon<LDA>(r1, c0); wr11[sp] = canon<LDA>(r1, c1);
  }
  const int dgi = MULTIWAVE ? lane : tid;   // pair handled in phase 1 (every wave when MULTIWAVE)
  const int dg_rd = (2 * dgi) * LDA + 2 * dgi;
  const int dg_s0 = slot_next(2 * dgi, NB), dg_s1 = slot_next(2 * dgi + 1, NB);
  const int dg_w0 = dg_s0 * LDA + dg_s0, dg_w1 = dg_s1 * LDA + dg_s1, dg_we = canon<LDA>(dg_s0, dg_s1);
  const int vj = tid % NB, vr0 = tid / NB;
  const bool vact = vr0 < VSTEP && !(P.max_sweeps & 256);   // bit 8: timing experiment without V
  const int vw0 = slot_next(2 * vj, NB), vw1 = slot_next(2 * vj + 1, NB);
  // strictly-upper elements owned by this lane (stopping rule, first-order correction)
  int ta[TP], tb[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) tri_decode(tid + t * NT, ta[t], tb[t]);

  const float km1 = float(k - 1);
  const float reg = P.reg;
  const float f0 = P.dual ? sqrtf(km1 / reg) : 0.0f;
  const float ar = sqrtf(reg);
  const int KS = (k + 3) >> 2;          // K steps of the 16x16x4 MFMA Gram

  // one grid point per workgroup: no point loop, so nothing per-lane is hoisted and kept live
  // across phases (with a loop the compiler precomputed ~100 address registers and spilled them)
  // XCD-aware block -> point map: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
  // share an L2), so giving XCD x the contiguous point range x*ng/8.. keeps the records shared by
  // neighbouring grid points in ONE L2 and lets the partial-line X / Xa accesses of consecutive points
  // merge there (speed / traffic only - any placement is correct).
  const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (bid >= P.ng) return;
  const int64_t q8 = P.ng >> 3, r8 = P.ng & 7, xcd = bid & 7;
  const int64_t pt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (P.only_flagged && !(P.flags[pt] & MIA_FLAG_RETRY)) return;   // wave-uniform
  {
    const int64_t g = P.g0 + pt;
    const int cnt = P.cnt[pt];
    int flag = 0;
    if (cnt > pm || cnt > P.p_cap || (P.dual ? cnt : k) > NMAX) {   // loud failure, never truncate
      if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
      const float nanv = __builtin_nanf("");
      for (int it = tid; it < P.m * k; it += NT) P.Xa[(int64_t)it * P.ldo + P.o0 + pt] = nanv;
      if (P.W) for (int it = tid; it < k * k; it += NT) P.W[pt * (int64_t)k * k + it] = nanv;
      return;
    }
    const int xskip = P.max_sweeps >> 9;   // timing experiments only (MIA_EXPERIMENT_SKIP), 0 in production
    // first state row: issue the (strided, latency-bound) loads now, consume after the eigensolve
    float xval = 0.0f;
    if (tid < k) xval = P.X[(int64_t)tid * P.ldx + g];
    for (int j = tid; j < cnt; j += NT) {
      lidx[j] = P.idx[pt * P.p_cap + j];
      lw[j] = float(P.w[pt * P.p_cap + j]);
    }
    __syncthreads();
    // ---- gather + sqrt(rho) scale (wrapper.py:91-97): a scaled float4 copy of whole records
    {
      const int kpv = kp >> 2;
      for (int it = tid; it < ((xskip & 1) ? 0 : cnt * kpv); it += NT) {
        const int j = it / kpv, c = it - j * kpv;
        float4 v = reinterpret_cast<const float4*>(P.rec + (int64_t)lidx[j] * kp)[c];
        const float wj = lw[j];
        v.x *= wj; v.y *= wj; v.z *= wj; v.w *= wj;
        reinterpret_cast<float4*>(Yt + (size_t)j * kp)[c] = v;
      }
    }
    const int ntrue = P.dual ? cnt : k;
    __syncthreads();
    // ---- Gram matrix, canonical upper storage, zero padded to NMAX
    if (xskip & 2) {
      for (int it = tid; it < NMAX * LDA; it += NT) S[it] = 0.0f;
    } else if (P.dual) {
      // S = Yl^T Yl on the matrix cores: v_mfma_f32_16x16x4_f32 (exact f32), one 16x16 tile of S per
      // accumulator, K = members.  A[row][kk] = Yt[16*ta + (lane&15)][KS*(lane>>4) + s] (any K order
      // sums the same Gram entry), B is the same map on the column tile.
      const int lr = lane & 15, h = lane >> 4;
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        if (tile % NWAVE != wave) continue;
        int tb_ = 0;
        while ((tb_ + 1) * (tb_ + 2) / 2 <= tile) ++tb_;
        const int ta_ = tile - tb_ * (tb_ + 1) / 2;            // ta_ <= tb_
        const int ra = 16 * ta_ + lr, rb = 16 * tb_ + lr;
        const float* pa = Yt + (size_t)ra * kp + KS * h;
        const float* pb = Yt + (size_t)rb * kp + KS * h;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s_ = 0; s_ < KS; ++s_) {
          const bool kin = KS * h + s_ < k;
          const float av_ = (ra < cnt && kin) ? pa[s_] : 0.0f;
          const float bv_ = (rb < cnt && kin) ? pb[s_] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, bv_, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = 16 * ta_ + h * 4 + q, b = 16 * tb_ + lr;   // D[row = 4*(lane>>4)+q][col = lane&15]
          if (a <= b && b < NMAX) S[a * LDA + b] = acc[q];
        }
      }
    } else {
      for (int it = tid; it < NMAX * (NMAX + 1) / 2; it += NT) {
        int b = (int)((__builtin_amdgcn_sqrtf(1.0f + 8.0f * (float)it) - 1.0f) * 0.5f);
        while (b * (b + 1) / 2 > it) --b;
        while ((b + 1) * (b + 2) / 2 <= it) ++b;
        const int a = it - b * (b + 1) / 2;
        float acc = 0.0f;
        if (b < k) {
          if (P.kernel_mode == 0) {
            for (int j = 0; j < cnt; ++j) acc += Yt[(size_t)j * kp + a] * Yt[(size_t)j * kp + b];
          } else {   // RBF Gram exp(-gamma |y_a - y_b|^2)  (kernels/rbf.py:75-81,110-111)
            for (int j = 0; j < cnt; ++j) { const float df = Yt[(size_t)j * kp + a] - Yt[(size_t)j * kp + b]; acc += df * df; }
            acc = __expf(-P.gamma * acc);
          }
        }
        S[a * LDA + b] = acc;
      }
    }
    __syncthreads();
    // ---- right-hand side of the mean weights -> tv (dual: d itself; primal: Yl d or centred k(Yb, d))
    if (P.dual) {
      for (int b = tid; b < NMAX; b += NT) tv[b] = b < cnt ? Yt[(size_t)b * kp + k] : 0.0f;
    } else if (P.kernel_mode == 0) {
      for (int i = tid; i < NMAX; i += NT) {
        float acc = 0.0f;
        if (i < k) for (int j = 0; j < cnt; ++j) acc += Yt[(size_t)j * kp + i] * Yt[(size_t)j * kp + k];
        tv[i] = acc;
      }
    } else {   // double centring of K and centring of k(Yb, d)   (core/ketkf.py:77-89)
      for (int i = tid; i < k; i += NT) {
        float acc = 0.0f;
        for (int j = 0; j < k; ++j) acc += S[canon<LDA>(i, j)];
        uq[i] = acc / float(k);
        float ko = 0.0f;
        for (int j = 0; j < cnt; ++j) { const float df = Yt[(size_t)j * kp + i] - Yt[(size_t)j * kp + k]; ko += df * df; }
        tv[i] = __expf(-P.gamma * ko);
      }
      __syncthreads();
      if (tid == 0) {
        float gm = 0.0f, om = 0.0f;
        for (int i = 0; i < k; ++i) { gm += uq[i]; om += tv[i]; }
        red[0] = gm / float(k); red[1] = om / float(k);
      }
      __syncthreads();
      for (int it = tid; it < k * k; it += NT) {
        const int a = it / k, b = it - a * k;
        if (a <= b) S[a * LDA + b] = S[a * LDA + b] - uq[b] - (uq[a] - red[0]);
      }
      for (int i = tid; i < NMAX; i += NT) tv[i] = i < k ? tv[i] - red[1] - (uq[i] - red[0]) : 0.0f;
    }
    __syncthreads();
    // ---- z = X' B for every state row while the local block is still in LDS
    //      (dual: B = Yl -> z_b = sum_i x'_i Yl[i][b]; primal: B = I)
    {
      const int k4 = (k + 3) >> 2;
      for (int mi = 0; mi < P.m; ++mi) {
        if (mi > 0) { xval = 0.0f; if (tid < k) xval = P.X[((int64_t)mi * k + tid) * P.ldx + g]; }
        float xm;
        if (!MULTIWAVE) xm = wave_sum_dpp(xval) / float(k);
        else {
          if (tid < kp) xp[tid] = tid < k ? xval : 0.0f;
          __syncthreads();
          xm = 0.0f;
          for (int i = 0; i < k; ++i) xm += xp[i];
          xm /= float(k);
          __syncthreads();
        }
        if (tid < kp) xp[tid] = tid < k ? xval - xm : 0.0f;
        __syncthreads();
        if (tid < NMAX) {
          const int b = tid;
          float z_b = 0.0f;
          if (P.dual) {
            if (b < cnt) {
              const float4* yb = reinterpret_cast<const float4*>(Yt + (size_t)b * kp);
              const float4* x4 = reinterpret_cast<const float4*>(xp);   // xp is 0 beyond k: d / pad drop out
              for (int i = 0; i < k4; ++i) { const float4 y = yb[i], x = x4[i]; z_b += y.x * x.x + y.y * x.y + y.z * x.z + y.w * x.w; }
            }
          } else z_b = b < k ? xp[b] : 0.0f;
          zall[mi * NMAX + b] = z_b;
        }
        __syncthreads();
      }
    }
    // ---- V = I (overwrites the local block unless the weights output keeps it)
    for (int it = tid; it < NMAX * N4; it += NT) {
      const int a = it / N4, c4 = (it - a * N4) * 4;
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a == c4) e.x = 1.f; else if (a == c4 + 1) e.y = 1.f; else if (a == c4 + 2) e.z = 1.f; else if (a == c4 + 3) e.w = 1.f;
      reinterpret_cast<float4*>(V + a * LDA)[c4 >> 2] = e;
    }
    __syncthreads();

    // ================================================================ systolic Jacobi
    int sweeps = 0;
    bool conv = false;
#pragma unroll 1
    for (; sweeps < (P.max_sweeps & 255); ++sweeps) {
      // stopping rule: largest relative off-diagonal element (natural order at sweep boundaries)
      int big = 0;
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        if (tid + t * NT < NTRI) {
          const float e = S[ta[t] * LDA + tb[t]];
          big |= (e * e > P.stop_tol2 * (fabsf(S[ta[t] * LDA + ta[t]]) + reg) * (fabsf(S[tb[t] * LDA + tb[t]]) + reg)) ? 1 : 0;
        }
      }
      if (!__syncthreads_or(big)) { conv = true; break; }
#pragma unroll 1
      for (int r = 0; r < NMAX - 1; ++r) {
        // ---- phase 1: rotation of each pair from its diagonal block
        float nd0 = 0.0f, nd1 = 0.0f, nde = 0.0f;
        if (dgi < NB) {
          const float2 de = *reinterpret_cast<const float2*>(S + dg_rd);
          const float d1 = S[dg_rd + LDA + 1];
          float c = 1.0f, s = 0.0f, t = 0.0f;
          nde = de.y;   // pair left alone: its off-diagonal element travels on unchanged
          if (de.y * de.y > P.rot_tol2 * (fabsf(de.x) + reg) * (fabsf(d1) + reg)) { rot_params_f(de.x, d1, de.y, c, s, t); nde = 0.0f; }
          nd0 = de.x - t * de.y;
          nd1 = d1 + t * de.y;
          cs2[dgi] = make_float2(c, s);
        }
        MIA_LDS_ORDER();   // same-wave LDS write -> read: in-order, no barrier
        // ---- phase 2: every read of the round, rotate in registers
        float o00[SP], o01[SP], o10[SP], o11[SP];
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) {
          if (tid + sp * NT < NOFF) {
            const float2 r1 = cs2[cbi[sp]], r2 = cs2[cbj[sp]];
            const float2 a0 = *reinterpret_cast<const float2*>(S + rd[sp]);
            const float2 a1 = *reinterpret_cast<const float2*>(S + rd[sp] + LDA);
            const float b00 = r1.x * a0.x - r1.y * a1.x, b01 = r1.x * a0.y - r1.y * a1.y;
            const float b10 = r1.y * a0.x + r1.x * a1.x, b11 = r1.y * a0.y + r1.x * a1.y;
            o00[sp] = r2.x * b00 - r2.y * b01; o01[sp] = r2.y * b00 + r2.x * b01;
            o10[sp] = r2.x * b10 - r2.y * b11; o11[sp] = r2.y * b10 + r2.x * b11;
          }
        }
        float v0[VP], v1[VP];
        if (vact) {
          const float2 r2 = cs2[vj];
#pragma unroll
          for (int vp = 0; vp < VP; ++vp) {
            const int row = vr0 + vp * VSTEP;
            if (row < NMAX) {
              const float2 x = *reinterpret_cast<const float2*>(V + row * LDA + 2 * vj);
              v0[vp] = r2.x * x.x - r2.y * x.y;
              v1[vp] = r2.y * x.x + r2.x * x.y;
            }
          }
        }
        if (MULTIWAVE) __syncthreads(); else MIA_LDS_ORDER();
        // ---- phase 3: every write of the round, to the slots of the NEXT round
        if (tid < NB) { S[dg_w0] = nd0; S[dg_w1] = nd1; S[dg_we] = nde; }
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) {
          if (tid + sp * NT < NOFF) {
            S[wr00[sp]] = o00[sp]; S[wr01[sp]] = o01[sp]; S[wr10[sp]] = o10[sp]; S[wr11[sp]] = o11[sp];
          }
        }
        if (vact) {
#pragma unroll
          for (int vp = 0; vp < VP; ++vp) {
            const int row = vr0 + vp * VSTEP;
            if (row < NMAX) { V[row * LDA + vw0] = v0[vp]; V[row * LDA + vw1] = v1[vp]; }
          }
        }
        if (MULTIWAVE) __syncthreads(); else MIA_LDS_ORDER();
      }
      __syncthreads();
    }
    if (!conv) flag |= MIA_FLAG_NOCONV;

    // ---- per-mode values (clamp >= 0 then + reg: core/utils.py:58-59), u = sqrt(l + reg).
    //      S = V (D + E) V^T with E the residual off-diagonals (relative size <= stop_tol):
    //      f(D + E) = f(D) + F o E + O(E^2), F_pq = (f(d_p) - f(d_q)) / (d_p - d_q)  (Daleckii-Krein)
    float gM_r = 0.0f, a_r = 0.0f;
    if (!(xskip & 4)) {
      if (tid < NMAX) {
        const int r = tid;
        float lam = S[r * LDA + r];
        lam = lam > 0.0f ? lam : 0.0f;
        const float le = lam + reg;
        const float u = __builtin_amdgcn_sqrtf(le);
        const bool live = r < ntrue;
        float gw;
        if (P.dual) gw = -sqrtf(km1) / (u * ar * (ar + u));
        else gw = sqrtf(km1) / u;
        gM_r = live ? 1.0f / le : 0.0f;
        // a = V^T rhs : column r of V (stride LDA, conflict-free across lanes) against the broadcast rhs
        float acc = 0.0f;
#pragma unroll UNR
        for (int b4 = 0; b4 < N4; ++b4) {
          const float4 t4 = reinterpret_cast<const float4*>(tv)[b4];
          acc += V[(4 * b4 + 0) * LDA + r] * t4.x + V[(4 * b4 + 1) * LDA + r] * t4.y +
                 V[(4 * b4 + 2) * LDA + r] * t4.z + V[(4 * b4 + 3) * LDA + r] * t4.w;
        }
        a_r = acc;
        gWs[r] = live ? gw : 0.0f;
        uq[r] = u;
        hq[r] = live ? 1.0f / (u * (ar + u)) : 0.0f;     // dual divided-difference helper
      }
      __syncthreads();
      if (tid < NMAX) tv[tid] = gM_r * a_r;               // (rhs no longer needed)
      __syncthreads();
      // mean term: (D + E + reg)^-1 a ~= gM o (a - E (gM o a))      [F_pq = -gM_p gM_q]
      if (tid < NMAX) {
        const int r = tid;
        float acc = 0.0f;
#pragma unroll 4
        for (int b = 0; b < NMAX; ++b) if (b != r) acc += S[canon<LDA>(r, b)] * tv[b];
        zb[r] = gM_r * (a_r - acc);                        // av
      }
      __syncthreads();
      // square-root term: S becomes the FULL symmetric matrix  diag(gW) + F o E
      {
        const float cdual = sqrtf(km1) / ar;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
          if (tid + t * NT < NTRI) {
            const int a = ta[t], b = tb[t];
            const float ua = uq[a], ub = uq[b];
            float F;
            if (P.dual) F = cdual * (ar + ua + ub) * hq[a] * hq[b] * __builtin_amdgcn_rcpf(ua + ub);
            else F = (b < ntrue) ? -sqrtf(km1) * __builtin_amdgcn_rcpf(ua * ub * (ua + ub)) : 0.0f;
            const float e = S[a * LDA + b] * F;
            S[a * LDA + b] = e; S[b * LDA + a] = e;
          }
        }
        if (tid < NMAX) S[tid * LDA + tid] = gWs[tid];
      }
      // u = V av : row b of V (contiguous) against the broadcast av
      if (tid < NMAX) {
        const int b = tid;
        float acc = 0.0f;
#pragma unroll UNR
        for (int r4 = 0; r4 < N4; ++r4) {
          const float4 vv = reinterpret_cast<const float4*>(V + b * LDA)[r4];
          const float4 a4 = reinterpret_cast<const float4*>(zb)[r4];
          acc += vv.x * a4.x + vv.y * a4.y + vv.z * a4.z + vv.w * a4.w;
        }
        uvs[b] = acc;
      }
    }
    __syncthreads();
    if (P.W) {   // w_mean explicitly only for the weights output
      for (int i = tid; i < k; i += NT) {
        float acc;
        if (P.dual) { acc = 0.0f; for (int b = 0; b < cnt; ++b) acc += Yt[(size_t)b * kp + i] * uvs[b]; }
        else acc = uvs[i];
        wbar[i] = acc;
      }
    }
    // ---- ensemble transform (interface/base.py:257-278), one state row at a time
    for (int mi = 0; mi < ((xskip & 8) ? 0 : P.m); ++mi) {
      xval = 0.0f;
      if (tid < k) xval = P.X[((int64_t)mi * k + tid) * P.ldx + g];     // L2-resident re-read
      float zv_r = 0.0f, z_b = 0.0f;
      if (tid < NMAX) {   // zv = V^T z
        const int r = tid;
        z_b = zall[mi * NMAX + r];
#pragma unroll UNR
        for (int b4 = 0; b4 < N4; ++b4) {
          const float4 z4 = reinterpret_cast<const float4*>(zall + mi * NMAX)[b4];
          zv_r += V[(4 * b4 + 0) * LDA + r] * z4.x + V[(4 * b4 + 1) * LDA + r] * z4.y +
                  V[(4 * b4 + 2) * LDA + r] * z4.z + V[(4 * b4 + 3) * LDA + r] * z4.w;
        }
        tv[r] = zv_r;
      }
      __syncthreads();
      if (tid < NMAX) {   // q = (diag(gW) + F o E) zv : row r of the symmetric matrix against the broadcast zv
        const int r = tid;
        float acc = 0.0f;
#pragma unroll UNR
        for (int b4 = 0; b4 < N4; ++b4) {
          const float4 s4 = reinterpret_cast<const float4*>(S + r * LDA)[b4];
          const float4 z4 = reinterpret_cast<const float4*>(tv)[b4];
          acc += s4.x * z4.x + s4.y * z4.y + s4.z * z4.z + s4.w * z4.w;
        }
        qb[r] = acc;
      }
      __syncthreads();
      float zu, xm;   // X' w_mean = z . u ; ensemble mean of this row
      if (!MULTIWAVE) {
        zu = wave_sum_dpp(tid < NMAX ? z_b * uvs[tid] : 0.0f);
        xm = wave_sum_dpp(xval) / float(k);
      } else {
        zu = 0.0f;
        for (int b = 0; b < NMAX; ++b) zu += zall[mi * NMAX + b] * uvs[b];
        if (tid < kp) xp[tid] = tid < k ? xval : 0.0f;
        __syncthreads();
        xm = 0.0f;
        for (int i = 0; i < k; ++i) xm += xp[i];
        xm /= float(k);
      }
      if (tid < NMAX) {   // s = V q, pre-multiplied by the localisation weight of its observation
        const int b = tid;
        float acc = 0.0f;
#pragma unroll UNR
        for (int r4 = 0; r4 < N4; ++r4) {
          const float4 vv = reinterpret_cast<const float4*>(V + b * LDA)[r4];
          const float4 q4 = reinterpret_cast<const float4*>(qb)[r4];
          acc += vv.x * q4.x + vv.y * q4.y + vv.z * q4.z + vv.w * q4.w;
        }
        sb[b] = P.dual ? (b < cnt ? acc * lw[b] : 0.0f) : acc;
      }
      __syncthreads();
      const float mterm = xm + zu;
      float* orow = P.Xa + (int64_t)mi * k * P.ldo + P.o0 + pt;
      for (int j = tid; j < k; j += NT) {
        float acc;
        if (P.dual) {   // Yl s straight from the (L2-resident) records: member j is contiguous across lanes
          acc = f0 * (xval - xm);
          for (int b = 0; b < cnt; ++b) acc += sb[b] * P.rec[(int64_t)lidx[b] * kp + j];
        } else acc = sb[j];
        const float out = mterm + acc;
        if (!(fabsf(out) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
        orow[(int64_t)j * P.ldo] = out;
      }
      __syncthreads();
    }
    // ---- optional weights output (what estimate_weights returns, letkf.py:145-146):
    //      W_ij = w_mean_i + f0 delta_ij + (M Q M^T)_ij,  M = B V,  Q = diag(gW) + F o E  (S holds Q by now, so the
    //      first-order treatment of the residual off-diagonals E carries over and the sweeps stop at the same
    //      tolerance as for the transform).  Three small products on the matrix cores (v_mfma_f32_16x16x4_f32,
    //      exact f32): the scalar version spent ~2000 LDS reads per lane here, more than the rest of the kernel.
    if (P.W) {
      const float* Mm = V;
      if (P.dual) {      // M = Yl V : (k x cnt) (cnt x NMAX)
        mfma_abt(k, NMAX, N4, wave, NWAVE, lane,
                 [&](int i, int b) { return (i < k && b < cnt) ? Yt[(size_t)b * kp + i] : 0.0f; },
                 [&](int r, int b) { return (r < NMAX && b < cnt) ? V[b * LDA + r] : 0.0f; },
                 [&](int i, int r, float v) { if (i < k && r < NMAX) Mq[i * LDA + r] = v; });
        Mm = Mq;
        __syncthreads();
      }
      const int mrows = P.dual ? k : NMAX;           // rows of M (primal: M = V, padded rows are zero columns of Q)
      mfma_abt(mrows, NMAX, N4, wave, NWAVE, lane,   // N = M Q  (Q symmetric: row b of S is column b)
               [&](int i, int r) { return i < mrows ? Mm[i * LDA + r] : 0.0f; },
               [&](int b, int r) { return b < NMAX ? S[b * LDA + r] : 0.0f; },
               [&](int i, int b, float v) { if (i < mrows && b < NMAX) Nq[i * LDA + b] = v; });
      __syncthreads();
      float* wout = P.W + pt * (int64_t)k * k;
      mfma_abt(k, k, N4, wave, NWAVE, lane,          // W = N M^T + w_mean 1^T + f0 I
               [&](int i, int r) { return i < k ? Nq[i * LDA + r] : 0.0f; },
               [&](int j, int r) { return j < k ? Mm[j * LDA + r] : 0.0f; },
               [&](int i, int j, float v) { if (i < k && j < k) wout[i * k + j] = wbar[i] + (i == j ? f0 : 0.0f) + v; });
    }
    if (P.flags) {
      if (tid == 0) iflag[2] = 0;
      __syncthreads();
      if (flag) atomicOr(&iflag[2], flag);
      __syncthreads();
      if (tid == 0) P.flags[pt] = iflag[2] | (sweeps << 8) | ((sweeps * (NMAX - 1)) << 16);
    }
  }
}

static size_t sys_lds_bytes(int k, int kp, int m, int p_max, int nmax, int rows, bool want_w, bool dual) {
  const int lda = (nmax % 8 == 0) ? nmax + 4 : nmax;
  const size_t yt = (size_t)rows * kp, vv = (size_t)nmax * lda;
  const size_t ureg = want_w ? yt + vv : (yt > vv ? yt : vv);
  const int nwave = nmax > 32 ? 4 : 1;
  size_t e = (size_t)nmax * lda + (size_t)nwave * nmax /*cs2*/ + 8 * (size_t)nmax + 8 + 2 * (size_t)kp + (size_t)m * nmax + ureg +
             ((p_max + 3) & ~1);
  size_t b = e * sizeof(float) + 4 * sizeof(int) + (size_t)((p_max + 3) & ~1) * sizeof(int);
  if (want_w) b += (size_t)((dual ? k : 0) + (k > nmax ? k : nmax)) * lda * sizeof(float);   // Mq (dual) + Nq
  return align_up(b, 16);
}

template <int NMAX, int NT>
static int sys_launch(const SysParams& ap, size_t lds, dim3 grid, hipStream_t stream) {
  auto kern = letkf_sys_kernel<NMAX, NT>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<grid, dim3(NT), lds, stream>>>(ap);
  if (!ap.only_flagged) note_analysis_kernel("letkf_sys_kernel<%d, %d>", NMAX, NT);      // (the redo of declined points is not a step's analysis kernel)
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// returns MIA_ERR_UNSUPPORTED when no bucket fits (caller falls back to the runtime-order kernel)
int sys_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                        const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                        float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                        float* W_opt, int32_t* flags_opt, int only_flagged, hipStream_t stream) {
  SysParams ap;
  ap.only_flagged = only_flagged;
  if (only_flagged && !flags_opt) return MIA_ERR_NULL;
  ap.X = X; ap.ldx = ldx; ap.m = m; ap.k = k; ap.g0 = g0; ap.ng = ng; ap.rec = rec;
  ap.kp = (k + 1 + 3) & ~3;
  ap.cnt = nbr_cnt; ap.idx = nbr_idx; ap.w = nbr_w; ap.p_cap = p_cap; ap.p_max = p_max;
  ap.reg = float(k - 1) / inf_factor;
  ap.Xa = Xa; ap.ldo = ldo; ap.o0 = o0; ap.W = W_opt; ap.flags = flags_opt;
  ap.kernel_mode = kernel_mode; ap.gamma = gamma;
  ap.dual = (kernel_mode == 0 && p_max <= k) ? 1 : 0;
  const int ntrue = ap.dual ? p_max : k;
  static const int buckets[] = {4, 8, 12, 16, 20, 24, 32, 40, 48, 64};
  int nmax = 0;
  for (int b : buckets) if (b >= ntrue) { nmax = b; break; }
  if (nmax == 0) return MIA_ERR_UNSUPPORTED;
  ap.rows = ap.dual ? nmax : (p_max > 0 ? p_max : 1);
  ap.max_sweeps = 16;
  MIA_EXP_SET(ap.max_sweeps, "MIA_MAX_SWEEPS", atoi);
  if (MIA_EXP_FLAG("MIA_EXPERIMENT_NO_V")) ap.max_sweeps |= 256;
  { int skip = 0; MIA_EXP_SET(skip, "MIA_EXPERIMENT_SKIP", atoi); ap.max_sweeps |= skip << 9; }
  // stop at ~sqrt(eps): the first-order correction (also applied to the weights output) leaves O(stop_tol^2)
  float stop_tol = 1.0e-3f;   // 4*sqrt(eps): second-order remainder ~1e-6
  // weights output on the primal route (KETKF): the mean weights of a centred kernel matrix are small differences of
  // O(1) terms, and W is judged entry by entry, not through an analysis it is applied to: one more (quadratically
  // converging) sweep puts the remainder at ~1e-9 (measured at 1e-3: 4e-5 relative error against the reference)
  if (W_opt && !ap.dual) stop_tol = 3.0e-5f;
  MIA_EXP_SET(stop_tol, "MIA_JACOBI_STOP_TOL", (float)atof);
  const float rot_tol = 1e-7f;
  ap.stop_tol2 = stop_tol * stop_tol;
  ap.rot_tol2 = rot_tol * rot_tol;
  const size_t lds = sys_lds_bytes(k, ap.kp, m, p_max, nmax, ap.rows, W_opt != nullptr, ap.dual != 0);
  if ((k > 64 && nmax <= 32) || k > 256) return MIA_ERR_UNSUPPORTED;   // one member per lane
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  // one grid point per workgroup (measured on MI355X, C2: 1-2 points per workgroup beat 4-10 by 5-12 %:
  // the dispatcher's dynamic placement balances the data-dependent sweep counts)
  ap.pts_per_block = 1;
  const int64_t gx = ng < 65536 ? ng : 65536;
  const int64_t gy = (ng + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)gx, (unsigned)gy);
  switch (nmax) {
    case 4: return sys_launch<4, 64>(ap, lds, grid, stream);
    case 8: return sys_launch<8, 64>(ap, lds, grid, stream);
    case 12: return sys_launch<12, 64>(ap, lds, grid, stream);
    case 16: return sys_launch<16, 64>(ap, lds, grid, stream);
    case 20: return sys_launch<20, 64>(ap, lds, grid, stream);
    case 24: return sys_launch<24, 64>(ap, lds, grid, stream);
    case 32: return sys_launch<32, 64>(ap, lds, grid, stream);
    case 40: return sys_launch<40, 256>(ap, lds, grid, stream);
    case 48: return sys_launch<48, 256>(ap, lds, grid, stream);
    case 64: return sys_launch<64, 256>(ap, lds, grid, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

}  // namespace mia

// Fused LETKF analysis without an explicit eigendecomposition: matrix functions applied to
// vectors by Chebyshev expansion ("matfun" route), float32, one wavefront per grid point.
//
// What the reference computes per grid point (core/etkf.py:57-103 + interface/base.py:257-278)
// depends on the local matrix only through two FUNCTIONS of it,
//     w_mean = (C + reg)^-1 (Yl d),      W = sqrt(k-1) (C + reg)^-1/2,     C = Yl Yl^T, reg = (k-1)/inf
// (eigenvalue clamp and shift of core/utils.py:57-60 included: C is positive semi-definite), and the
// analysis of a state row x only needs their action on ONE vector:
//     xa = mean + x' w_mean + x' W .
// SURVEY.md section 7 ("Hard parts") notes that W and Pa being functions of A alone makes
// eigensolver-free evaluations valid.  On the dual route (p <= k, see letkf_entry.hip) with
// S = Yl^T Yl, z = Yl^T x' :
//     x' W      = f0 x' + Yl (phi(S) z),        phi(l) = -sqrt(k-1) / (u a (a + u)),  u = sqrt(l+reg), a = sqrt(reg)
//     x' w_mean = d_l . (psi(S) z),             psi(l) = 1 / (l + reg)
// (primal route / RBF-KETKF: S = C or the centred kernel matrix, z = x', phi(l) = sqrt(k-1)/u).
// phi(S) z and psi(S) z share ONE three-term Chebyshev recurrence t_{j+1} = 2 A t_j - t_{j-1} on
// A = 2 S / L - I, where L >= lambda_max is the Gershgorin bound of S; both functions are analytic on
// [0, L] with their nearest singularity at -reg, so the truncation error decays like rho^-d with
// rho = (sqrt(1+L/reg)+1)/(sqrt(1+L/reg)-1): the degree d is fixed a priori per grid point from L/reg
// (no convergence loop).  Cost O(d n^2) per state row instead of O(sweeps n^3) for the Jacobi
// eigensolver, so this route is used when few state rows are transformed and the weights matrix is
// not requested; grid points whose spectrum would need d > d_max are flagged MIA_FLAG_RETRY and
// redone by the eigensolver kernel.
//
// Mapping: lane r holds row r of S in registers (n <= 64); each recurrence step is n FMAs per lane
// against the broadcast vector (LDS float4 reads of one address = broadcast), one LDS write of the new
// vector.  The Gram matrix comes from the matrix cores (v_mfma_f32_16x16x4_f32) as in letkf_sys.hip.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "mia_common.h"
#include "mia_localize_dev.h"
#include "mia_kernels.h"
#include "mia_options.h"

namespace mia {

// phase-skipping hooks of the timing experiments: compiled out of the production kernels (they cost SALU and registers)
#ifdef MIA_EXPERIMENTS
#define MIA_XSKIP(P) ((P).xskip)
#else
#define MIA_XSKIP(P) 0
#endif

struct ChebParams {
  const float* X; int64_t ldx; int m; int k;
  int64_t g0, ng;
  const float* rec; int kp;
  const int32_t* cnt; const int32_t* idx; const double* w; int p_cap; int p_max;
  float reg; float* Xa; int64_t ldo, o0; int32_t* flags;
  int dual; int rows; int dmax; float log_tol;
  int kernel_mode; float gamma;
  int32_t* retry_count;
  // fused localisation: the wavefront scans the observation index itself instead of reading lists
  int fused; ScanParams scan; int32_t* stats;
  int xskip;   // timing experiments only (MIA_EXPERIMENT_SKIP, -DMIA_EXPERIMENTS builds); the production kernels read 0
  int lds_per_wave;
  // segmented launch (native step driver): the ng points are seg_len-sized segments; segment s writes
  // its own (m*k, seg_len) buffer at Xa + s * seg_stride and counts its finished points in done[s*64 + ..]
  int seg_len; int64_t seg_stride; int32_t* done;
  int kpv_magic;   // ceil(2^20 / (kp / 4))
  float* W;        // weights-output variant: [ng][k][k]
  // IEnKS update through the weights variant (tau = 1): 0 off, 1 transform (valid while Wp = I), 2 bundle (D = Yl / eps)
  int ienks; const float* ienks_Win; int64_t ienks_wstride; float ienks_inv_eps;
  // launch-uniform scalars, prepared on the host (an IEEE division or square root is ~10 VALU instructions per
  // wavefront, and the kernel is VALU-issue bound): 1/reg, sqrt(reg), sqrt((k-1)/reg), sqrt(k-1), 1/k
  float inv_reg, sqrt_reg, f0_dual, sqrt_km1, inv_k;
  // coefficient table (see CoefTable below); tab_hdr == nullptr: coefficients are computed in the kernel
  const int2* tab_hdr; const float2* tab_c; float cs_phi, cs_psi;
};

// ---- Chebyshev coefficients from a table ------------------------------------------------------------------------
// In the scaled variable t = lambda / reg both weight functions are universal up to a factor,
//     dual    phi = sqrt(k-1) reg^-3/2 * [ -1 / (sqrt(t+1) (1 + sqrt(t+1))) ],    psi = reg^-1 * [ 1 / (t+1) ]
//     primal  phi = sqrt(k-1) reg^-1/2 * [  1 / sqrt(t+1) ],
// so their Chebyshev coefficients on [0, T] depend on T alone.  The spectral bound T = L / reg of a grid point is rounded
// UP to the next point of a geometric grid (32 per octave, 2^-24 .. 2^8: an expansion on a larger interval stays valid
// and costs < 2.2 % of interval, i.e. ~1 % of degree) and the coefficients of that grid point are read from a table built
// once per device, route and truncation target -- in float64, by the same Gauss-node cosine transform.  In the kernel
// the degree selection (sqrt, two reciprocals, log) and the transform (N samples, N x N cosines spread over the wave)
// were ~160 of the ~840 VALU instructions of a C2 analysis; the lookup is a v_log, one 8-byte and one 512-byte read.
// (kTabPerOctave, kTabIdx0, kTabN, kTabDeg: mia_kernels.h -- the tile kernel reads the same tables)

__global__ __launch_bounds__(64) void cheb_table_kernel(int2* hdr, float2* c, int dual, double log_tol, int margin) {
  __shared__ double fs[kTabDeg][2];
  const int idx = blockIdx.x, tid = threadIdx.x;
  const double T = exp2(double(idx - kTabIdx0) / double(kTabPerOctave));
  const double sq = sqrt(1.0 + T);
  const double rho = (sq + 1.0) / fmax(sq - 1.0, 1e-12);
  double dd = ceil(log_tol / log(rho)) + (double)margin;
  dd = dd < 3.0 ? 3.0 : (dd > 32767.0 ? 32767.0 : dd);
  const int deg = (int)dd;
  if (tid == 0) hdr[idx] = make_int2(deg, __float_as_int((float)(2.0 / T)));
  c[(size_t)idx * kTabDeg + tid] = make_float2(0.0f, 0.0f);
  if (deg > kTabDeg - 1) return;                      // the kernel declines such points (eigensolver route)
  const int N = deg + 1;
  if (tid < N) {
    const double x = cospi((tid + 0.5) / double(N));
    const double u = sqrt(0.5 * T * (x + 1.0) + 1.0);   // sqrt(t + 1)
    fs[tid][0] = dual ? -1.0 / (u * (1.0 + u)) : 1.0 / u;
    fs[tid][1] = 1.0 / (u * u);
  }
  __syncthreads();
  if (tid < N) {
    double a0 = 0.0, a1 = 0.0;
    for (int i = 0; i < N; ++i) {
      const double cs = cospi(double((long long)tid * (2 * i + 1) % (4LL * N)) / double(2 * N));
      a0 += fs[i][0] * cs; a1 += fs[i][1] * cs;
    }
    const double sc = (tid == 0 ? 1.0 : 2.0) / double(N);
    c[(size_t)idx * kTabDeg + tid] = make_float2((float)(a0 * sc), (float)(a1 * sc));
  }
}

struct CoefTable { int device; int dual; float log_tol; int margin; int2* hdr; float2* c; };
// nullptr pair when the table cannot be had (allocation failure, stream being captured): the kernel then computes
// its coefficients itself.  Built synchronously on first use (one 1024-workgroup launch, ~0.1 ms, then a wait for
// that stream): afterwards the table is immutable and visible to every stream of the device.
// margin: degrees added to the a-priori count log_tol / log(rho) (2 everywhere but the kernelised tile route, whose spectra
// are bounded by the ensemble size: lketkf_tile.hip)
static const CoefTable* cheb_coef_table(int dual, float log_tol, hipStream_t stream, int margin = 2) {
  static std::mutex mu;
  static std::vector<CoefTable*> tabs;
  if (!option(MIA_OPT_CHEB_TABLE)) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  for (const CoefTable* t : tabs)
    if (t->device == dev && t->dual == dual && t->log_tol == log_tol && t->margin == margin) return t;
  if (tabs.size() >= 64) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  CoefTable* t = new CoefTable{dev, dual, log_tol, margin, nullptr, nullptr};
  if (hipMalloc((void**)&t->hdr, sizeof(int2) * kTabN) != hipSuccess ||
      hipMalloc((void**)&t->c, sizeof(float2) * kTabN * kTabDeg) != hipSuccess) {
    (void)hipGetLastError();
    if (t->hdr) (void)hipFree(t->hdr);
    delete t;
    return nullptr;
  }
  cheb_table_kernel<<<dim3(kTabN), dim3(64), 0, stream>>>(t->hdr, t->c, dual, (double)log_tol, margin);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(t->hdr); (void)hipFree(t->c);
    delete t;
    return nullptr;
  }
  tabs.push_back(t);
  return t;
}

// the dual-route table at the default truncation target, for the kernels of other translation units (letkf_tile2.hip)
bool cheb_dual_table(hipStream_t stream, const int2** hdr, const float2** c) {
  const CoefTable* t = cheb_coef_table(1, 12.0f, stream);
  if (!t) return false;
  *hdr = t->hdr; *c = t->c;
  return true;
}

// the primal-route table (1/sqrt(1+t), 1/(1+t)) at the default truncation target (lketkf_tile.hip)
bool cheb_primal_table(hipStream_t stream, const int2** hdr, const float2** c) {
  // degree = the a-priori count alone: the kernel matrix has entries in (0, 1], its spectrum lies in [0, k] and the scaled
  // bound T = L / reg never exceeds ~1.2 (rho >= 5): tools/lk_stress.py over k 5 .. 40, gamma 0.01 .. 10, observation
  // strength x 0.1 .. 10 measures 1.4e-7 worst with two extra degrees, 3.9e-7 with none (gate 1e-5)
  int margin = 0;
  MIA_EXP_SET(margin, "MIA_LK_MARGIN", atoi);
  const CoefTable* t = cheb_coef_table(0, 12.0f, stream, margin);
  if (!t) return false;
  *hdr = t->hdr; *c = t->c;
  return true;
}

// v_rcp_f32 (1 ulp) where a correctly rounded quotient buys nothing: degree selection, interval scale, function samples
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

using f32x4c = __attribute__((ext_vector_type(4))) float;

// one row of S (registers, kept as float2 pairs) against the broadcast vector in LDS.  Written on 2-vectors
// so that hipcc emits v_pk_fma_f32 with operands already in place (the scalar form compiled to v_pk_mul /
// v_pk_add plus ~17 v_mov per step: 45 VALU instead of 12); two accumulators for instruction-level parallelism
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NMAX>
__device__ inline float matvec_row(const f2v (&srow2)[NMAX / 2], const float* tv) {
  f2v acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
#pragma unroll
  for (int b4 = 0; b4 < NMAX / 4; ++b4) {
    const f4v v = reinterpret_cast<const f4v*>(tv)[b4];
    acc0 = srow2[2 * b4] * v.xy + acc0;
    acc1 = srow2[2 * b4 + 1] * v.zw + acc1;
  }
  const f2v a = acc0 + acc1;
  return a.x + a.y;
}

// wave-local ordering of LDS traffic: all LDS operations of this wavefront issued so far have completed
// and the compiler may not move memory operations across (the waves of a workgroup are independent here)
#define MIA_WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// v + (v of lane ^ 32) and v + (v of lane ^ 16) through gfx950's row-swap instructions (v_permlane32_swap exchanges
// the upper half of its first operand with the lower half of the second, v_permlane16_swap the odd rows of the first
// with the even rows of the second: with both operands equal the two results are the two halves / row pairs, each
// spread over the wave).  A ds_bpermute-based __shfl_xor is ~10 instructions, this is 2.
__device__ __forceinline__ float add_xor32(float v) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float add_xor16(float v) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// Chebyshev coefficients (phi_j, psi_j), j = 0 .. deg, of the two weight functions on [0, L]: samples at the deg + 1
// Gauss nodes, then a cosine transform spread over the wavefront.  c2 / f2s: LDS scratch of 64 pairs each.
__device__ __forceinline__ void cheb_coefficients(int tid, int deg, float L, float reg, float sqrt_km1, float ar, bool dual,
                                                  f2v* f2s, f2v* c2) {
  const int N = deg + 1;
  {
    const float invN = fast_rcp(float(N));
    if (tid < N) {
      const float x = __builtin_amdgcn_cosf(float(2 * tid + 1) * 0.25f * invN);     // cos(pi (i+1/2)/N), argument in turns
      const float lam = 0.5f * L * (x + 1.0f);
      const float le = lam + reg;
      const float u = __builtin_amdgcn_sqrtf(le);
      f2v f;
      f.x = dual ? -sqrt_km1 * fast_rcp(u * ar * (ar + u)) : sqrt_km1 * fast_rcp(u);
      f.y = fast_rcp(le);
      f2s[tid] = f;
    }
    MIA_WAVE_SYNC();
    {
      // The N x N cosine transform is spread over the WHOLE wavefront: lane = (coefficient j, part), every part
      // sums a slice of the nodes and the parts are added with cross-lane moves.  (One lane per coefficient left
      // 48 of 64 lanes idle through N quarter-rate v_cos -- the most expensive stretch of the kernel at C2.)
      // cos(j pi (i+1/2)/N) = cos(2 pi m / 4N), m = j (2i+1) mod 4N kept as an exact integer phase
      // (a float phase accumulated over i loses ~N*eps turns: 2e-5 in the analysis at degree 40)
      const int sh = N <= 16 ? 4 : (N <= 32 ? 5 : 6);            // lanes per part = 2^sh >= N
      const int j = tid & ((1 << sh) - 1), part = tid >> sh;
      const int chunk = (N + (64 >> sh) - 1) >> (6 - sh);          // nodes per part
      const int i0 = part * chunk;
      const int i1 = i0 + chunk < N ? i0 + chunk : N;
      const int n4 = 4 * N;
      const float inv4n = 0.25f * invN;
      f2v a = {0.0f, 0.0f};
      if (j < N) {
        const int x = j * (2 * i0 + 1);                             // < 2^13: exact in float
        int ph = x - (int)(float(x) * inv4n) * n4;                  // x mod 4N, up to one period off
        ph = ph < 0 ? ph + n4 : (ph >= n4 ? ph - n4 : ph);
        for (int i = i0; i < i1; ++i) {
          const float c = __builtin_amdgcn_cosf(float(ph) * inv4n);
          ph += 2 * j;
          ph = ph >= n4 ? ph - n4 : ph;
          a = f2s[i] * c + a;
        }
      }
      if (sh <= 5) { a.x = add_xor32(a.x); a.y = add_xor32(a.y); }
      if (sh == 4) { a.x = add_xor16(a.x); a.y = add_xor16(a.y); }
      if (tid < N) {
        const float sc = (tid == 0 ? 1.0f : 2.0f) * invN;
        c2[tid] = a * sc;
      }
    }
    MIA_WAVE_SYNC();
  }
}

// Do the coefficient tables (2 x 64 pairs) fit into the storage of S (full matrix, or the 16-row staging panel of the
// streamed orders)?  Used by the kernel (layout) and by cheb_lds_bytes (size): keep them in step.
__host__ __device__ constexpr bool cheb_alias_tables(int nmax) {
  const int lda = (nmax % 8 == 0) ? nmax + 4 : nmax;
  return nmax <= 32 && nmax * lda >= 4 * 64;      // (orders above 32 may stream S on the dual route and not on the primal)
}

// Compact LDS layout of the scalar-rows variant on the dual route (orders 20 .. 32): everything but the local block Yt
// takes its turn in the storage of S --
//   until the Gram product:  neighbour list (weights, indices) and the all-zero panel row,
//   Gram product -> rows:    S,
//   afterwards:              coefficient tables (2 x 64 pairs), recurrence vector, phi(S) z, 8 scalars, x'
// and the right-hand side is read from Yt's innovation column where it is needed.  C2 (order 20, k = 40): 5120 B per
// wavefront instead of 6000 -- four LDS allocation granules of 1280 B instead of five, i.e. 32 instead of 25 wavefronts
// per CU as far as LDS goes (the 70 VGPRs allow 28).  Used by the kernel (layout) and cheb_lds_bytes (size).
__host__ __device__ constexpr bool cheb_compact_layout(int nmax, int kp, int p_max, bool dual, bool fused) {
  const int lda = (nmax % 8 == 0) ? nmax + 4 : nmax;
  const int pmr = (p_max + 3) & ~1;
  return dual && !fused && nmax >= 20 && nmax <= 32 && 4 * 64 + 2 * nmax + 8 + kp <= nmax * lda && 2 * pmr + kp <= nmax * lda;
}

constexpr int kRowBatch = 16;     // state rows per MFMA batch of the many-rows variant (one 16-column tile)

template <int NMAX, int KL, bool FUSED, int WPB, bool SEG, int MODE = 0>   // MODE 1: many state rows, 2: weights output
__device__ __forceinline__ void letkf_cheb_point(const ChebParams& P) {
  constexpr int LDA = (NMAX % 8 == 0) ? NMAX + 4 : NMAX;
  constexpr int N4 = NMAX / 4;
  constexpr int TT = (NMAX + 15) / 16, NTILE = TT * (TT + 1) / 2;
  constexpr int DCAP = 64;                       // storage for Chebyshev coefficients
  static_assert(NMAX % 4 == 0 && NMAX <= 64, "order must be a multiple of 4, one matrix row per lane");

  // WPB wavefronts per workgroup, each with its own grid point and its own LDS slice (no barrier between
  // them): 1e5 single-wave workgroups cost ~0.11 ms in dispatch alone on MI355X (measured with all phases
  // skipped), four points per workgroup cut that to a quarter
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = P.k, kp = P.kp, pm = P.p_max;
  // Orders above 32 on the dual route: S never exists as a whole in LDS -- its rows go from the MFMA accumulators to
  // the lanes' registers one 16-row panel at a time through a 16 x LDA staging area.  At order 64 (config 4: k = 80,
  // 64 local observations) the full matrix was 17 KB of the 42 KB a wavefront needed, which capped a CU at 3 waves.
  constexpr bool STREAM_OK = NMAX > 32;
  const bool stream_s = STREAM_OK && P.dual;
  float* S = reinterpret_cast<float*>(smem_raw + (size_t)wave * P.lds_per_wave);    // [NMAX][LDA] full symmetric | [16][LDA] staging
  const bool compact = MODE == 0 && cheb_compact_layout(NMAX, kp, pm, P.dual != 0, FUSED);   // (see cheb_compact_layout)
  const int pmr = (pm + 3) & ~1;
  float* tv = compact ? S + 4 * DCAP : S + (stream_s ? 16 : NMAX) * LDA;     // [NMAX] recurrence vector (broadcast source)
  float* rhs = tv + NMAX;                           // [NMAX]   (compact: not there -- Yt's innovation column is read instead)
  float* uq = rhs + NMAX;                           // [NMAX] scratch (RBF centring; primal route only)
  // Scalar-rows variant: the coefficient tables live where S was -- S is dead once its rows sit in the lanes'
  // registers, before the first coefficient is computed -- which is 1 KB of the 7 KB a C2 wavefront needs
  constexpr bool ALIAS_C = MODE == 0 && cheb_alias_tables(NMAX);
  f2v* c2 = reinterpret_cast<f2v*>(ALIAS_C ? S : uq + NMAX);   // [DCAP] Chebyshev coefficients (phi_j, psi_j)
  f2v* f2s = c2 + DCAP;                             // [DCAP] function samples (phi, psi) at the Chebyshev nodes
  float* red = compact ? tv + 2 * NMAX : (ALIAS_C ? uq + NMAX : reinterpret_cast<float*>(f2s + DCAP));   // [8]
  float* xp = red + 8;                              // [kp]
  float* sw = compact ? tv + NMAX : xp + kp;        // [NMAX] phi(S) z
  float* Yt = compact ? S + NMAX * LDA : sw + NMAX; // [rows][kp]
  float* lw = compact ? S : Yt + (size_t)(P.rows + 1) * kp;       // [pm + 2]   (Yt has one extra, all-zero row)
  int* lidx = reinterpret_cast<int*>(lw + pmr);     // [pm + 2]
  float* zrow = compact ? S + 2 * pmr : Yt + (size_t)P.rows * kp;   // [kp] all-zero row of the Gram panels
  // many-rows variant only: a batch of state rows, the recurrence matrix, column sums of Yl, per-row scalars
  float* Xb = reinterpret_cast<float*>(lidx + ((pm + 3) & ~1));   // [kRowBatch][kp] raw state rows of the batch
  float* Tl = Xb + kRowBatch * kp;                                // [NMAX (padded to 16s)][kRowBatch]
  float* csum = Tl + ((NMAX + 15) & ~15) * kRowBatch;             // [NMAX] sum_i Yl[i][b]

  // XCD-aware block -> point map (letkf_sys.hip): block b's XCD group x = b % 8 owns a contiguous range of
  // point groups; the WPB waves of a block take consecutive points
  int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  int64_t nblk = (P.ng + WPB - 1) / WPB;
  if (bid >= nblk) return;
  int64_t seg0 = 0;
  float* Xab = P.Xa;
  if constexpr (SEG) {     // the same map inside every segment (seg_len % 8 == 0 keeps bid & 7 the XCD)
    const unsigned sg = (unsigned)bid / (unsigned)P.seg_len;
    seg0 = (int64_t)sg * P.seg_len;
    Xab += (int64_t)sg * P.seg_stride;
    bid -= seg0;
    nblk = nblk - seg0 < P.seg_len ? nblk - seg0 : P.seg_len;
  }
  const int64_t q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int64_t grp = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t pt = seg0 + grp * WPB + wave;
  const int64_t ocol = SEG ? grp : P.o0 + pt;          // output column of this point
  if (pt >= P.ng) return;
  const int64_t g = P.g0 + pt;
  if (MIA_XSKIP(P) & 16) return;            // experiment: dispatch floor
  int flag = 0;
  const float km1 = float(k - 1), reg = P.reg;
  const float ar = P.sqrt_reg;
  const float f0 = P.dual ? P.f0_dual : 0.0f;
  float xval[KL];
#pragma unroll
  for (int u = 0; u < KL; ++u) { const int i = tid + 64 * u; xval[u] = (MODE != 2 && i < k) ? P.X[(int64_t)i * P.ldx + g] : 0.0f; }
  int cnt;
  if constexpr (FUSED) {
    // Gaspari-Cohn localisation fused in: scan the cell index, taper in float64, ballot-compact into LDS
    cnt = scan_neighbours<float>(P.scan, g, tid, pm, lidx, lw);
    if (tid == 0) {
      if (cnt > __hip_atomic_load(&P.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&P.stats[0], cnt);
      if (cnt > pm) atomicAdd(&P.stats[1], 1);
    }
  } else {
    // count, list entries and the state row are all requested before anything is waited for: one memory
    // round trip instead of a count -> entries chain (entries beyond the count are loaded and ignored)
    cnt = P.cnt[pt];                           // (first in program order: hipcc issued it after the loop's own wait)
    const int nl = pm < P.p_cap ? pm : P.p_cap;
    for (int j = tid; j < nl; j += 64) {       // (one trip for the dual route: p_max <= 64)
      lidx[j] = P.idx[pt * P.p_cap + j];
      lw[j] = float(P.w[pt * P.p_cap + j]) * (MODE == 2 && P.ienks == 2 ? P.ienks_inv_eps : 1.0f);
    }
  }
  if (cnt > pm || (!FUSED && cnt > P.p_cap) || (P.dual ? cnt : k) > NMAX) {   // loud failure, never truncate
    if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
    const float nanv = __builtin_nanf("");
    for (int it = tid; it < P.m * k; it += 64) {
      if constexpr (SEG) __hip_atomic_store(&Xab[(int64_t)it * P.ldo + ocol], nanv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else Xab[(int64_t)it * P.ldo + ocol] = nanv;
    }
    return;
  }
  if (MIA_XSKIP(P) & 32) return;            // experiment: after the list / state loads
  MIA_WAVE_SYNC();
  {   // gather + sqrt(rho) scale (wrapper.py:91-97).  Four record quads per lane are requested before any is
      // consumed (one memory round trip for the usual <= 256 quads instead of one per 64), and the quad ->
      // (observation, column) split is a multiply-shift (P.kpv_magic = ceil(2^20 / kpv), exact below 2^20 / kpv
      // quads) instead of an integer division per quad
    const unsigned kpv = (unsigned)kp >> 2;
    const int total = ((MIA_XSKIP(P) & 1) || !P.dual) ? 0 : cnt * (int)kpv;   // (the primal route streams the records, see below)
    constexpr int GQ = NMAX > 32 ? 8 : 4;       // quads in flight per lane: large blocks (config 4: 1344 quads) run at
                                                // few waves per CU, so each trip's memory latency is exposed
    // Quad `it` of the block IS quad `it` of Yt (row j, column c <-> j kpv + c), so the store needs no index at all; the
    // load address is one 32 x 32 -> 64 bit multiply-add on the (non-negative) list entry.  it * magic < 2^27 here
    // (it < 64 kpv on the dual route): the split stays in 32-bit arithmetic.
    const float4* rec4 = reinterpret_cast<const float4*>(P.rec);
    float4* Yt4 = reinterpret_cast<float4*>(Yt);
    for (int base = 0; base < total; base += 64 * GQ) {
      float4 v[GQ];
      unsigned jj[GQ];
#pragma unroll
      for (int u = 0; u < GQ; ++u) {
        // (lanes beyond the block re-read quad 0: an unconditional load -- a predicated one made hipcc wait for
        //  vmcnt(0) before each of the GQ requests, i.e. GQ memory round trips instead of one)
        const unsigned it = base + 64 * u + tid < total ? (unsigned)(base + 64 * u + tid) : 0u;
        const unsigned j = (it * (unsigned)P.kpv_magic) >> 20;
        jj[u] = j;
        v[u] = rec4[(uint64_t)(unsigned)lidx[j] * kpv + (it - j * kpv)];
      }
#pragma unroll
      for (int u = 0; u < GQ; ++u) {
        const int it = base + 64 * u + tid;
        if (it < total) {
          const float wj = lw[jj[u]];
          float4 t = v[u];
          t.x *= wj; t.y *= wj; t.z *= wj; t.w *= wj;
          Yt4[it] = t;
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    if (P.ienks == 2) {     // bundle variant: D = Yl / eps, the innovations stay as they are (core/ienks.py:167-173)
      MIA_WAVE_SYNC();
      const float eps = 1.0f / P.ienks_inv_eps;
      for (int j = tid; j < cnt; j += 64) Yt[(size_t)j * kp + k] *= eps;
    }
  }
  if (P.dual) for (int i = tid; i < kp; i += 64) zrow[i] = 0.0f;   // the zero row of the Gram panels
  const int ntrue = P.dual ? cnt : k;
  MIA_WAVE_SYNC();
  // ---- S (full symmetric storage, zero padded)
  f2v srow2[NMAX / 2];
  float rsum = 0.0f;
  if (stream_s) {
    if constexpr (STREAM_OK) {
      // S = Yl^T Yl: the NTILE upper 16x16 tiles are accumulated in ONE pass over the members (every K step loads the
      // TT row panels once and feeds all tiles -- accumulating panel by panel re-read the operands TT times and left
      // the wave waiting on LDS: 0.6 of 1.36 ms per 5e4 analyses of config 4).  Then, panel by panel, the rows
      // 16 p .. 16 p + 15 are assembled in the staging area -- tile (p, c >= p) as it lies in the accumulators
      // (row 4 (lane >> 4) + q, column lane & 15), tile (c < p, p) transposed, which is one 16-byte store per lane --
      // and read back as whole rows by the 16 lanes that own them.
      const int lr = tid & 15, h = tid >> 4;
      const int KS = (k + 3) >> 2;
      const float* prow[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int row = 16 * t + lr;
        prow[t] = (row < cnt ? Yt + (size_t)row * kp : zrow) + KS * h;
      }
#pragma unroll
      for (int b2 = 0; b2 < NMAX / 2; ++b2) srow2[b2] = f2v{0.0f, 0.0f};
      const bool kfull = (k & 3) == 0;
      f32x4c acc[NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) acc[t] = f32x4c{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int s_ = 0; s_ < ((MIA_XSKIP(P) & 2) ? 0 : KS); ++s_) {
        float av_[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) av_[t] = (kfull || KS * h + s_ < k) ? prow[t][s_] : 0.0f;
#pragma unroll
        for (int tb_ = 0, tile = 0; tb_ < TT; ++tb_)
#pragma unroll
          for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
            acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[ta_], av_[tb_], acc[tile], 0, 0, 0);   // rows ta, columns tb
      }
#pragma unroll
      for (int p_ = 0; p_ < TT; ++p_) {
#pragma unroll
        for (int c_ = 0; c_ < TT; ++c_) {
          if (c_ >= p_) {            // tile (p, c): rows of the panel down the q index
            const int tile = c_ * (c_ + 1) / 2 + p_;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * c_ + lr < NMAX) S[(4 * h + q) * LDA + 16 * c_ + lr] = acc[tile][q];
          } else {                   // tile (c, p) transposed: S[16 p + lr][16 c + 4 h + q] = D[4 h + q][lr]
            const int tile = p_ * (p_ + 1) / 2 + c_;
            *reinterpret_cast<f4v*>(S + lr * LDA + 16 * c_ + 4 * h) = f4v{acc[tile][0], acc[tile][1], acc[tile][2], acc[tile][3]};
          }
        }
        MIA_WAVE_SYNC();
        if ((tid >> 4) == p_ && tid < NMAX) {
#pragma unroll
          for (int b4 = 0; b4 < N4; ++b4) {
            const f4v v = reinterpret_cast<const f4v*>(S + lr * LDA)[b4];
            srow2[2 * b4] = v.xy; srow2[2 * b4 + 1] = v.zw;
            rsum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
          }
        }
        MIA_WAVE_SYNC();
      }
    }
  } else if (MIA_XSKIP(P) & 2) {
    for (int it = tid; it < NMAX * LDA; it += 64) S[it] = 0.0f;
  } else if (P.dual) {
    // S = Yl^T Yl on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32).  All TT row panels are loaded
    // once per K step and feed the NTILE upper tiles, whose accumulators are independent (back-to-back
    // MFMA issue).  A[row][kk] = Yt[16*t + (lane&15)][KS*(lane>>4) + s]: any K order sums the same Gram
    // entry.  Panel rows beyond the local observation count read an all-zero row.
    const int lr = tid & 15, h = tid >> 4;
    const int KS = (k + 3) >> 2;
    const float* prow[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int row = 16 * t + lr;
      prow[t] = (row < cnt ? Yt + (size_t)row * kp : zrow) + KS * h;
    }
    f32x4c acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = f32x4c{0.f, 0.f, 0.f, 0.f};
    const bool kfull = (k & 3) == 0;
    if ((k & 7) == 0) {
      // k a multiple of 8: a lane's KS members are an even number of floats at an 8-byte aligned address -- two K steps
      // per 64-bit LDS read, no per-step predicate (the general loop below spends ~10 VALU and two exec-mask
      // round trips per K step on a kernel that is VALU-issue bound)
      const int KS2 = KS >> 1;
#pragma unroll 5
      for (int s2 = 0; s2 < KS2; ++s2) {
        f2v av_[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) av_[t] = reinterpret_cast<const f2v*>(prow[t])[s2];
#pragma unroll
        for (int tb_ = 0, tile = 0; tb_ < TT; ++tb_)
#pragma unroll
          for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile) {
            acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[ta_].x, av_[tb_].x, acc[tile], 0, 0, 0);
            acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[ta_].y, av_[tb_].y, acc[tile], 0, 0, 0);
          }
      }
    } else {
#pragma unroll 5
    for (int s_ = 0; s_ < KS; ++s_) {
      float av_[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) av_[t] = (kfull || KS * h + s_ < k) ? prow[t][s_] : 0.0f;
#pragma unroll
      for (int tb_ = 0, tile = 0; tb_ < TT; ++tb_)
#pragma unroll
        for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
          acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[ta_], av_[tb_], acc[tile], 0, 0, 0);
    }
    }
#pragma unroll
    for (int tb_ = 0, tile = 0; tb_ < TT; ++tb_)
#pragma unroll
      for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = 16 * ta_ + h * 4 + q, b = 16 * tb_ + lr;   // D[row = 4*(lane>>4)+q][col = lane&15]
          if (a < NMAX && b < NMAX) { S[a * LDA + b] = acc[tile][q]; if (ta_ != tb_) S[b * LDA + a] = acc[tile][q]; }
        }
  } else {
    // primal route (p > k, or the RBF-kernelised filter): the k x k matrix of member dot products over the local
    // observations, C[a][b] = sum_j rho_j y_aj y_bj, on the matrix cores -- STREAMED from the packed records: lane (lr, h)
    // feeds A[row 16 t + lr][obs 4 s + h] = rec[idx][16 t + lr] * sqrt(rho), 16 consecutive floats of one record per
    // lane group, so no local block is staged in LDS and the number of local observations is bounded by the list
    // storage only (the LDS block capped it at ~850 for k = 40 and cost the occupancy long before).  The records carry
    // the innovation in column k: the tiles are extended by that one row / column and deliver b = Yl d (and |d_l|^2)
    // for free.  The RBF Gram exp(-gamma |y_a - y_b|^2) (kernels/rbf.py:75-81,110-111) follows as
    // |y_a - y_b|^2 = C_aa + C_bb - 2 C_ab -- the form torch.cdist itself uses above 25 rows -- and the kernel vector
    // k(Yb, d)_a from C_aa + |d|^2 - 2 b_a.
    constexpr int TTE = (NMAX + 16) / 16, NTE = TTE * (TTE + 1) / 2;     // tiles covering rows 0 .. NMAX (>= k + 1 rows)
    const int lr = tid & 15, h = tid >> 4;
    f32x4c acc[NTE];
#pragma unroll
    for (int t = 0; t < NTE; ++t) acc[t] = f32x4c{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (MIA_XSKIP(P) & 2) ? 0 : (cnt + 3) >> 2;
    for (int s_ = 0; s_ < ksteps; ++s_) {
      const int j = 4 * s_ + h;
      float av_[TTE];
      const bool jin = j < cnt;
      const float wj = jin ? lw[j] : 0.0f;
      const float* rj = P.rec + (int64_t)(jin ? lidx[j] : 0) * kp;
#pragma unroll
      for (int t = 0; t < TTE; ++t) av_[t] = (jin && 16 * t + lr <= k) ? rj[16 * t + lr] * wj : 0.0f;
#pragma unroll
      for (int tb_ = 0, tile = 0; tb_ < TTE; ++tb_)
#pragma unroll
        for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
          acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[ta_], av_[tb_], acc[tile], 0, 0, 0);
    }
    // squared norms (the diagonal, incl. |d|^2 at index k) through uq / red, b = column k through rhs
#pragma unroll
    for (int t = 0, tile = 0; t < TTE; tile += t + 2, ++t)      // tile index of (t, t) in the (ta <= tb) enumeration
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (h * 4 + q == lr) {
          const int a = 16 * t + lr;
          if (a < NMAX && a < k) uq[a] = acc[tile][q];
          if (a == k) red[2] = acc[tile][q];
        }
#pragma unroll
    for (int tb_ = 0, tile = 0; tb_ < TTE; ++tb_)
#pragma unroll
      for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = 16 * ta_ + h * 4 + q, b = 16 * tb_ + lr;   // D[row = 4*(lane>>4)+q][col = lane&15], a <= b by tiles
          if (b == k && a < k) rhs[a] = acc[tile][q];              // (a < b: the column lies in the upper tiles)
        }
    MIA_WAVE_SYNC();
#pragma unroll
    for (int tb_ = 0, tile = 0; tb_ < TTE; ++tb_)
#pragma unroll
      for (int ta_ = 0; ta_ <= tb_; ++ta_, ++tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = 16 * ta_ + h * 4 + q, b = 16 * tb_ + lr;
          if (a < NMAX && b < NMAX) {
            float v = (a < k && b < k) ? acc[tile][q] : 0.0f;
            if (P.kernel_mode != 0 && a < k && b < k) v = __expf(-P.gamma * fmaxf(uq[a] + uq[b] - 2.0f * acc[tile][q], 0.0f));
            S[a * LDA + b] = v;
            if (ta_ != tb_) S[b * LDA + a] = v;
          }
        }
    if (tid < NMAX) {
      if (tid >= k) rhs[tid] = 0.0f;
      else if (P.kernel_mode != 0) rhs[tid] = __expf(-P.gamma * fmaxf(uq[tid] + red[2] - 2.0f * rhs[tid], 0.0f));   // k(Yb, d), uncentred
    }
    MIA_WAVE_SYNC();   // uq is reused by the centring below
  }
  MIA_WAVE_SYNC();
  // ---- right-hand side of the mean weights
  if (P.dual) {
    if (tid < NMAX && !compact) rhs[tid] = tid < cnt ? Yt[(size_t)tid * kp + k] : 0.0f;
  } else if (P.kernel_mode == 0) {
    // (rhs = Yl d came with the streamed Gram)
  } else {   // double centring of K and centring of k(Yb, d)   (core/ketkf.py:77-89)
    // Row means and the kernel vector by one lane per member; the two grand means by wave reductions (a single lane
    // summing them serially cost ~5000 cycles); the centring of K itself is folded into the register load of the
    // rows below (K is only ever read there), so the k x k read-modify-write pass over LDS is gone.
    float um = 0.0f, ko_ = 0.0f;
    if (tid < k) {
#pragma unroll
      for (int b4 = 0; b4 < N4; ++b4) {      // columns >= k are zero
        const f4v v = reinterpret_cast<const f4v*>(S + tid * LDA)[b4];
        um += (v.x + v.y) + (v.z + v.w);
      }
      um /= float(k);
      ko_ = rhs[tid];                         // exp(-gamma |y_a - d|^2), from the streamed Gram
    }
    const float gm = wave_sum_dpp(um) / float(k), om = wave_sum_dpp(ko_) / float(k);
    if (tid < NMAX) {
      uq[tid] = tid < k ? um : 0.0f;
      rhs[tid] = tid < k ? ko_ - om - (um - gm) : 0.0f;
    }
    if (tid == 0) red[0] = gm;
  }
  MIA_WAVE_SYNC();
  // ---- row r of S into registers (already there when streamed); Gershgorin bound L >= lambda_max
  if (!stream_s) {
    const int r = tid < NMAX ? tid : NMAX - 1;
#pragma unroll
    for (int b4 = 0; b4 < N4; ++b4) {
      f4v v = reinterpret_cast<const f4v*>(S + r * LDA)[b4];
      if (!P.dual && P.kernel_mode != 0) {   // centred kernel matrix: K_rb - mean_b - (mean_r - grand mean), zero padding kept
        const f4v u4 = reinterpret_cast<const f4v*>(uq)[b4];
        const float cr = uq[r] - red[0];
        v.x = (r < k && 4 * b4 + 0 < k) ? v.x - u4.x - cr : 0.0f;
        v.y = (r < k && 4 * b4 + 1 < k) ? v.y - u4.y - cr : 0.0f;
        v.z = (r < k && 4 * b4 + 2 < k) ? v.z - u4.z - cr : 0.0f;
        v.w = (r < k && 4 * b4 + 3 < k) ? v.w - u4.w - cr : 0.0f;
      }
      srow2[2 * b4] = v.xy; srow2[2 * b4 + 1] = v.zw;
      rsum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
    }
  }
  const float rhs_r = compact ? (tid < cnt ? Yt[(size_t)tid * kp + k] : 0.0f) : (tid < NMAX ? rhs[tid] : 0.0f);
  float L = wave_max_nonneg_dpp(tid < NMAX ? rsum : 0.0f);     // (a NaN row sum survives as the maximum)
  L = fmaxf(L, 1e-30f * reg) * 1.0001f;
  const bool use_tab = P.tab_hdr != nullptr;
  int deg, tab_idx = 0;
  float alpha;                             // A v = alpha S v - v
  if (use_tab) {
    // ---- degree and coefficients of the next tabulated interval [0, Tq reg] that contains [0, L]
    if (!(L == L) || !(fabsf(L) < 1e30f)) { flag |= MIA_FLAG_NONFINITE; L = reg; }
    tab_idx = (int)ceilf(float(kTabPerOctave) * __builtin_amdgcn_logf(L * P.inv_reg)) + kTabIdx0;
    tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);      // (the last entries decline: T = 2^8)
    const int2 h = P.tab_hdr[tab_idx];
    deg = h.x;
    alpha = __int_as_float(h.y) * P.inv_reg;
  } else {
    // ---- degree from the Bernstein-ellipse parameter of the singularity at -reg
    const float sq = __builtin_amdgcn_sqrtf(fmaf(L, P.inv_reg, 1.0f));
    const float rho = (sq + 1.0f) * fast_rcp(fmaxf(sq - 1.0f, 1e-12f));     // in (1, 2e12]: v_log_f32 needs no range fix
    deg = (int)ceilf(P.log_tol * fast_rcp(0.6931471806f * __builtin_amdgcn_logf(rho))) + 2;
    deg = deg < 3 ? 3 : deg;
    if (MIA_XSKIP(P) & 4) deg = 3;
    if (!(L == L) || !(fabsf(L) < 1e30f)) { flag |= MIA_FLAG_NONFINITE; deg = 3; }
    alpha = 2.0f * fast_rcp(L);
  }
  if (deg > P.dmax || deg > DCAP - 1) {   // spectrum too wide for the polynomial route: eigensolver redoes this point
    if (tid == 0) {
      if (P.flags) P.flags[pt] = MIA_FLAG_RETRY;
      atomicAdd(P.retry_count, 1);
    }
    return;
  }
  if constexpr (ALIAS_C) MIA_WAVE_SYNC();     // the rows of S have been read: its storage becomes the coefficient tables
  if (use_tab) {
    const float2 cj = P.tab_c[(size_t)tab_idx * kTabDeg + tid];      // (entries beyond the degree are zero)
    c2[tid] = f2v{cj.x * P.cs_phi, cj.y * P.cs_psi};
    MIA_WAVE_SYNC();
  } else {
    cheb_coefficients(tid, deg, L, reg, P.sqrt_km1, ar, P.dual != 0, f2s, c2);
  }
  if constexpr (MODE == 1) {
    // ---- many state rows: kRowBatch rows at a time as ONE matrix recurrence on the matrix cores.
    //      T (n x 16) lives in the MFMA result layout (lane (lr, h) holds T[16 t + 4 h + q][lr], q = 0..3); every step
    //      is  Y = S T  (A fragments of S stay in registers, T goes through LDS to become the B operand), then the
    //      three-term update and the (phi, psi) accumulation on four values per lane and row tile.  z = Yl^T x' and
    //      the final Yl s are two more small products.  (v_mfma_f32_16x16x4_f32, exact f32; lane (lr, h) feeds
    //      A[16 t + lr][4 s + h] and B[col lr][4 s + h].)  The scalar path below costs ~0.1 ms per row per 1e5
    //      grid points at C2 with 20 of 64 lanes busy.
    //      Primal route (p > k, RBF filter; k <= 48): z = x' itself and (x' W) = phi(S) x', so the batch is the matrix
    //      of centred state rows and the result leaves straight from the registers -- no product with Yl at all.
    static_assert(NMAX <= 48 && KL == 1, "the many-rows variant keeps S in LDS and one member per lane");
    const int lr = tid & 15, h = tid >> 4;
    float afrag[TT][N4];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int s_ = 0; s_ < N4; ++s_) {
        const int r = 16 * t + lr, c = 4 * s_ + h;
        float v = r < NMAX ? S[r * LDA + c] : 0.0f;
        if (!P.dual && P.kernel_mode != 0)      // the centring of K lives in the register rows only: repeat it here
          v = (r < k && c < k) ? v - uq[c] - (uq[r] - red[0]) : 0.0f;
        afrag[t][s_] = v;
      }
    if (P.dual && tid < NMAX) {          // column sums of the local block: centring of z without centring the rows
      float acc = 0.0f;
      if (tid < cnt) for (int i = 0; i < k; ++i) acc += Yt[(size_t)tid * kp + i];
      csum[tid] = acc;
    }
    const int kq = kp >> 2;                 // K steps over the members (pad columns of the rows are zero)
    for (int m0 = 0; m0 < P.m; m0 += kRowBatch) {
      const int nrow = P.m - m0 < kRowBatch ? P.m - m0 : kRowBatch;
      MIA_WAVE_SYNC();
      // -- the batch's state rows, raw, [row c][member i]; means by (row lr, member quarter h)
      for (int c = 0; c < kRowBatch; ++c) {
        float v = 0.0f;
        if (c < nrow && tid < k) v = P.X[((int64_t)(m0 + c) * k + tid) * P.ldx + g];
        for (int i = tid; i < kp; i += 64) Xb[c * kp + i] = i == tid ? v : 0.0f;   // (kp may exceed 64: zero the pad columns)
      }
      MIA_WAVE_SYNC();
      float xm;
      {
        float part = 0.0f;
        for (int i = h; i < k; i += 4) part += Xb[lr * kp + i];
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        xm = part / float(k);               // mean of row lr, in every lane of column lr
      }
      // -- z = Yl^T x' = Yt X^T - csum xm^T
      f32x4c tcur[TT], tprev[TT];
      f2v aphi[TT][4];
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        f32x4c acc = {0.f, 0.f, 0.f, 0.f};
        if (P.dual) {
          const int row = 16 * t + lr;
          const float* pa = Yt + (size_t)(row < cnt ? row : P.rows) * kp;      // rows beyond the list: the zero row
          for (int s_ = 0; s_ < kq; ++s_)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32((4 * s_ + h < k) ? pa[4 * s_ + h] : 0.0f, Xb[lr * kp + 4 * s_ + h], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int b = 16 * t + 4 * h + q;
            acc[q] = (b < cnt) ? acc[q] - csum[b < NMAX ? b : 0] * xm : 0.0f;
          }
        } else {                 // primal: z = x', member b of state row lr
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int b = 16 * t + 4 * h + q;
            acc[q] = (b < k && lr < nrow) ? Xb[lr * kp + b] - xm : 0.0f;
          }
        }
        tprev[t] = acc;
      }
      // -- t1 = A t0, then the recurrence; T is published to LDS before every product
      auto publish = [&](const f32x4c (&tt)[TT]) {
        MIA_WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) Tl[(16 * t + 4 * h + q) * kRowBatch + lr] = tt[t][q];
        MIA_WAVE_SYNC();
      };
      auto product = [&](f32x4c (&yy)[TT]) {
        float bf[N4];
#pragma unroll
        for (int s_ = 0; s_ < N4; ++s_) bf[s_] = Tl[(4 * s_ + h) * kRowBatch + lr];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s_ = 0; s_ < N4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][s_], bf[s_], acc, 0, 0, 0);
          yy[t] = acc;
        }
      };
      f32x4c y[TT];
      publish(tprev);
      product(y);
      {
        const f2v c0 = c2[0], c1 = c2[1];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            tcur[t][q] = alpha * y[t][q] - tprev[t][q];
            aphi[t][q] = c0 * tprev[t][q] + c1 * tcur[t][q];
          }
      }
      for (int j = 2; j <= deg; ++j) {
        publish(tcur);
        product(y);
        const f2v cj = c2[j];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float tn = 2.0f * (alpha * y[t][q] - tcur[t][q]) - tprev[t][q];
            tprev[t][q] = tcur[t][q]; tcur[t][q] = tn;
            aphi[t][q] = cj * tn + aphi[t][q];
          }
      }
      // -- x' w_mean per row (column lr): sum_b rhs_b psi_b, and s = phi(S) z published for the last product
      float zu = 0.0f;
      MIA_WAVE_SYNC();
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int b = 16 * t + 4 * h + q;
          const bool live = b < ntrue;
          zu += live ? rhs[b < NMAX ? b : 0] * aphi[t][q].y : 0.0f;
          Tl[b * kRowBatch + lr] = live ? aphi[t][q].x : 0.0f;
        }
      zu += __shfl_xor(zu, 16, 64);
      zu += __shfl_xor(zu, 32, 64);
      MIA_WAVE_SYNC();
      // -- out[j][c] = xm_c + zu_c + f0 (x_c[j] - xm_c) + sum_b Yl[j][b] s[b][c]
      const float mterm = xm + zu;
      if (!P.dual) {             // primal: (x' W)_j = (phi(S) x')_j, already in this lane's registers
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int jj = 16 * t + 4 * h + q;
            if (jj < k && lr < nrow) {
              const float out = mterm + aphi[t][q].x;
              if (!(fabsf(out) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
              Xab[((int64_t)(m0 + lr) * k + jj) * P.ldo + ocol] = out;
            }
          }
      }
      const int KT = P.dual ? (k + 15) >> 4 : 0;
      for (int tj = 0; tj < KT; ++tj) {
        f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < N4; ++s_) {
          const int b = 4 * s_ + h, jj = 16 * tj + lr;
          const float av = (b < cnt && jj < k) ? Yt[(size_t)b * kp + jj] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Tl[b * kRowBatch + lr], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jj = 16 * tj + 4 * h + q;
          if (jj < k && lr < nrow) {
            const float out = mterm + f0 * (Xb[lr * kp + jj] - xm) + acc[q];
            if (!(fabsf(out) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
            Xab[((int64_t)(m0 + lr) * k + jj) * P.ldo + ocol] = out;
          }
        }
      }
    }
  } else if constexpr (MODE == 2) {
    // ---- weights output without an eigensolver (what LETKF.estimate_weights returns, letkf.py:145-146):
    //      W = w_mean 1^T + f0 I + Yl phi(S) Yl^T,  w_mean = Yl (psi(S) d_l).
    //      psi(S) d_l is the vector recurrence of the transform path applied to d_l; phi(S) is the SAME recurrence
    //      run on the identity, i.e. on n x n matrices T_j = T_j(A), as MFMA products S T_j (A fragments of S in
    //      registers, T_j through LDS into the B operand, tiles in the result layout: lane (lr, h) holds
    //      T[16 ti + 4 h + q][16 tj + lr]).  Two more products, N = Yl Phi and W = N Yl^T, finish the job; the state rows
    //      are then transformed with W.  ~365 MFMAs per grid point at C2 against the Jacobi route's sweeps.
    //      On the primal route (p > k, and the RBF filter) S is k x k already and W = w_mean 1^T + phi(S) needs no further
    //      product: w_mean = psi(S) (Yl d) [or the centred kernel vector], phi(S) = sqrt(k-1) (S + reg)^-1/2.
    static_assert(NMAX <= 48 && KL == 1, "the weights variant keeps S in LDS and one member per lane");
    constexpr int NP = 16 * TT;
    const int lr = tid & 15, h = tid >> 4;
    const int KT = (k + 15) >> 4, KP16 = KT * 16;
    constexpr int NS = NMAX + 4;                                    // row stride of N (columns < NMAX are all that is read)
    float* Tl = reinterpret_cast<float*>(lidx + ((pm + 3) & ~1));   // [NMAX][NP]  (rows < NMAX are all that is read)
    float* Nl = Tl + NMAX * NP;                                     // [KP16][NS]
    float* wbar = Nl + KP16 * NS;                                   // [k]
    float afrag[TT][N4];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int s_ = 0; s_ < N4; ++s_) {
        const int r = 16 * t + lr, c = 4 * s_ + h;
        float v = r < NMAX ? S[r * LDA + c] : 0.0f;
        if (!P.dual && P.kernel_mode != 0)      // the centring of K lives in the register rows only: repeat it here
          v = (r < k && c < k) ? v - uq[c] - (uq[r] - red[0]) : 0.0f;
        afrag[t][s_] = v;
      }
    // -- IEnKS update (tau = 1) through this kernel: with D = Wp^-1 Yl (transform) or Yl / eps (bundle) and reg = k - 1,
    //    Wp' = sqrt(k-1) Pn^-1/2 = I + D phi(S) D^T  and  w_mean' = w_mean - Pn^-1 ((k-1) w_mean - D d) = D psi(S) (d + D^T w_mean)
    //    (core/ienks.py:89-126; Pn = (k-1) I + D D^T, I - (k-1) Pn^-1 = D psi(S) D^T): the LETKF weights with inflation 1
    //    and the start vector of the psi recurrence shifted by D^T w_mean.  Transform variant: only while Wp = I (the
    //    first iteration from the prior weights) -- any other Wp needs its inverse: such points are declined (RETRY).
    float z2 = 0.0f;
    if (P.ienks) {
      const float* win = P.ienks_Win + pt * P.ienks_wstride;
      if (cnt == 0) {        // no local observation: the weights come back as they went in (core/ienks.py:135)
        float* wo = P.W + pt * (int64_t)k * k;
        for (int it = tid; it < k * k; it += 64) wo[it] = win[it];
        if (P.flags && tid == 0) P.flags[pt] = 0;
        return;
      }
      float wmean = 0.0f;
      int not_identity = 0;
      if (tid < k) {
        float acc = 0.0f;
        for (int j = 0; j < k; ++j) acc += win[tid * k + j];
        wmean = (acc - 1.0f) / float(k);
        if (P.ienks == 1)
          for (int j = 0; j < k; ++j) not_identity |= fabsf(win[tid * k + j] - wmean - (j == tid ? 1.0f : 0.0f)) > 1e-6f ? 1 : 0;
      }
      if (__any(not_identity)) {
        if (tid == 0) { if (P.flags) P.flags[pt] = MIA_FLAG_RETRY; atomicAdd(P.retry_count, 1); }
        return;
      }
      MIA_WAVE_SYNC();
      if (tid < kp) xp[tid] = tid < k ? wmean : 0.0f;
      MIA_WAVE_SYNC();
      if (tid < cnt) {
        const float4* yb = reinterpret_cast<const float4*>(Yt + (size_t)tid * kp);
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = 0; i < ((k + 3) >> 2); ++i) { const float4 yq = yb[i], xq = x4[i]; z2 += yq.x * xq.x + yq.y * xq.y + yq.z * xq.z + yq.w * xq.w; }
      }
    }
    // -- u = psi(S) d_l, w_mean = Yl u
    {
      const float t0 = rhs_r + z2;
      if (tid < NMAX) tv[tid] = t0;
      MIA_WAVE_SYNC();
      float yv = matvec_row<NMAX>(srow2, tv);
      float tprev = t0, tcur = alpha * yv - t0;
      float apsi = c2[0].y * t0 + c2[1].y * tcur;
      for (int j = 2; j <= deg; ++j) {
        MIA_WAVE_SYNC();
        if (tid < NMAX) tv[tid] = tcur;
        MIA_WAVE_SYNC();
        yv = matvec_row<NMAX>(srow2, tv);
        const float tnext = 2.0f * (alpha * yv - tcur) - tprev;
        tprev = tcur; tcur = tnext;
        apsi = c2[j].y * tcur + apsi;
      }
      MIA_WAVE_SYNC();
      if (tid < NMAX) sw[tid] = tid < ntrue ? apsi : 0.0f;
      MIA_WAVE_SYNC();
      if (tid < k) {
        float acc = 0.0f;
        if (P.dual) { for (int b = 0; b < cnt; ++b) acc += sw[b] * Yt[(size_t)b * kp + tid]; }
        else acc = sw[tid];                   // primal: w_mean = psi(S) rhs itself
        wbar[tid] = acc;
      }
    }
    // -- Phi = phi(S): matrix recurrence from T_0 = I.  Every T_j is a polynomial in S, hence symmetric: only the
    //    upper tiles (ti <= tj) are computed; publishing a tile also writes its transpose (one 16-byte store per lane)
    constexpr int NUP = TT * (TT + 1) / 2;
    f32x4c tp[NUP], tc[NUP], ph[NUP];
    auto publish = [&](const f32x4c (&tt)[NUP]) {
      MIA_WAVE_SYNC();
#pragma unroll
      for (int tj = 0, tile = 0; tj < TT; ++tj)
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti, ++tile) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (16 * ti + 4 * h + q < NMAX) Tl[(16 * ti + 4 * h + q) * NP + 16 * tj + lr] = tt[tile][q];
          if (ti != tj && 16 * tj + lr < NMAX)
            *reinterpret_cast<f4v*>(Tl + (16 * tj + lr) * NP + 16 * ti + 4 * h) = f4v{tt[tile][0], tt[tile][1], tt[tile][2], tt[tile][3]};
        }
      MIA_WAVE_SYNC();
    };
    auto product = [&](f32x4c (&yy)[NUP]) {       // yy = S T  (T read from Tl), upper tiles
      float bf[TT][N4];
#pragma unroll
      for (int tj = 0; tj < TT; ++tj)
#pragma unroll
        for (int s_ = 0; s_ < N4; ++s_) bf[tj][s_] = Tl[(4 * s_ + h) * NP + 16 * tj + lr];
#pragma unroll
      for (int tj = 0, tile = 0; tj < TT; ++tj)
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti, ++tile) {
          f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s_ = 0; s_ < N4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ti][s_], bf[tj][s_], acc, 0, 0, 0);
          yy[tile] = acc;
        }
    };
#pragma unroll
    for (int tj = 0, tile = 0; tj < TT; ++tj)
#pragma unroll
      for (int ti = 0; ti <= tj; ++ti, ++tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) tp[tile][q] = (16 * ti + 4 * h + q == 16 * tj + lr) ? 1.0f : 0.0f;
    f32x4c y[NUP];
    publish(tp);
    product(y);
    {
      const float c0 = c2[0].x, c1 = c2[1].x;
#pragma unroll
      for (int t = 0; t < NUP; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          tc[t][q] = alpha * y[t][q] - tp[t][q];
          ph[t][q] = c0 * tp[t][q] + c1 * tc[t][q];
        }
    }
    for (int j = 2; j <= deg; ++j) {
      publish(tc);
      product(y);
      const float cj = c2[j].x;
#pragma unroll
      for (int t = 0; t < NUP; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tn = 2.0f * (alpha * y[t][q] - tc[t][q]) - tp[t][q];
          tp[t][q] = tc[t][q]; tc[t][q] = tn;
          ph[t][q] = cj * tn + ph[t][q];
        }
    }
#pragma unroll
    for (int tj = 0, tile = 0; tj < TT; ++tj)        // rows / columns beyond the local observations carry nothing
#pragma unroll
      for (int ti = 0; ti <= tj; ++ti, ++tile)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (16 * ti + 4 * h + q >= ntrue || 16 * tj + lr >= ntrue) ph[tile][q] = 0.0f;
    publish(ph);
    float* wout = P.W + pt * (int64_t)k * k;
    if (!P.dual) {       // primal: W = w_mean 1^T + Phi, written row by row (member j across the lanes)
      if (tid < k)
        for (int i = 0; i < k; ++i) {
          const float v = wbar[i] + Tl[i * NP + tid];
          if (!(fabsf(v) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
          wout[i * k + tid] = v;
        }
    } else {
    // -- N = Yl Phi (k x n) and W = w_mean 1^T + f0 I + N Yl^T.  The fragments of Yl (member 16 t + lr, observation
    //    4 s + h) are the A operand of the first product and the B operand of the second: loaded once.
    constexpr int KTM = 4;                        // k <= 64
    float yfrag[KTM][N4];
#pragma unroll
    for (int t = 0; t < KTM; ++t)
#pragma unroll
      for (int s_ = 0; s_ < N4; ++s_) {
        const int b = 4 * s_ + h, i = 16 * t + lr;
        yfrag[t][s_] = (t < KT && b < cnt && i < k) ? Yt[(size_t)b * kp + i] : 0.0f;
      }
    if (!(MIA_XSKIP(P) & 256)) {
      float bf[TT][N4];
#pragma unroll
      for (int tj = 0; tj < TT; ++tj)
#pragma unroll
        for (int s_ = 0; s_ < N4; ++s_) bf[tj][s_] = Tl[(4 * s_ + h) * NP + 16 * tj + lr];
#pragma unroll
      for (int ti = 0; ti < KTM; ++ti) {
        if (ti < KT) {
#pragma unroll
          for (int tj = 0; tj < TT; ++tj) {
            f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < N4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(yfrag[ti][s_], bf[tj][s_], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * tj + lr < NMAX) Nl[(16 * ti + 4 * h + q) * NS + 16 * tj + lr] = acc[q];
          }
        }
      }
    }
    MIA_WAVE_SYNC();
    if (!(MIA_XSKIP(P) & 512)) {
#pragma unroll
      for (int ti = 0; ti < KTM; ++ti) {
        if (ti < KT) {
          float nfr[N4];
#pragma unroll
          for (int s_ = 0; s_ < N4; ++s_) nfr[s_] = Nl[(16 * ti + lr) * NS + 4 * s_ + h];
#pragma unroll
          for (int tj = 0; tj < KTM; ++tj) {
            if (tj < KT) {
              const int j = 16 * tj + lr;
              f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s_ = 0; s_ < N4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(nfr[s_], yfrag[tj][s_], acc, 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int i = 16 * ti + 4 * h + q;
                if (i < k && j < k) {
                  const float v = wbar[i] + (i == j ? f0 : 0.0f) + acc[q];
                  if (!(fabsf(v) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
                  wout[i * k + j] = v;
                }
              }
            }
          }
        }
      }
    }
    }
    // -- the state rows: xa_j = mean + sum_i x'_i W_ij (base.py:257-278), W read back from where this wavefront just
    //    wrote it (L2-hot, rows contiguous across lanes) -- a copy in LDS would cost 7.7 KB of occupancy at k = 40
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // this wavefront's stores of W are visible to its loads
    __builtin_amdgcn_wave_barrier();
    for (int mi = 0; mi < ((MIA_XSKIP(P) & 1024) ? 0 : P.m); ++mi) {
      const float xv = tid < k ? P.X[((int64_t)mi * k + tid) * P.ldx + g] : 0.0f;
      const float xm = wave_sum_dpp(xv) / float(k);
      if (tid < kp) xp[tid] = tid < k ? xv - xm : 0.0f;
      MIA_WAVE_SYNC();
      if (tid < k) {
        float acc = 0.0f;
        for (int i = 0; i < k; ++i) acc += xp[i] * wout[i * k + tid];
        Xab[((int64_t)mi * k + tid) * P.ldo + ocol] = xm + acc;
      }
      MIA_WAVE_SYNC();
    }
  } else {
  // ---- per state row: z, the shared recurrence, the output
  const int k4 = (k + 3) >> 2;
  for (int mi = 0; mi < P.m; ++mi) {
    if (mi > 0) {
#pragma unroll
      for (int u = 0; u < KL; ++u) { const int i = tid + 64 * u; xval[u] = i < k ? P.X[((int64_t)mi * k + i) * P.ldx + g] : 0.0f; }
    }
    float xs = 0.0f;
#pragma unroll
    for (int u = 0; u < KL; ++u) xs += xval[u];
    const float xm = wave_sum_dpp(xs) * P.inv_k;
#pragma unroll
    for (int u = 0; u <= KL; ++u) {         // one pass more than members per lane: zero the d / pad slots
      const int i = tid + 64 * u;
      if (i < kp) xp[i] = (u < KL && i < k) ? xval[u < KL ? u : 0] - xm : 0.0f;
    }
    MIA_WAVE_SYNC();
    float t0 = 0.0f;                        // z_r
    if (tid < NMAX) {
      if (P.dual) {
        if (tid < cnt) {
          const f4v* yb = reinterpret_cast<const f4v*>(Yt + (size_t)tid * kp);
          const f4v* x4 = reinterpret_cast<const f4v*>(xp);
          f2v a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};          // (pad columns of xp are zero)
#pragma unroll 2
          for (int i = 0; i < k4; ++i) { const f4v y = yb[i], x = x4[i]; a0 = y.xy * x.xy + a0; a1 = y.zw * x.zw + a1; }
          const f2v a = a0 + a1;
          t0 = a.x + a.y;
        }
      } else t0 = tid < k ? xp[tid] : 0.0f;
      tv[tid] = t0;
    }
    MIA_WAVE_SYNC();
    // t1 = A t0
    float y = matvec_row<NMAX>(srow2, tv);
    float tprev = t0, tcur = alpha * y - t0;
    f2v ap = c2[0] * t0 + c2[1] * tcur;       // (phi(S) z, psi(S) z) row r, accumulated together
    for (int j = 2; j <= deg; ++j) {
      MIA_WAVE_SYNC();                      // every lane has read tv
      if (tid < NMAX) tv[tid] = tcur;
      MIA_WAVE_SYNC();
      y = matvec_row<NMAX>(srow2, tv);
      const float tnext = 2.0f * (alpha * y - tcur) - tprev;
      tprev = tcur; tcur = tnext;
      ap = c2[j] * tcur + ap;
    }
    const float aphi = ap.x, apsi = ap.y;
    const bool live = tid < ntrue;
    const float zu = wave_sum_dpp(live ? rhs_r * apsi : 0.0f);      // x' w_mean
    if (tid < NMAX) sw[tid] = live ? aphi : 0.0f;
    MIA_WAVE_SYNC();
    const float mterm = xm + zu;
    float* orow = Xab + (int64_t)mi * k * P.ldo + ocol;
#pragma unroll
    for (int u = 0; u < KL; ++u) {
      const int j = tid + 64 * u;
      if (j < k) {
        float acc;
        if (P.dual) {   // Yl s from the LDS block (still resident on this route; member j contiguous across lanes)
          acc = f0 * xp[j];                        // x' of member j (kept in LDS, not in a register across the recurrence)
#pragma unroll 4
          for (int b = 0; b < ((MIA_XSKIP(P) & 8) ? 0 : cnt); ++b) acc += sw[b] * Yt[(size_t)b * kp + j];
        } else acc = sw[j];
        const float out = mterm + acc;
        if (!(fabsf(out) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
        // segmented launch: write-through (agent-scope) stores, so that the segment counter below can
        // publish them without a cache-wide release (a __threadfence per point costs 18x the kernel)
        if constexpr (SEG) {
          if (MIA_XSKIP(P) & 128) orow[(int64_t)j * P.ldo] = out;    // experiment
          else __hip_atomic_store(&orow[(int64_t)j * P.ldo], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else orow[(int64_t)j * P.ldo] = out;
      }
    }
    MIA_WAVE_SYNC();
  }
  }
  if (P.flags) {
    const int any = __any(flag != 0) ? MIA_FLAG_NONFINITE : 0;
    if (tid == 0) P.flags[pt] = any | (deg << 8);     // bits 8-15: polynomial degree used (diagnostics)
  }
}

template <int NMAX, int KL, bool FUSED, int WPB>
__global__ __launch_bounds__(64 * WPB, (NMAX <= 24 ? (FUSED || KL > 1 ? 5 : 6) : (NMAX <= 40 ? 4 : 2))) void letkf_cheb_kernel(ChebParams P) {
  letkf_cheb_point<NMAX, KL, FUSED, WPB, false>(P);
}

// Many state rows per grid point (m >= 8, dual route, order <= 32): rows are transformed 16 at a time on the matrix cores.
template <int NMAX>
__global__ __launch_bounds__(64, (NMAX <= 32 ? 3 : 2)) void letkf_cheb_rows_kernel(ChebParams P) {
  letkf_cheb_point<NMAX, 1, false, 1, false, 1>(P);
}

// Weights output (dual route, order <= 32): W as a matrix function on the matrix cores, no eigensolver.
template <int NMAX>
__global__ __launch_bounds__(64, (NMAX <= 24 ? 4 : (NMAX <= 32 ? 3 : 2))) void letkf_cheb_weights_kernel(ChebParams P) {
  letkf_cheb_point<NMAX, 1, false, 1, false, 2>(P);
}

// Segmented launch: one grid over the whole block; every workgroup (= one grid point), whatever path it left
// the analysis on, has its output written through to memory and then counts itself in its segment's slot counters.
// A waiter on another stream (segment_wait_kernel) sees a segment complete while later segments still run.
template <int NMAX, int KL>
__global__ __launch_bounds__(64, (NMAX <= 24 ? (KL > 1 ? 5 : 6) : (NMAX <= 40 ? 4 : 2))) void letkf_cheb_seg_kernel(ChebParams P) {
  letkf_cheb_point<NMAX, KL, false, 1, true>(P);
  const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (bid >= P.ng) return;
  // all output stores of this wavefront (write-through, see above) have been acknowledged before it counts
  if (!(MIA_XSKIP(P) & 64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) {
    const unsigned sg = (unsigned)bid / (unsigned)P.seg_len;
    __hip_atomic_fetch_add(P.done + ((size_t)sg * 64 + (unsigned)(bid & 63)) * kSlotStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(64) segment_wait_kernel(const int32_t* done, int expected, int32_t* err, int max_polls) {
  const int lane = threadIdx.x;
  for (int poll = 0; poll < max_polls; ++poll) {
    const int v = __hip_atomic_load(done + (size_t)lane * kSlotStride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (wave_sum(v) >= expected) return;
    __builtin_amdgcn_s_sleep(64);
  }
  if (lane == 0) atomicOr(err, 1);      // exit condition every wave reaches: ~seconds, then report
}

// ------------------------------------------------------------------------------------------------------------
// Large ensembles on the primal route (64 < k <= 128 with more than 64 local observations somewhere in the shard): the
// k x k member Gram is streamed from the records as above, but one lane can no longer keep a whole row of it in
// registers -- S stays in LDS and every lane owns the rows r = lane and r = lane + 64 of the recurrence (float4 reads of
// its rows against the broadcast vector).  Same degree selection, coefficients, flags and retry protocol as the
// kernels above; linear and RBF cores.  Without it these shapes fell to the k x k Jacobi kernel (k = 80, 193 local
// observations: 2.7 ms PER 1e3 analyses).
template <int NMAX>
__global__ __launch_bounds__(64) void letkf_cheb_big_kernel(ChebParams P) {
  constexpr int LDA = NMAX + 4, N4 = NMAX / 4, R = (NMAX + 63) / 64;
  constexpr int TTE = (NMAX + 16) / 16;          // tile rows covering members 0 .. k and the innovation at index k
  constexpr int GC = 3;                          // tile columns accumulated per pass over the local observations
  static_assert(NMAX % 16 == 0 && NMAX > 64 && NMAX <= 128, "orders 80, 96, 112, 128");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int k = P.k, kp = P.kp, pm = P.p_max;
  float* S = reinterpret_cast<float*>(smem_raw);      // [NMAX][LDA]
  float* tv = S + NMAX * LDA;                          // [NMAX]
  float* rhs = tv + NMAX;                              // [NMAX]
  float* uq = rhs + NMAX;                              // [NMAX + 16]  squared norms / row means
  float* sw = uq + NMAX + 16;                          // [NMAX]
  f2v* c2 = reinterpret_cast<f2v*>(sw + NMAX);         // [64]
  f2v* f2s = c2 + 64;                                  // [64]
  float* red = reinterpret_cast<float*>(f2s + 64);     // [8]
  float* lw = red + 8;                                 // [pm + 2]
  int* lidx = reinterpret_cast<int*>(lw + ((pm + 3) & ~1));

  int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (bid >= P.ng) return;
  const int64_t q8 = P.ng >> 3, r8 = P.ng & 7, xcd = bid & 7;
  const int64_t pt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // XCD-aware point map
  const int64_t g = P.g0 + pt, ocol = P.o0 + pt;
  int flag = 0;
  const float km1 = float(k - 1), reg = P.reg, ar = P.sqrt_reg;
  const int nl = pm < P.p_cap ? pm : P.p_cap;
  for (int j = tid; j < nl; j += 64) { lidx[j] = P.idx[pt * P.p_cap + j]; lw[j] = float(P.w[pt * P.p_cap + j]); }
  const int cnt = P.cnt[pt];
  if (cnt > pm || cnt > P.p_cap || k > NMAX) {
    if (P.flags && tid == 0) P.flags[pt] = MIA_FLAG_OVERFLOW;
    const float nanv = __builtin_nanf("");
    for (int it = tid; it < P.m * k; it += 64) P.Xa[(int64_t)it * P.ldo + ocol] = nanv;
    return;
  }
  MIA_WAVE_SYNC();
  // ---- extended member Gram from the records, GC tile columns per pass
  const int lr = tid & 15, h = tid >> 4;
  const int ksteps = (cnt + 3) >> 2;
  for (int it = tid; it < NMAX * LDA; it += 64) S[it] = 0.0f;
  for (int it = tid; it < NMAX; it += 64) rhs[it] = 0.0f;
  MIA_WAVE_SYNC();
#pragma unroll
  for (int c0 = 0; c0 < TTE; c0 += GC) {
    f32x4c acc[GC][TTE];
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
      for (int t = 0; t < TTE; ++t) acc[c][t] = f32x4c{0.f, 0.f, 0.f, 0.f};
    for (int s_ = 0; s_ < ksteps; ++s_) {
      const int j = 4 * s_ + h;
      const bool jin = j < cnt;
      const float wj = jin ? lw[j] : 0.0f;
      const float* rj = P.rec + (int64_t)(jin ? lidx[j] : 0) * kp;
      float av_[TTE];
#pragma unroll
      for (int t = 0; t < TTE; ++t) av_[t] = (jin && 16 * t + lr <= k && t <= c0 + GC - 1) ? rj[16 * t + lr] * wj : 0.0f;
#pragma unroll
      for (int c = 0; c < GC; ++c)
#pragma unroll
        for (int t = 0; t < TTE; ++t)
          if (c0 + c < TTE && t <= c0 + c) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[t], av_[c0 + c], acc[c][t], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
      for (int t = 0; t < TTE; ++t)
        if (c0 + c < TTE && t <= c0 + c) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int a = 16 * t + 4 * h + q, b = 16 * (c0 + c) + lr;       // rows from tile t, columns from tile c0 + c
            const float v = acc[c][t][q];
            if (a < k && b < k) { S[a * LDA + b] = v; S[b * LDA + a] = v; }
            if (a == b && a < k) uq[a] = v;
            if (b == k && a < k) rhs[a] = v;
            if (a == k && b == k) red[2] = v;
          }
        }
  }
  MIA_WAVE_SYNC();
  // ---- RBF core: K_ab = exp(-gamma (C_aa + C_bb - 2 C_ab)), k(Yb, d)_a from C_aa + |d|^2 - 2 b_a, double centring
  if (P.kernel_mode != 0) {
    for (int it = tid; it < k * k; it += 64) {
      const int a = it / k, b = it - a * k;
      S[a * LDA + b] = __expf(-P.gamma * fmaxf(uq[a] + uq[b] - 2.0f * S[a * LDA + b], 0.0f));
    }
    for (int a = tid; a < k; a += 64) rhs[a] = __expf(-P.gamma * fmaxf(uq[a] + red[2] - 2.0f * rhs[a], 0.0f));
    MIA_WAVE_SYNC();
    float um_sum = 0.0f, ko_sum = 0.0f;
    for (int a = tid; a < k; a += 64) {
      float um = 0.0f;
      for (int b4 = 0; b4 < N4; ++b4) { const f4v v = reinterpret_cast<const f4v*>(S + a * LDA)[b4]; um += (v.x + v.y) + (v.z + v.w); }
      um /= float(k);
      sw[a] = um;                                   // row means (sw is free until the state rows)
      um_sum += um; ko_sum += rhs[a];
    }
    const float gm = wave_sum_dpp(um_sum) / float(k), om = wave_sum_dpp(ko_sum) / float(k);
    MIA_WAVE_SYNC();
    for (int it = tid; it < k * k; it += 64) {
      const int a = it / k, b = it - a * k;
      S[a * LDA + b] = S[a * LDA + b] - sw[b] - (sw[a] - gm);
    }
    for (int a = tid; a < k; a += 64) rhs[a] = rhs[a] - om - (sw[a] - gm);
    MIA_WAVE_SYNC();
  }
  // ---- Gershgorin bound, degree, coefficients
  float rsum = 0.0f;
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int r = tid + 64 * u;
    if (r < k) {
      float rs_ = 0.0f;
      for (int b4 = 0; b4 < N4; ++b4) { const f4v v = reinterpret_cast<const f4v*>(S + r * LDA)[b4]; rs_ += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)); }
      rsum = fmaxf(rsum, rs_);
    }
  }
  float L = wave_max_dpp(rsum);
  L = fmaxf(L, 1e-30f * reg) * 1.0001f;
  const float sq = sqrtf(1.0f + L / reg);
  const float rho = (sq + 1.0f) / fmaxf(sq - 1.0f, 1e-12f);
  int deg = (int)ceilf(P.log_tol / __logf(rho)) + 2;
  deg = deg < 3 ? 3 : deg;
  if (!(L == L) || !(fabsf(L) < 1e30f)) { flag |= MIA_FLAG_NONFINITE; deg = 3; }
  if (deg > P.dmax || deg > 63) {
    if (tid == 0) { if (P.flags) P.flags[pt] = MIA_FLAG_RETRY; atomicAdd(P.retry_count, 1); }
    return;
  }
  cheb_coefficients(tid, deg, L, reg, P.sqrt_km1, ar, false, f2s, c2);
  const float alpha = 2.0f / L;
  float rhs_r[R];
#pragma unroll
  for (int u = 0; u < R; ++u) rhs_r[u] = (tid + 64 * u < k) ? rhs[tid + 64 * u] : 0.0f;
  auto matvec = [&](float (&y)[R]) {               // y_r = sum_b S[r][b] tv[b] for this lane's rows
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int r = tid + 64 * u < NMAX ? tid + 64 * u : NMAX - 1;
      f2v a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
      for (int b4 = 0; b4 < N4; ++b4) {
        const f4v sv = reinterpret_cast<const f4v*>(S + r * LDA)[b4];
        const f4v tq = reinterpret_cast<const f4v*>(tv)[b4];
        a0 = sv.xy * tq.xy + a0; a1 = sv.zw * tq.zw + a1;
      }
      const f2v a = a0 + a1;
      y[u] = a.x + a.y;
    }
  };
  // ---- the state rows: z = x' (primal), shared recurrence for phi(S) z and psi(S) z, output
  for (int mi = 0; mi < P.m; ++mi) {
    float xv[R], xs = 0.0f;
#pragma unroll
    for (int u = 0; u < R; ++u) { const int i = tid + 64 * u; xv[u] = i < k ? P.X[((int64_t)mi * k + i) * P.ldx + g] : 0.0f; xs += xv[u]; }
    const float xm = wave_sum_dpp(xs) / float(k);
    float t0[R], tprev[R], tcur[R], y[R];
    f2v ap[R];
    MIA_WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < R; ++u) { const int i = tid + 64 * u; t0[u] = i < k ? xv[u] - xm : 0.0f; if (i < NMAX) tv[i] = t0[u]; }
    MIA_WAVE_SYNC();
    matvec(y);
#pragma unroll
    for (int u = 0; u < R; ++u) { tprev[u] = t0[u]; tcur[u] = alpha * y[u] - t0[u]; ap[u] = c2[0] * t0[u] + c2[1] * tcur[u]; }
    for (int j = 2; j <= deg; ++j) {
      MIA_WAVE_SYNC();
#pragma unroll
      for (int u = 0; u < R; ++u) if (tid + 64 * u < NMAX) tv[tid + 64 * u] = tcur[u];
      MIA_WAVE_SYNC();
      matvec(y);
      const f2v cj = c2[j];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const float tn = 2.0f * (alpha * y[u] - tcur[u]) - tprev[u];
        tprev[u] = tcur[u]; tcur[u] = tn;
        ap[u] = cj * tn + ap[u];
      }
    }
    float zpart = 0.0f;
#pragma unroll
    for (int u = 0; u < R; ++u) if (tid + 64 * u < k) zpart += rhs_r[u] * ap[u].y;
    const float mterm = xm + wave_sum_dpp(zpart);                    // mean + x' w_mean
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int jm = tid + 64 * u;
      if (jm < k) {
        const float out = mterm + ap[u].x;                           // primal: (x' W)_j = (phi(S) x')_j
        if (!(fabsf(out) <= 1e30f)) flag |= MIA_FLAG_NONFINITE;
        P.Xa[((int64_t)mi * k + jm) * P.ldo + ocol] = out;
      }
    }
  }
  if (P.flags) {
    const int any = __any(flag != 0) ? MIA_FLAG_NONFINITE : 0;
    if (tid == 0) P.flags[pt] = any | (deg << 8);
  }
}

template <int NMAX>
static int cheb_launch_big(const ChebParams& ap, hipStream_t stream) {
  const size_t e = (size_t)NMAX * (NMAX + 4) + 4 * (size_t)NMAX + 16 + 4 * 64 + 8 + (size_t)((ap.p_max + 3) & ~1);
  const size_t lds = align_up(e * sizeof(float) + (size_t)((ap.p_max + 3) & ~1) * sizeof(int), 16);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  auto kern = letkf_cheb_big_kernel<NMAX>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t gx = ap.ng < 65536 ? ap.ng : 65536;
  const int64_t gy = (ap.ng + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(ap);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

static size_t cheb_lds_bytes(int kp, int p_max, int nmax, int rows, bool dual, bool batch = false, int k_weights = 0,
                             bool fused = false) {
  const int lda = (nmax % 8 == 0) ? nmax + 4 : nmax;
  if (!batch && !k_weights && cheb_compact_layout(nmax, kp, p_max, dual, fused))
    return align_up(((size_t)nmax * lda + (size_t)nmax * kp) * sizeof(float), 16);
  const int srows = (dual && nmax > 32) ? 16 : nmax;      // streamed S: staging panel only (see letkf_cheb_point)
  const bool alias = !batch && !k_weights && cheb_alias_tables(nmax);   // scalar-rows variant: tables inside S
  size_t e = (size_t)srows * lda + 4 * (size_t)nmax + (alias ? 0 : 4 * 64) + 8 + (size_t)kp + (size_t)(rows + 1) * kp + ((p_max + 3) & ~1);
  if (batch) e += (size_t)kRowBatch * kp + (size_t)((nmax + 15) & ~15) * kRowBatch + (size_t)nmax;
  if (k_weights) {
    const size_t np = (size_t)((nmax + 15) & ~15), kp16 = (size_t)((k_weights + 15) & ~15);
    e += (size_t)nmax * np + kp16 * (size_t)(nmax + 4) + (size_t)k_weights;
  }
  return align_up(e * sizeof(float) + (size_t)((p_max + 3) & ~1) * sizeof(int), 16);
}

template <int NMAX>
static int cheb_launch_weights(const ChebParams& ap, size_t lds, hipStream_t stream) {
  if constexpr (NMAX <= 48) {
    auto kern = letkf_cheb_weights_kernel<NMAX>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t gx = ap.ng < 65536 ? ap.ng : 65536;
    const int64_t gy = (ap.ng + gx - 1) / gx;
    if (gy > 65535) return MIA_ERR_UNSUPPORTED;
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(ap);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  } else {
    return MIA_ERR_UNSUPPORTED;
  }
}

template <int NMAX>
static int cheb_launch_rows(const ChebParams& ap, size_t lds, hipStream_t stream) {
  if constexpr (NMAX <= 48) {
    auto kern = letkf_cheb_rows_kernel<NMAX>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t gx = ap.ng < 65536 ? ap.ng : 65536;
    const int64_t gy = (ap.ng + gx - 1) / gx;
    if (gy > 65535) return MIA_ERR_UNSUPPORTED;
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(ap);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  } else {
    return MIA_ERR_UNSUPPORTED;
  }
}

template <int NMAX, int KL, bool FUSED, int WPB>
static int cheb_launch_w(const ChebParams& ap, size_t lds, hipStream_t stream) {
  auto kern = letkf_cheb_kernel<NMAX, KL, FUSED, WPB>;
  const size_t tot = lds * WPB;
  if (tot > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tot));
  const int64_t nblk = (ap.ng + WPB - 1) / WPB;
  const int64_t gx = nblk < 65536 ? nblk : 65536;
  const int64_t gy = (nblk + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64 * WPB), tot, stream>>>(ap);
  note_analysis_kernel("letkf_cheb_kernel<%d, %d, %s, %d>", NMAX, KL, FUSED ? "true" : "false", WPB);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <int NMAX, int KL, bool FUSED>
static int cheb_launch(const ChebParams& ap, size_t lds, dim3, hipStream_t stream) {
  // four independent points per workgroup when their LDS slices fit comfortably
  if (MIA_EXP_FLAG("MIA_CHEB_WPB4") && lds * 4 <= 40 * 1024) return cheb_launch_w<NMAX, KL, FUSED, 4>(ap, lds, stream);
  return cheb_launch_w<NMAX, KL, FUSED, 1>(ap, lds, stream);
}

template <int NMAX, int KL>
static int cheb_launch_seg(const ChebParams& ap, size_t lds, hipStream_t stream) {
  auto kern = letkf_cheb_seg_kernel<NMAX, KL>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t gx = ap.ng < 65536 ? ap.ng : 65536;
  const int64_t gy = (ap.ng + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(ap);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

int segment_wait_launch(const int32_t* done64, int expected, int32_t* err, hipStream_t stream) {
  segment_wait_kernel<<<1, 64, 0, stream>>>(done64, expected, err, 1 << 21);   // ~1 us per poll
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// MIA_ERR_UNSUPPORTED when the shape is outside this route (caller uses the eigensolver kernels).
// seg_len > 0: segmented launch (see letkf_cheb_seg_kernel); Xa = first segment buffer, ldo = seg_len.
// The dispatch rule below for the plain matfun call (no weights output, no fused localisation, no segments), without
// launching: dual route of at most 64 local observations, coefficient table present, shape and sizes the tile kernel takes.
bool cheb_tile_will_serve(int m, int k, int p_max, int p_cap, float gamma, int64_t ldx, int64_t ldo, int64_t ng, hipStream_t stream) {
  if (gamma > 0.0f || m < 1 || k < 2 || k > 128) return false;
  if (p_max > p_cap) p_max = p_cap;
  if (p_max > k || p_max > 64) return false;
  if (MIA_EXP_FLAG("MIA_EXPERIMENT_SKIP") || MIA_EXP_FLAG("MIA_CHEB_LOGTOL")) return false;
  if (!tile_route_covers(m, k, p_max) || !tile_launch_would_serve(m, k, p_max, p_cap, ldx, ldo, ng, 0)) return false;
  return cheb_coef_table(1, 12.0f, stream) != nullptr;
}

int cheb_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                         float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                         int32_t* flags, int32_t* retry_count, const ScanParams* scan, int32_t* stats,
                         hipStream_t stream, int seg_len, int64_t seg_stride, int32_t* done, float* W_out, const IenksOpts* ienks) {
  if (!flags || !retry_count) return MIA_ERR_UNSUPPORTED;   // the retry protocol needs both
  if (seg_len > 0 && (scan || !done || seg_len % 8 || ng >= (int64_t)1 << 31)) return MIA_ERR_UNSUPPORTED;
  ChebParams ap;
  ap.seg_len = seg_len; ap.seg_stride = seg_stride; ap.done = done;
  ap.W = W_out;
  ap.ienks = ienks ? ienks->variant : 0;
  ap.ienks_Win = ienks ? ienks->W_in : nullptr;
  ap.ienks_wstride = ienks ? ienks->w_stride : 0;
  ap.ienks_inv_eps = ienks ? ienks->inv_eps : 1.0f;
  if (ienks && (!W_out || !ienks->W_in || (ienks->variant != 1 && ienks->variant != 2))) return MIA_ERR_NULL;
  ap.fused = scan != nullptr;
  if (scan) { ap.scan = *scan; ap.stats = stats; if (!stats) return MIA_ERR_NULL; } else ap.stats = nullptr;
  ap.X = X; ap.ldx = ldx; ap.m = m; ap.k = k; ap.g0 = g0; ap.ng = ng; ap.rec = rec;
  ap.kp = (k + 1 + 3) & ~3;
  ap.kpv_magic = ((1 << 20) + (ap.kp >> 2) - 1) / (ap.kp >> 2);
  ap.cnt = nbr_cnt; ap.idx = nbr_idx; ap.w = nbr_w; ap.p_cap = p_cap; ap.p_max = p_max;
  ap.reg = float(k - 1) / inf_factor;
  ap.inv_reg = (float)(1.0 / (double)ap.reg);
  ap.sqrt_reg = (float)sqrt((double)ap.reg);
  ap.f0_dual = (float)sqrt((double)(k - 1) / (double)ap.reg);
  ap.sqrt_km1 = (float)sqrt((double)(k - 1));
  ap.inv_k = (float)(1.0 / (double)k);
  ap.tab_hdr = nullptr; ap.tab_c = nullptr; ap.cs_phi = ap.cs_psi = 1.0f;
  ap.Xa = Xa; ap.ldo = ldo; ap.o0 = o0; ap.flags = flags; ap.retry_count = retry_count;
  ap.kernel_mode = kernel_mode; ap.gamma = gamma;
  ap.dual = (kernel_mode == 0 && p_max <= k) ? 1 : 0;
  const int ntrue = ap.dual ? p_max : k;
  if (ntrue > 64) {     // the eigenproblem is larger than one row per lane: primal route with two rows per lane
    if (k <= 64 || k > 128 || scan || seg_len > 0 || W_out) return MIA_ERR_UNSUPPORTED;
    ap.dual = 0;
    ap.rows = 0;
    ap.dmax = option(MIA_OPT_CHEB_DMAX);
    ap.xskip = 0;
    ap.log_tol = 12.0f;
    MIA_EXP_SET(ap.log_tol, "MIA_CHEB_LOGTOL", (float)atof);
    if (!option(MIA_OPT_CHEB_BIG)) return MIA_ERR_UNSUPPORTED;
    if (k <= 80) return cheb_launch_big<80>(ap, stream);
    if (k <= 96) return cheb_launch_big<96>(ap, stream);
    if (k <= 112) return cheb_launch_big<112>(ap, stream);
    return cheb_launch_big<128>(ap, stream);
  }
  static const int buckets[] = {4, 8, 12, 16, 20, 24, 32, 40, 48, 64};
  int nmax = 0;
  for (int b : buckets) if (b >= ntrue) { nmax = b; break; }
  if (nmax == 0 || k > 128) return MIA_ERR_UNSUPPORTED;
  ap.rows = ap.dual ? nmax : 0;      // the primal route streams the records: no local block in LDS
  // N = deg + 1 coefficient lanes <= 64; 62 covers lambda_max / reg up to ~100 (dense local networks, p >> k)
  ap.dmax = option(MIA_OPT_CHEB_DMAX);
  ap.xskip = 0;
  MIA_EXP_SET(ap.xskip, "MIA_EXPERIMENT_SKIP", atoi);
  // a-priori truncation bound exp(-log_tol) = 6e-6 of the function scale.  Measured against the reference
  // (tools/matfun_tol.py, C2 / C4 / C5): the total error is flat at 1.7e-7 .. 3e-7 (float32 rounding) from 17.5 down to
  // 13, 2.5e-7 / 6.4e-7 / 1.3e-7 at 11, and only at 9 does truncation show (1e-6 / 5e-6); every unit costs ~0.85 of a
  // degree (C2: 15.3 at 15, 11.9 at 11)
  ap.log_tol = 12.0f;
  MIA_EXP_SET(ap.log_tol, "MIA_CHEB_LOGTOL", (float)atof);
  if (!ap.xskip) {
    if (const CoefTable* t = cheb_coef_table(ap.dual, ap.log_tol, stream)) {
      ap.tab_hdr = t->hdr; ap.tab_c = t->c;
      const double rg = (double)ap.reg, km = (double)(k - 1);
      ap.cs_phi = (float)(ap.dual ? sqrt(km) / (rg * sqrt(rg)) : sqrt(km) / sqrt(rg));
      ap.cs_psi = (float)(1.0 / rg);
    }
  }
  // sixteen grid points per wavefront (letkf_tile.hip): dual route, few state rows, no weights output
  // (a launch the tile kernel refuses for its sizes -- offsets beyond 32 bits, LDS, a segmented launch of a large shape -- falls
  //  through to the one-point-per-wavefront kernels below instead of to the eigensolver)
  if (ap.dual && ap.tab_hdr && !W_out && !ienks && !ap.fused && tile_route_covers(m, k, p_max) && option(MIA_OPT_TILE)) {
    const int trc = tile_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, 0, p_cap, p_max, inf_factor, Xa, ldo, o0,
                                         flags, retry_count, ap.dmax, ap.tab_hdr, ap.tab_c, stream, seg_len, seg_stride, done);
    if (trc != MIA_ERR_UNSUPPORTED) return trc;
  }
  const size_t lds = cheb_lds_bytes(ap.kp, p_max, nmax, ap.rows, ap.dual != 0, false, 0, ap.fused != 0);
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  ap.lds_per_wave = (int)lds;
  const int64_t gx = ng < 65536 ? ng : 65536;
  const int64_t gy = (ng + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)gx, (unsigned)gy);
  const bool two = k > 64;
  if (W_out) {   // weights output: dual route up to order 32, primal route (linear / RBF) up to k = 48, one member per lane
    const bool ok_dual = ap.dual && nmax <= 32;
    const bool ok_primal = !ap.dual && nmax <= 48 && !ienks;
    if (!((ok_dual || ok_primal) && !two && !ap.fused && seg_len == 0)) return MIA_ERR_UNSUPPORTED;
    ap.lds_per_wave = (int)cheb_lds_bytes(ap.kp, p_max, nmax, ap.rows, ap.dual != 0, false, k);
    if (ap.lds_per_wave > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
    switch (nmax) {
      case 4: return cheb_launch_weights<4>(ap, ap.lds_per_wave, stream);
      case 8: return cheb_launch_weights<8>(ap, ap.lds_per_wave, stream);
      case 12: return cheb_launch_weights<12>(ap, ap.lds_per_wave, stream);
      case 16: return cheb_launch_weights<16>(ap, ap.lds_per_wave, stream);
      case 20: return cheb_launch_weights<20>(ap, ap.lds_per_wave, stream);
      case 24: return cheb_launch_weights<24>(ap, ap.lds_per_wave, stream);
      case 32: return cheb_launch_weights<32>(ap, ap.lds_per_wave, stream);
      case 40: return cheb_launch_weights<40>(ap, ap.lds_per_wave, stream);
      case 48: return cheb_launch_weights<48>(ap, ap.lds_per_wave, stream);
    }
    return MIA_ERR_UNSUPPORTED;
  }
  // many state rows: batches of 16 rows on the matrix cores (dual route, order <= 32, one member per lane)
  if (m >= 8 && ((ap.dual && nmax <= 32) || (!ap.dual && nmax <= 48)) && !two && !ap.fused && seg_len == 0 &&
      option(MIA_OPT_CHEB_ROWBATCH)) {
    ap.lds_per_wave = (int)cheb_lds_bytes(ap.kp, p_max, nmax, ap.rows, ap.dual != 0, true);
    switch (nmax) {
      case 4: return cheb_launch_rows<4>(ap, ap.lds_per_wave, stream);
      case 8: return cheb_launch_rows<8>(ap, ap.lds_per_wave, stream);
      case 12: return cheb_launch_rows<12>(ap, ap.lds_per_wave, stream);
      case 16: return cheb_launch_rows<16>(ap, ap.lds_per_wave, stream);
      case 20: return cheb_launch_rows<20>(ap, ap.lds_per_wave, stream);
      case 24: return cheb_launch_rows<24>(ap, ap.lds_per_wave, stream);
      case 32: return cheb_launch_rows<32>(ap, ap.lds_per_wave, stream);
      case 40: return cheb_launch_rows<40>(ap, ap.lds_per_wave, stream);
      case 48: return cheb_launch_rows<48>(ap, ap.lds_per_wave, stream);
    }
  }
  if (seg_len > 0) {
#define MIA_CHEB_SEG(N) case N: return two ? cheb_launch_seg<N, 2>(ap, lds, stream) : cheb_launch_seg<N, 1>(ap, lds, stream);
    switch (nmax) {
      MIA_CHEB_SEG(4) MIA_CHEB_SEG(8) MIA_CHEB_SEG(12) MIA_CHEB_SEG(16) MIA_CHEB_SEG(20)
      MIA_CHEB_SEG(24) MIA_CHEB_SEG(32) MIA_CHEB_SEG(40) MIA_CHEB_SEG(48) MIA_CHEB_SEG(64)
    }
#undef MIA_CHEB_SEG
    return MIA_ERR_UNSUPPORTED;
  }
#define MIA_CHEB_CASE(N) case N: return ap.fused ? (two ? cheb_launch<N, 2, true>(ap, lds, grid, stream) : cheb_launch<N, 1, true>(ap, lds, grid, stream)) \
                                        : (two ? cheb_launch<N, 2, false>(ap, lds, grid, stream) : cheb_launch<N, 1, false>(ap, lds, grid, stream));
  switch (nmax) {
    MIA_CHEB_CASE(4) MIA_CHEB_CASE(8) MIA_CHEB_CASE(12) MIA_CHEB_CASE(16) MIA_CHEB_CASE(20)
    MIA_CHEB_CASE(24) MIA_CHEB_CASE(32) MIA_CHEB_CASE(40) MIA_CHEB_CASE(48) MIA_CHEB_CASE(64)
  }
#undef MIA_CHEB_CASE
  return MIA_ERR_UNSUPPORTED;
}

}  // namespace mia

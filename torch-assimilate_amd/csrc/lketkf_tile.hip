// Localised kernel ETKF (RBF / Gauss kernel), sixteen grid points per wavefront FROM TILE LISTS (round 4; list format: mia_tiles.h).
//
// Reference: KETKFModule._estimate_weights (pytassim/core/ketkf.py:65-94) with RBFKernel / GaussKernel
// (pytassim/kernels/rbf.py:75-81,110-111) under wrapper_localization (pytassim/interface/wrapper.py:86-98) and the
// transform of interface/base.py:257-278.  Per grid point g, members a, b and the point's local observations j with
// Gaspari-Cohn weights rho_gj:
//     K_g[a][b] = exp(-gamma sum_j rho_gj (y_aj - y_bj)^2)            k x k, symmetric, ones on the diagonal
//     ko_g[a]   = exp(-gamma sum_j rho_gj (y_aj - d_j)^2)
//     Kc = C K C (double centring, C = I - 1 1^T / k),  koc = ko - mean(ko) - (rowmean(K) - mean(K))
//     xa = mean + x'^T (Kc + reg)^-1 koc + sqrt(k-1) (Kc + reg)^-1/2 x'
// The one-point-per-wavefront kernel (letkf_cheb.hip) builds a Gram matrix per point; here the PAIR STATISTIC is shared
// by the sixteen points of a tile: in the index space of the union of their lists (U slots)
//     Dist[(a, b), g] = sum_s L[(a, b), s] rho[s, g],      L[(a, b), s] = (y_as - y_bs)^2
// is ONE matrix product whose left factor belongs to the tile -- exact f32 on the matrix cores (v_mfma_f32_16x16x4_f32:
// rows = member pairs, columns = the 16 points, depth = slots; the sqrt(rho) matrix of the tile list, squared, is the
// right-hand operand as it stands).  The 16 x 16 result layout hands lane (point = lane & 15, h = lane >> 4) four pairs
// per product, so the pairs are ENUMERATED such that every lane receives exactly the part of ITS point's matrix it will
// multiply with, in registers, and nothing is ever moved:
//
//   * a point's matrix lives in the four lanes h = 0..3 of its column; lane h owns rows a = R h + i, i < R (k' = 4 R >= k,
//     rows / columns >= k are zero), and of every row the CIRCULANT HALF BAND b = a + delta (mod k'), delta = 0 .. H = k'/2
//     -- every unordered pair once (delta = H twice), 820 of 1600 entries at k = 40;
//   * registers hold them as pairs along anti-diagonals, KP[ip][n] = (K[2ip][n+1], K[2ip+1][n]), n < H, so that ONE packed
//     multiply-add serves the forward product (both halves times the same vector entry w[2ip+n+1]) and one the transposed
//     product (both halves into the same partial sum, times (w[2ip], w[2ip+1])); KP[ip][H] = (K[2ip][0], K[2ip+1][H]),
//     KP[ip][H+1] = (ko[2ip], ko[2ip+1]);
//   * a matrix-vector product K u is then R (H + 1) multiply-adds forward + R (H - 1) transposed per lane, the window of
//     u (own rows + the next two lanes') and the transposed partial sums for the next two lanes travel through
//     ds_bpermute_b32 (4 R - 1 per product), sums in a fixed order: a point's result does not depend on its tile mates.
//
// The matrix functions come from the same Chebyshev recurrence as everywhere else (primal table: 1/sqrt(1+t), 1/(1+t)) on
// Kc, applied implicitly (x' is centred; every K u is centred again: 14 instructions instead of 2 per matrix entry);
// the spectral bound is the largest row sum of K (K > 0 entrywise, ||C K C|| <= ||K||).  Degree ~5 at config 5.
// One wavefront per tile, ~300 registers: one wave per SIMD (the matrix of 16 points is the register file).
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"

#ifndef LK_SPLIT8
#define LK_SPLIT8 split8
#endif
#ifndef LK_NV
#define LK_NV 14
#endif
#ifndef LK_NV_PAD
#define LK_NV_PAD 18
#endif
#ifndef LK_MVSB
#define LK_MVSB
#endif
#ifndef LK_ROT
#define LK_ROT 1
#endif
#ifndef LK_SB1
#define LK_SB1
#endif
#ifndef LK_SB2
#define LK_SB2
#endif
namespace mia {

// In-kernel phase stamps (diagnostic builds only, tools/lk_stamps.py; as letkf_tile2.hip): s_memtime at phase boundaries
#ifdef MIA_LK_STAMPS
constexpr int kLkStampN = 12, kLkStampTiles = 8192;
__device__ long long g_lk_stamps[kLkStampTiles * kLkStampN];
#define LK_STAMP(i) do { if (lane == 0 && bid < kLkStampTiles) g_lk_stamps[bid * kLkStampN + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define LK_STAMP_REAL(i) do { if (lane == 0 && bid < kLkStampTiles) g_lk_stamps[bid * kLkStampN + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LK_STAMP(i) do { } while (0)
#define LK_STAMP_REAL(i) do { } while (0)
#endif

struct LkTileParams {
  const float* X; int64_t ldx; int m, k;
  int64_t g0, ng;
  const float* Yb; const float* d; int64_t P;       // [k][P] perturbations, [P] innovations (R^-1/2-normalised)
  const int4* thdr; const int32_t* tidx; const f4w* tD;
  float inv_reg, inv_k, cs_phi, cs_psi, ng2;        // ng2 = -gamma log2(e)
  float* Xa; int64_t ldo, o0; int32_t* flags; int32_t* retry_count;
  int dmax;
  const int2* tab_hdr; const float2* tab_c;
  int seg_len; int64_t seg_stride;                  // pieces, as Tile2Params
  int* clr_counts; const int* clr_n; unsigned* clr_err; int32_t* err_out;   // bucket-index housekeeping, as Tile2Params
};

__device__ __forceinline__ float lk_bperm(int byte_addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}
__device__ __forceinline__ f2w lk_fma2(f2w a, f2w b, f2w c) { return __builtin_elementwise_fma(a, b, c); }
// a value parked in an accumulation register (the matrix of sixteen points is larger than the 256 architectural registers
// of a lane: the tail of every row's band lives in AGPRs and is read into a temporary where it is used)
// (the compiler pads nothing around inline assembly: the value parked is the result of a transcendental instruction, whose
//  consumer needs a wait state -- without the s_nop the write picked up the register's previous content)
// (round 5: the write into the accumulation register is the COMPILER's -- the empty statement only pins the value's register
//  class -- so its hazard recognizer pads it where an in-flight matrix instruction still reads or writes that register: the
//  hand-written v_accvgpr_write of round 4 sat 2 wait states behind a matrix instruction's SrcC read and 7 behind a result
//  write in one instantiation, tools/check_mfma_hazards.py)
__device__ __forceinline__ float lk_park(float v) { float a = v; asm("" : "+a"(a)); return a; }
__device__ __forceinline__ float lk_fetch(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

// PAD: the ensemble is smaller than the 4 R rows the lanes of a point hold (k < 4 R): rows / columns beyond it are zeroed
template <int R, int UT, bool PAD>
__global__ __launch_bounds__(64, 1)
void lketkf_tile_kernel(LkTileParams P) {
  constexpr int KP = 4 * R, H = 2 * R, NP = H + 2, NS = 4 * UT, NSLOT = 16 * UT, S = NSLOT + 4;
  constexpr int ROBS = KP + H, NROW = ROBS + 1;
  constexpr int NV = R >= 10 ? (PAD ? LK_NV_PAD : LK_NV) : H + 1;     // pairs n < NV of a row pair stay in architectural registers
  static_assert(R % 2 == 0 && R >= 2 && R <= 10, "row pairs per lane");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* T = reinterpret_cast<float*>(smem);                    // [NROW][S]: member rows (wrapped copy of the first H), obs row
  int* ukey = reinterpret_cast<int*>(smem + (size_t)NROW * S * 4);   // [NSLOT] observation index of a slot
  unsigned* smax = reinterpret_cast<unsigned*>(ukey + NSLOT);        // [NSLOT] largest magnitude of a slot's column (bit pattern)
  float* esc2 = reinterpret_cast<float*>(smax + NSLOT);              // [NSLOT] 2^-2e of a slot, in operand order
  const int lane = threadIdx.x, lr = lane & 15, h = lane >> 4;
  const int k = P.k;

  const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int64_t ntile = (P.ng + 15) >> 4;
  if (bid >= ntile) return;
  if (P.clr_counts) {
    const int ncl = *P.clr_n;
    for (int64_t i = bid * 64 + lane; i < ncl; i += ntile * 64) P.clr_counts[i] = 0;
    if (bid == 0 && lane == 0) {
      const unsigned e = *P.clr_err;
      if (e) { atomicOr(P.err_out, (int)(e << 3)); *P.clr_err = 0u; }
    }
  }
  const int64_t q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
  const int64_t tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t p0 = tile << 4;
  const int npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
  int64_t oc0 = P.o0 + p0;
  if (P.seg_len > 0) {
    const unsigned sgi = (unsigned)p0 / (unsigned)P.seg_len;
    oc0 = p0 - (int64_t)sgi * P.seg_len;
    P.Xa += (int64_t)sgi * P.seg_stride;
  }
  const unsigned ldxb = (unsigned)P.ldx * 4u, ldob = (unsigned)P.ldo * 4u;
  const int lrc = lr < npts ? lr : npts - 1;
  const bool colok = lr < npts;
  constexpr bool pad = PAD;

  LK_STAMP(0);
  LK_STAMP_REAL(10);
  // ---- first round trip: header, slot table, sqrt(rho) matrix
  const int4 hd = P.thdr[tile];
  constexpr bool FIXSLOT = (64 % NSLOT) == 0;      // a lane's elements lane + 64 it all belong to slot lane % NSLOT
  int myj = -1;
  if constexpr (FIXSLOT) myj = t2_ld<int32_t>(P.tidx + tile * NSLOT, (unsigned)(lane & (NSLOT - 1)) * 4u);
  else
    for (int s = lane; s < NSLOT; s += 64) ukey[s] = t2_ld<int32_t>(P.tidx + tile * NSLOT, (unsigned)s * 4u);
  f4w dreg[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) dreg[t] = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
  MIA_T2_SYNC();
  const int U = __builtin_amdgcn_readfirstlane(hd.x);
  if (U < 0) {                     // the union of this tile did not fit its slots: loud failure, never a truncated analysis
    if (colok && h == 0) P.flags[p0 + lr] = MIA_FLAG_OVERFLOW;
    const float nanv = __builtin_nanf("");
    if (colok)
      for (int it = h; it < P.m * k; it += 4) P.Xa[(int64_t)it * P.ldo + oc0 + lr] = nanv;
    return;
  }
  LK_STAMP(1);      // header, slot table
  // ---- second round trip: the union's columns of Yb (and d) as an f32 image T[member][slot]; slot 16 t + 4 kk + q sits at
  //      position kk NS + 4 t + q of its row: the lane group kk of the A operand reads its NS values as UT 16-byte pieces.
  //      Every slot's column is scaled by its own power of two 2^e (largest magnitude -> [2^5, 2^6): squared differences stay
  //      below 2^14, inside half precision); the scale returns through the weights, rho_hat = rho 2^-2e.
  {
    constexpr int NE = (KP + 1) * NSLOT, NLD = (NE + 63) / 64;
    if constexpr (!FIXSLOT)
      for (int s_ = lane; s_ < NSLOT; s_ += 64) smax[s_] = 0u;
    float v[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int e = lane + 64 * it;
      const int a = e / NSLOT, sl = e - a * NSLOT;
      int j;
      if constexpr (FIXSLOT) j = e < NE ? myj : -1; else j = e < NE ? ukey[sl] : -1;
      const bool isobs = a == KP;
      const bool ld = j >= 0 && (a < k || isobs);
      const unsigned off = ((unsigned)(isobs ? 0 : a) * (unsigned)P.P + (unsigned)(j < 0 ? 0 : j)) * 4u;
      const float* src = isobs ? P.d : P.Yb;
      v[it] = ld ? t2_ld<float>(src, off) : 0.0f;
    }
    unsigned mymx = 0u;
    if constexpr (FIXSLOT) {       // the column's largest magnitude: in registers, over the 64 / NSLOT lanes that share the slot
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const unsigned a_ = __float_as_uint(v[it]) & 0x7fffffffu;
        mymx = (lane + 64 * it < NE && a_ > mymx) ? a_ : mymx;
      }
      typedef unsigned u2v __attribute__((ext_vector_type(2)));
      if constexpr (NSLOT <= 32) {
        const u2v r_ = __builtin_amdgcn_permlane32_swap(mymx, mymx, false, false);
        mymx = r_.x > r_.y ? r_.x : r_.y;
      }
      if constexpr (NSLOT <= 16) {
        const u2v r_ = __builtin_amdgcn_permlane16_swap(mymx, mymx, false, false);
        mymx = r_.x > r_.y ? r_.x : r_.y;
      }
    } else {
      MIA_T2_SYNC();
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int e = lane + 64 * it;
        const int a = e / NSLOT, sl = e - a * NSLOT;
        if (e < NE) atomicMax(&smax[sl], __float_as_uint(v[it]) & 0x7fffffffu);
      }
      MIA_T2_SYNC();
    }
    bool badrec = false;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int e = lane + 64 * it;
      const int a = e / NSLOT, sl = e - a * NSLOT;
      const int pos = ((sl >> 2) & 3) * NS + 4 * (sl >> 4) + (sl & 3);
      if (e < NE) {
        const unsigned mx = FIXSLOT ? mymx : smax[sl];
        badrec = badrec || mx >= 0x7f800000u;
        int es;
        const float sc = pow2_scale(mx, 5, &es);
        const float x = v[it] * sc;
        const int row = a == KP ? ROBS : a;
        T[row * S + pos] = x;
        if (a < H) T[(KP + a) * S + pos] = x;
        if (a == 0) esc2[pos] = __uint_as_float((unsigned)(127 - 2 * es) << 23);      // 2^-2e, in the operand's slot order
      }
    }
    // a non-finite record: through the shared product it would reach all 16 columns (NaN * 0), also the points that do not see
    // that observation -- every point of such a tile goes to the eigensolver kernel, which works point by point
    if (__any(badrec)) {
      if (colok && h == 0) { P.flags[p0 + lr] = MIA_FLAG_RETRY; atomicAdd(P.retry_count, 1); }
      return;
    }
  }
  MIA_T2_SYNC();
  LK_STAMP(2);      // image of the union's records in LDS

  // ---- pair statistic on the matrix cores: Dist[(a, b), g] = sum_s (yhat_as - yhat_bs)^2 rho_hat_gs as split half-precision
  //      products (v_mfma_f32_16x16x32_f16: hi = f16(x), lo = f16(x - hi); hi hi + hi lo + lo hi, 22 bits, f32 accumulation --
  //      all terms are non-negative, nothing cancels).  The f32 matrix instruction shares the vector unit's issue (measured:
  //      33 cycles each and NOT overlapped with the vector instructions that build its operand); the half-precision one
  //      runs beside them.  A operand: lane (r = lane & 15, kg = lane >> 4) supplies row r = 4 h' + q' of a row block -- pair
  //      (ip, n = 2 nn + (q' >> 1), half = q' & 1), row a = R h' + 2 ip + half, partner b = a + delta with delta = n + 1 - half
  //      (n < H), (0, H) for n = H, the observation row for n = H + 1 -- over the slots 16 t + 4 kg + q, k index 8 kg + 4 t + q
  //      of product t >> 1; B operand: this lane's own entries of the sqrt(rho) matrix, squared, times 2^-2e, scaled per point.
  f2w KPr[R / 2][NP];
  float KA[R / 2][NP][2];       // the band's tail, parked in accumulation registers
  {
    constexpr int NT2 = (UT + 1) / 2;
    h8v bh[NT2], bl[NT2];
    float exparg;
    {
      float bsq[2 * NT2][4];
      unsigned bmx = 0u;
#pragma unroll
      for (int t = 0; t < 2 * NT2; ++t) {
        const f4w e4 = t < UT ? *reinterpret_cast<const f4w*>(esc2 + h * NS + 4 * t) : f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bsq[t][q] = t < UT ? dreg[t < UT ? t : 0][q] * dreg[t < UT ? t : 0][q] * e4[q] : 0.0f;
          const unsigned u_ = __float_as_uint(bsq[t][q]) & 0x7fffffffu;
          bmx = u_ > bmx ? u_ : bmx;
        }
      }
      bmx = t2_max_h(bmx);
      int esb;
      const float sb = pow2_scale(bmx, 9, &esb);
      exparg = P.ng2 * __uint_as_float((unsigned)(127 - esb) << 23);
#pragma unroll
      for (int g = 0; g < NT2; ++g) {
        float b8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b8[i] = bsq[2 * g + (i >> 2)][i & 3] * sb;
        split8_tied(b8, bh[g], bl[g]);
      }
    }
    const int hp = lr >> 2, qp = lr & 3;
    const unsigned aoff = (unsigned)((R * hp + (qp & 1)) * S + h * NS) * 4u;
    const unsigned boff = aoff + (unsigned)(((qp >> 1) + 1 - (qp & 1)) * S) * 4u;
    const unsigned char* Tb = reinterpret_cast<const unsigned char*>(T);
    // the special block of a row pair (n = H, H + 1): q' = 0: delta 0; 1: delta H; 2, 3: observation row
    const unsigned ospec0 = qp >= 2 ? (unsigned)(ROBS * S + h * NS) * 4u : aoff + (unsigned)((qp == 1 ? H : 0) * S) * 4u;
    const unsigned ospec_step = qp >= 2 ? 0u : (unsigned)(2 * S) * 4u;
    constexpr int NBP = NP / 2, NB = (R / 2) * NBP;
    // software pipeline, one row block per stage (a scheduling barrier closes every stage): the operands of block b + 1 are
    // read from LDS, block b's differences are squared, split and multiplied, and block b - 1 goes through exp2, masking and
    // into its home registers
    auto ld_block = [&](int bb, f4w (&ta)[UT], f4w (&tb)[UT]) {
      const int ip = bb / NBP, nn = bb % NBP;
      const unsigned oa = aoff + (unsigned)(2 * ip * S) * 4u;
      const unsigned ob = nn < H / 2 ? boff + (unsigned)((2 * ip + 2 * nn) * S) * 4u : ospec0 + (unsigned)ip * ospec_step;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        ta[t] = *reinterpret_cast<const f4w*>(Tb + oa + 16u * t);
        tb[t] = *reinterpret_cast<const f4w*>(Tb + ob + 16u * t);
      }
    };
    auto fin_block = [&](int bb, const f4w acc) {
      const int ip = bb / NBP, nn = bb % NBP;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 2 * nn + (q >> 1), hf = q & 1;
        float kv = __builtin_amdgcn_exp2f(acc[q] * exparg);
        if (pad) {          // rows / columns beyond the ensemble (k < 4 R): zero
          const int dlt = n < H ? n + 1 - hf : (n == H ? (hf ? H : 0) : 0);
          const int a = R * h + 2 * ip + hf;
          int b = a + dlt;
          b = b >= KP ? b - KP : b;
          kv = (a < k && (n == H + 1 || b < k)) ? kv : 0.0f;
        }
        if (n >= NV && n <= H) KA[ip][n][hf] = lk_park(kv); else KPr[ip][n][hf] = kv;
      }
    };
    f4w ta[2][UT], tb[2][UT], acc[2];
    ld_block(0, ta[0], tb[0]);
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      if (bb + 1 < NB) ld_block(bb + 1, ta[(bb + 1) & 1], tb[(bb + 1) & 1]);
      f4w ac = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NT2; ++g) {
        float l8[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * g + tt;
          // (vector form: the compiler packs the four differences and squares of a 16-byte piece in pairs)
          const f4w dl = t < UT ? ta[bb & 1][t < UT ? t : 0] - tb[bb & 1][t < UT ? t : 0] : f4w{0.f, 0.f, 0.f, 0.f};
          const f4w sq = dl * dl;
#pragma unroll
          for (int q = 0; q < 4; ++q) l8[4 * tt + q] = sq[q];
        }
        h8v ah, al;
        LK_SPLIT8(l8, ah, al);
        ac = t2_mfma3(ac, ah, al, bh[g], bl[g]);
      }
      acc[bb & 1] = ac;
      if (bb >= 1) fin_block(bb - 1, acc[(bb - 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    fin_block(NB - 1, acc[(NB - 1) & 1]);
  }
  auto getK = [&](int ip, int n) -> f2w {
    if (n >= NV && n <= H) return f2w{lk_fetch(KA[ip][n][0]), lk_fetch(KA[ip][n][1])};
    return KPr[ip][n];
  };
  LK_STAMP(3);      // pair statistic, exp
  const int adr_p1 = ((lane + 16) & 63) * 4, adr_p2 = ((lane + 32) & 63) * 4, adr_m1 = ((lane + 48) & 63) * 4;
  // y = K u for the rows of this lane.  Anti-diagonal by anti-diagonal (j = row + delta, a scheduling barrier after each):
  // the forward products of all row pairs with w[j], the transposed products into the partial sum of target row j, which
  // leaves for its owner (this lane: j < R; the next lane: j < 2 R; the one after) as soon as it is complete and is added
  // two anti-diagonals later -- the live set is the matrix, the window (2 R), R accumulators and a few values in flight
  // (vectors are carried as PAIRS of rows, (u[2 ip], u[2 ip + 1]): the multiplier of the transposed products and the
  //  accumulators of the forward ones are such pairs as they stand -- no register moves to build operands)
  auto matvec = [&](const f2w (&u)[R / 2], f2w (&yo)[R / 2]) {
    f2w n1[R / 2], n2[R / 2];
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) n1[ip] = f2w{lk_bperm(adr_p1, u[ip][0]), lk_bperm(adr_p1, u[ip][1])};
    auto wv = [&](int j) -> float { return j < R ? u[j >> 1][j & 1] : (j < 2 * R ? n1[(j - R) >> 1][(j - R) & 1] : n2[(j - 2 * R) >> 1][(j - 2 * R) & 1]); };
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) yo[ip] = f2w{0.f, 0.f};
    float pend[3 * R];
    LK_MVSB;
#pragma unroll
    for (int j = 1; j <= 3 * R - 2; ++j) {
      if (j == R) {
#pragma unroll
        for (int ip = 0; ip < R / 2; ++ip) n2[ip] = f2w{lk_bperm(adr_p2, u[ip][0]), lk_bperm(adr_p2, u[ip][1])};
      }
      f2w a2 = f2w{0.f, 0.f}, b2 = f2w{0.f, 0.f};
      const float wj = wv(j);
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) {
        const int n = j - 1 - 2 * ip;
        if (n >= 0 && n < H) {
          const f2w kp = getK(ip, n);
          yo[ip] = lk_fma2(kp, f2w{wj, wj}, yo[ip]);
          const f2w mp = (n >= 1 && n <= H - 2) ? u[ip] : f2w{n <= H - 2 ? u[ip][0] : 0.0f, n >= 1 ? u[ip][1] : 0.0f};
          if (ip & 1) b2 = lk_fma2(kp, mp, b2); else a2 = lk_fma2(kp, mp, a2);
        }
      }
      const f2w ab = a2 + b2;
      const float ywj = ab[0] + ab[1];
      if (j < R) pend[j] = ywj;
      else if (j < 2 * R) pend[j] = lk_bperm(adr_m1, ywj);
      else pend[j] = lk_bperm(adr_p2, ywj);
      if (j >= 3) {          // the partial sum of two anti-diagonals ago has arrived: into its row's accumulator
        const int jj = j - 2, i = jj < R ? jj : (jj < 2 * R ? jj - R : jj - 2 * R);
        yo[i >> 1][i & 1] += pend[jj];
      }
      LK_MVSB;
    }
#pragma unroll
    for (int jj = 3 * R - 3; jj <= 3 * R - 2; ++jj) {
      if (jj >= 1) {
        const int i = jj < R ? jj : (jj < 2 * R ? jj - R : jj - 2 * R);
        yo[i >> 1][i & 1] += pend[jj];
      }
    }
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip)         // the pairs that have no partner: (K[2ip][0], K[2ip+1][H])
      yo[ip] = lk_fma2(getK(ip, H), f2w{u[ip][0], wv(2 * ip + 1 + H)}, yo[ip]);
    LK_MVSB;
  };
  auto rowok = [&](int i) -> bool { return !pad || R * h + i < k; };

  // ---- row sums r = K 1: spectral bound (largest row sum), the centring terms of the kernel vector
  float alpha = 0.0f;
  int deg = 0, tab_idx = 0, degmax = 0, pflag = 0;
  bool decl = false;
  {
    f2w ones[R / 2], r[R / 2];
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) ones[ip] = f2w{rowok(2 * ip) ? 1.0f : 0.0f, rowok(2 * ip + 1) ? 1.0f : 0.0f};
    matvec(ones, r);
    float L = 0.0f;
    f2w rs2 = f2w{0.f, 0.f}, ko2 = f2w{0.f, 0.f};
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) {
#pragma unroll
      for (int c = 0; c < 2; ++c) L = (r[ip][c] > L || r[ip][c] != r[ip][c]) ? r[ip][c] : L;
      rs2 += r[ip];
      ko2 += KPr[ip][H + 1];
    }
    L = __uint_as_float(t2_max_h(__float_as_uint(L)));          // (non-negative or NaN: bit patterns order like the values)
    const float rs = t2_add_h(rs2[0] + rs2[1]) * P.inv_k * P.inv_k;      // grand mean of K
    const float kos = t2_add_h(ko2[0] + ko2[1]) * P.inv_k;               // mean of ko
    // koc = ko - mean(ko) - (rowmean - grand mean), ketkf.py:77-88
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) {
      f2w kc = KPr[ip][H + 1] - f2w{kos, kos} - (r[ip] * f2w{P.inv_k, P.inv_k} - f2w{rs, rs});
      if (pad) kc = f2w{rowok(2 * ip) ? kc[0] : 0.0f, rowok(2 * ip + 1) ? kc[1] : 0.0f};
      KPr[ip][H + 1] = kc;
    }
    L = fmaxf(L, 1e-30f) * 1.00001f;
    if (!(L == L) || !(fabsf(L) < 1e30f)) { pflag |= MIA_FLAG_NONFINITE; L = 1.0f; }
    tab_idx = (int)ceilf(float(kTabPerOctave) * __builtin_amdgcn_logf(L * P.inv_reg)) + kTabIdx0;
    tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
    const int2 th = t2_ld<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
    deg = th.x;
    alpha = __int_as_float(th.y) * P.inv_reg;
    decl = colok && (deg > P.dmax || deg > kTabDeg - 1);
    if (decl && h == 0) {
      P.flags[p0 + lr] = MIA_FLAG_RETRY;
      atomicAdd(P.retry_count, 1);
    }
    degmax = (int)wave_max_nonneg_dpp((colok && !decl) ? float(deg) : 0.0f);
  }

  LK_STAMP(4);      // row sums, bound, degree
  // ---- per state row: recurrence u_{j+1} = 2 (alpha Kc u_j - u_j) - u_{j-1} on x', accumulating phi(Kc) x' and koc . psi(Kc) x'
  const unsigned cbase = (unsigned)tab_idx * (unsigned)(kTabDeg * 8);
  auto coef = [&](int j) -> float2 { return t2_ld<float2>(P.tab_c, cbase + (unsigned)(j < kTabDeg ? j : kTabDeg - 1) * 8u); };
  for (int mi = 0; mi < P.m; ++mi) {
    const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0;
    f2w va[R / 2], vb[R / 2];
    f2w xs2 = f2w{0.f, 0.f};
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int a = R * h + 2 * ip + c;
        const bool live = !pad || a < k;
        const float x = t2_ld<float>(xbase, (unsigned)(live ? a : 0) * ldxb + (unsigned)lrc * 4u);
        va[ip][c] = live ? x : 0.0f;
      }
      xs2 += va[ip];
    }
    const float2 c0 = coef(0), c1 = coef(1);
    float2 cn0 = coef(2), cn1 = coef(3);
    const float xm = t2_add_h(xs2[0] + xs2[1]) * P.inv_k;
#pragma unroll
    for (int ip = 0; ip < R / 2; ++ip) {
      va[ip] -= f2w{xm, xm};
      if (pad) va[ip] = f2w{rowok(2 * ip) ? va[ip][0] : 0.0f, rowok(2 * ip + 1) ? va[ip][1] : 0.0f};
    }
    f2w aphi[R / 2], y[R / 2];
    float zacc = 0.0f;
    auto centre = [&](const f2w (&yy)[R / 2]) -> float {       // alpha * mean of K u over the members
      f2w s2 = f2w{0.f, 0.f};
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) s2 += yy[ip];
      return t2_add_h(s2[0] + s2[1]) * P.inv_k * alpha;
    };
    auto kocdot = [&](const f2w (&vv)[R / 2]) -> float {
      f2w s2 = f2w{0.f, 0.f};
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) s2 = lk_fma2(KPr[ip][H + 1], vv[ip], s2);
      return s2[0] + s2[1];
    };
    const f2w al2 = f2w{alpha, alpha};
    matvec(va, y);
    {
      const float am = centre(y);
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) {
        f2w vq = lk_fma2(al2, y[ip], -(va[ip] + f2w{am, am}));
        if (pad) vq = f2w{rowok(2 * ip) ? vq[0] : 0.0f, rowok(2 * ip + 1) ? vq[1] : 0.0f};
        vb[ip] = vq;
        aphi[ip] = lk_fma2(f2w{c1.x, c1.x}, vq, f2w{c0.x, c0.x} * va[ip]);
      }
      zacc = fmaf(c1.y, kocdot(vb), c0.y * kocdot(va));
    }
    auto advance = [&](f2w (&vold)[R / 2], const f2w (&vcur)[R / 2], const float2 cj) {
      matvec(vcur, y);
      const float am = centre(y);
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) {
        const f2w tq = lk_fma2(al2, y[ip], -(vcur[ip] + f2w{am, am}));
        f2w vn = lk_fma2(f2w{2.0f, 2.0f}, tq, -vold[ip]);
        if (pad) vn = f2w{rowok(2 * ip) ? vn[0] : 0.0f, rowok(2 * ip + 1) ? vn[1] : 0.0f};
        vold[ip] = vn;
        aphi[ip] = lk_fma2(f2w{cj.x, cj.x}, vn, aphi[ip]);
      }
      zacc = fmaf(cj.y, kocdot(vold), zacc);
    };
#if LK_ROT
    for (int j = 2; j <= degmax; ++j) {
      const float2 cj = cn0;
      cn0 = cn1; cn1 = coef(j + 2);
      advance(va, vb, cj);
#pragma unroll
      for (int ip = 0; ip < R / 2; ++ip) { const f2w t = va[ip]; va[ip] = vb[ip]; vb[ip] = t; }
    }
#else
    int j = 2;
    for (; j + 1 <= degmax; j += 2) {
      const float2 cj = cn0, cj1 = cn1;
      cn0 = coef(j + 2); cn1 = coef(j + 3);
      advance(va, vb, cj);
      advance(vb, va, cj1);
    }
    if (j <= degmax) advance(va, vb, cn0);
#endif
    if (mi == 0) LK_STAMP(5);      // recurrence of the first state row
    const float zu = t2_add_h(zacc) * P.cs_psi;
    const float mterm = xm + zu;
    int pf = 0;
    if (colok && !decl) {
      float* obase = P.Xa + (int64_t)mi * k * P.ldo + oc0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int a = R * h + i;
        const float o = fmaf(P.cs_phi, aphi[i >> 1][i & 1], mterm);
        if (!pad || a < k) {
          if (!(fabsf(o) <= 1e30f)) pf = MIA_FLAG_NONFINITE;
          *reinterpret_cast<float*>(reinterpret_cast<char*>(obase) + ((unsigned)a * ldob + (unsigned)lr * 4u)) = o;
        }
      }
    }
    pflag |= pf;
  }
  {
    const unsigned long long fb = __ballot(pflag != 0);
    const bool anyf = ((fb >> lr) & 0x0001000100010001ull) != 0ull;
    if (h == 0 && colok && !decl) P.flags[p0 + lr] = (anyf ? MIA_FLAG_NONFINITE : 0) | (deg << 8);
  }
  LK_STAMP(6);
  LK_STAMP_REAL(11);
}

#ifdef MIA_LK_STAMPS
extern "C" int mia_debug_lk_stamps(long long* host, int n_tiles) {
  if (n_tiles > kLkStampTiles) n_tiles = kLkStampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lk_stamps), sizeof(long long) * kLkStampN * (size_t)n_tiles);
}
#endif

static size_t lk_lds_bytes(int r, int ut) { return (size_t)(6 * r + 1) * (16 * ut + 4) * 4 + (size_t)16 * ut * 12; }

template <int R, int UT, bool PAD>
static int lk_launch_p(const LkTileParams& tp, hipStream_t stream) {
  const size_t lds = lk_lds_bytes(R, UT);
  auto kern = lketkf_tile_kernel<R, UT, PAD>;
  const int64_t ntile = (tp.ng + 15) >> 4;
  const int64_t gx = ntile < 65536 ? ntile : 65536;
  const int64_t gy = (ntile + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  hipEvent_t& stop = launch_stop_event();
  if (stop) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64), (unsigned)lds, stream, launch_start_event(), stop, 0, tp);
    stop = nullptr;
    launch_start_event() = nullptr;
  } else {
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(tp);
  }
  ++tile_launch_count();
  note_analysis_kernel("lketkf_tile_kernel<%d, %d, %s>", R, UT, PAD ? "true" : "false");
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <int R, int UT>
static int lk_launch(const LkTileParams& tp, hipStream_t stream) {
  return tp.k == 4 * R ? lk_launch_p<R, UT, false>(tp, stream) : lk_launch_p<R, UT, true>(tp, stream);
}

// RBF kernel, float32, 2 <= k <= 40 members, the tile's union within 64 slots, every global access base + 32-bit byte offset
bool lketkf_tile_covers(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t ng, int64_t P) {
  if (!(m >= 1 && k >= 2 && k <= 40 && p_max >= 0 && extra_blocks >= 0)) return false;
  if ((int64_t)k * ldx * 4 >= ((int64_t)1 << 31) || (int64_t)k * ldo * 4 >= ((int64_t)1 << 31)) return false;
  if ((int64_t)k * (P > 0 ? P : 1) * 4 >= ((int64_t)1 << 31)) return false;
  const int ut = tile_ut_for(p_max) + extra_blocks;
  if (ut > 4 || ut > ((k + 15) >> 4) + 1) return false;       // (the workspace of the step driver is sized by the same rule)
  return ((ng + 15) >> 4) <= (int64_t)65536 * 65535;
}

int lketkf_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* Yb, const float* d,
                       int64_t P, const void* tile_lists, int ut, float inf_factor, float gamma, float* Xa, int64_t ldo,
                       int64_t o0, int32_t* flags, int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c,
                       hipStream_t stream, int seg_len, int64_t seg_stride, const Tile2Housekeeping* hk) {
  if (seg_len < 0 || (seg_len & 15) || (seg_len > 0 && ng >= ((int64_t)1 << 31))) return MIA_ERR_UNSUPPORTED;
  if (!flags || !retry_count || !tab_hdr || !tab_c || !tile_lists || (P > 0 && (!Yb || !d))) return MIA_ERR_UNSUPPORTED;
  if (ut < 1 || ut > 4 || !(gamma > 0.0f) || !lketkf_tile_covers(m, k, 0, ut - 1, ldx, ldo, ng, P)) return MIA_ERR_UNSUPPORTED;
  const TileListLayout L = tile_list_layout(ng, ut);
  const char* base = (const char*)tile_lists;
  LkTileParams tp;
  tp.X = X; tp.ldx = ldx; tp.m = m; tp.k = k; tp.g0 = g0; tp.ng = ng;
  tp.Yb = Yb; tp.d = d; tp.P = P;
  tp.thdr = (const int4*)(base + L.hdr); tp.tidx = (const int32_t*)(base + L.idx); tp.tD = (const f4w*)(base + L.D);
  const double rg = (double)(k - 1) / (double)inf_factor, km = (double)(k - 1);
  tp.inv_reg = (float)(1.0 / rg);
  tp.inv_k = (float)(1.0 / (double)k);
  tp.cs_phi = (float)(sqrt(km) / sqrt(rg));
  tp.cs_psi = (float)(1.0 / rg);
  tp.ng2 = (float)(-(double)gamma * 1.4426950408889634);
  tp.Xa = Xa; tp.ldo = ldo; tp.o0 = o0; tp.flags = flags; tp.retry_count = retry_count; tp.dmax = dmax;
  tp.tab_hdr = tab_hdr; tp.tab_c = tab_c;
  tp.seg_len = seg_len; tp.seg_stride = seg_stride;
  tp.clr_counts = hk ? hk->counts : nullptr; tp.clr_n = hk ? hk->n : nullptr; tp.clr_err = hk ? hk->err : nullptr;
  tp.err_out = hk ? hk->err_out : nullptr;
  const int r = k <= 8 ? 2 : (k <= 16 ? 4 : (k <= 24 ? 6 : (k <= 32 ? 8 : 10)));
#define MIA_LK_CASE(RR, UU) if (r == RR && ut == UU) return lk_launch<RR, UU>(tp, stream);
#ifdef MIA_LK_SINGLE          // (development builds: one instantiation, for register / ISA inspection)
  MIA_LK_CASE(10, 2)
#else
  MIA_LK_CASE(10, 1) MIA_LK_CASE(10, 2) MIA_LK_CASE(10, 3) MIA_LK_CASE(10, 4)
  MIA_LK_CASE(8, 1) MIA_LK_CASE(8, 2) MIA_LK_CASE(8, 3) MIA_LK_CASE(8, 4)
  MIA_LK_CASE(6, 1) MIA_LK_CASE(6, 2) MIA_LK_CASE(6, 3)
  MIA_LK_CASE(4, 1) MIA_LK_CASE(4, 2)
  MIA_LK_CASE(2, 1) MIA_LK_CASE(2, 2)
#endif
#undef MIA_LK_CASE
  return MIA_ERR_UNSUPPORTED;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_lketkf_rbf_analysis_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                                 const float* Yb, const float* d, int64_t P, const void* tile_lists,
                                                 int p_max, int extra_blocks, float inf_factor, float gamma, float* Xa,
                                                 int64_t ldo, int64_t o0, int32_t* flags, int32_t* retry_count, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 1 || k < 2 || g1 < g0 || g0 < 0 || P < 0 || p_max < 0 || extra_blocks < 0 || !(inf_factor > 0.0f) || !(gamma > 0.0f))
    return MIA_ERR_SIZE;
  if (g1 == g0) return MIA_OK;
  if (!X || !Xa || !tile_lists || !flags || !retry_count || (P > 0 && (!Yb || !d))) return MIA_ERR_NULL;
  if (!lketkf_tile_covers(m, k, p_max, extra_blocks, ldx, ldo, g1 - g0, P)) return MIA_ERR_UNSUPPORTED;
  const int2* th = nullptr;
  const float2* tc = nullptr;
  if (!cheb_primal_table(stream, &th, &tc)) return MIA_ERR_UNSUPPORTED;
  return lketkf_tile_launch(X, ldx, m, k, g0, g1 - g0, Yb, d, P, tile_lists, tile_ut_for(p_max) + extra_blocks, inf_factor, gamma,
                            Xa, ldo, o0, flags, retry_count, option(MIA_OPT_CHEB_DMAX), th, tc, stream, 0, 0, nullptr);
}

extern "C" int mia_letkf_tiles_cover(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t n_points,
                                     int64_t P, float gamma) {
  if (gamma > 0.0f) return lketkf_tile_covers(m, k, p_max, extra_blocks, ldx, ldo, n_points, P) ? 1 : 0;
  return (tile2_covers(m, k, p_max, extra_blocks, ldx, ldo, n_points) && tile2_records_addressable(k, P)) ? 1 : 0;
}

// Fused LETKF analysis, SIXTEEN grid points per wavefront, every contraction on the matrix cores.
//
// Same mathematics as letkf_cheb.hip (matrix functions of the local matrix applied by a Chebyshev recurrence, dual
// route: S_g = Yl^T Yl is p x p, reference: core/etkf.py:57-103 + interface/wrapper.py:86-98 + base.py:257-278), but
// the unit of work is a TILE of 16 consecutive grid points instead of one.  Neighbouring grid points see almost the
// same observations (config 2: 19 of 20 shared), and in the union index space of a tile
//
//     S_g = D_g G D_g,      G = Yw Yw^T  (U x U, U = |union of the tile's lists|),   D_g = diag(sqrt(rho_g)) (0 = not local)
//
// so ONE Gram product serves 16 points, and everything a point needs is a product with a 16-column right-hand side:
//
//     Z   = Yw X'                     (U x k)(k x 16)     z_g = D_g Z[:, g]
//     R   = |G| D                     Gershgorin bounds -> degree and interval of every point
//     T'  = 2 (alpha D o (G (D o T)) - T) - T''            the three-term recurrence, 16 points at a time
//     Xa' = Yw^T (D o Phi)            (k x U)(U x 16)
//
// Lane roles follow v_mfma_f32_16x16x4_f32: lane (lr, h) = (lane & 15, lane >> 4) owns COLUMN lr (= grid point lr of
// the tile) and, of every 16-row tile, rows 4h .. 4h+3.  A result tile therefore sits in the registers exactly where
// the next product wants its B operand, PROVIDED the summation index of that product is enumerated as
// slot(step (t, q), lane group h) = 16 t + 4 h + q -- any enumeration sums the same product, so none of the vectors of
// the recurrence ever leaves the registers (no LDS round trip, no cross-lane move); the Gram tiles themselves are,
// by symmetry, the A fragments of G in that enumeration.  Members are enumerated the same way in Z and Xa', which puts
// x' in the registers where the output tile needs it.
//
// Summation order is canonical: union slots are assigned by RANK of the observation index (hash-dedupe, then a counting
// rank), permuted so that the enumeration above visits them in ascending rank.  A grid point's own observations are thus
// always summed in ascending index order with exact zeros in between, whatever else is in the tile: on the f32 products
// (SPL = false) results do not depend on tile composition, shard boundaries or launch geometry, bit for bit.
//
// SPL = true (the default route, letkf_tile_split.hip): the same products as THREE half-precision MFMAs each
// (v_mfma_f32_16x16x32_f16 on operands carried as pairs of halves, see "split precision" below) -- f32 accuracy at a
// fifth of the matrix-pipe time, beside the vector unit instead of in its way.  The operand scale is the tile's, so there a
// point's result depends on its tile at rounding level.
//
// A tile whose union exceeds the 16 UT slots the instantiation has (UT <= 6: 96 slots) is processed in halves (quarters, ...)
// -- one point always fits (p_max <= 16 UT is checked on the host); degree cap / non-finite / overflow / retry protocol are those
// of letkf_cheb.hip (MIA_FLAG_RETRY points are redone by the eigensolver kernel).
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"

namespace mia {

using f4t = __attribute__((ext_vector_type(4))) float;
using f2v = __attribute__((ext_vector_type(2))) float;
using h2t = __attribute__((ext_vector_type(2))) _Float16;
using h8t = __attribute__((ext_vector_type(8))) _Float16;
using u4t = __attribute__((ext_vector_type(4))) unsigned;

#define MIA_TILE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

struct TileParams {
  const float* X; int64_t ldx; int m; int k; int kp;
  int64_t g0, ng;
  const float* rec;
  const int32_t* cnt; const int32_t* idx; const void* w; int w_f32; int p_cap; int p_max;
  float reg, inv_reg, f0, inv_k;
  float* Xa; int64_t ldo, o0; int32_t* flags; int32_t* retry_count;
  int dmax;
  const int2* tab_hdr; const float2* tab_c; float cs_phi, cs_psi;
  int kpv_magic;       // ceil(2^20 / (kp / 4))
  // segmented launch (native step driver, see letkf_cheb_seg_kernel): the ng points are seg_len-sized segments, segment s
  // writes its own (m k, seg_len) buffer at Xa + s * seg_stride with write-through stores and counts its finished POINTS
  // in done[(s * 64 + j) * kSlotStride]; tiles never straddle a segment
  int seg_len; int64_t seg_stride; int32_t* done;
  // split-precision variant (SPL): a union record occupies nc chunks of eight values (hi halves | lo halves, 32 bytes),
  // rows rsb = 32 nc + 16 bytes apart (an odd multiple of 16: sixteen rows read side by side touch every bank once)
  int nc, rsb;
  int lds_bytes;       // (diagnostic builds)
};

__device__ __forceinline__ float tile_add_h(float v) {       // sum over the four lanes (lr, h = 0..3), in every one of them
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float tile_max_h(float v) {       // maximum over the same four lanes (NaN-propagating by bits
  typedef unsigned u2v __attribute__((ext_vector_type(2)));   //  for values >= +0: integer compare, see mia_common.h)
  unsigned u = __float_as_uint(v);
  u2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = r.x > r.y ? r.x : r.y;
  r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = r.x > r.y ? r.x : r.y;
  return __uint_as_float(u);
}

__device__ __forceinline__ unsigned tile_wave_max_u32(unsigned u) {     // wave-uniform maximum (DPP, see mia_common.h)
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xf, 0xf, false); u = u > t ? u : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false); u = u > t ? u : t;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)u, 0), b = (unsigned)__builtin_amdgcn_readlane((int)u, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)u, 32), d = (unsigned)__builtin_amdgcn_readlane((int)u, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// ---- split precision (SPL instantiations) ---------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 occupies the SIMD for 33 cycles per 2048 flop and shares the FP32 pipe with the vector
// instructions (measured: tools/mfma_rate.hip -- the two never overlap, one wavefront or several).
// v_mfma_f32_16x16x32_f16 takes 16 cycles for EIGHT times the summation depth and runs beside vector instructions.  An
// f32 operand x is carried as two halves, hi = f16(x), lo = f16(x - hi) (round to nearest both times: hi + lo holds
// 22-23 significant bits of x, products of halves are exact in the f32 accumulator), and a product A B becomes
// Ah Bh + Ah Bl + Al Bh (the term Al Bl is below 2^-22 of the result).  Operands are first scaled by a power of two
// (exact) into the middle of the f16 range; results are unscaled by the inverse.  Accuracy equals the f32 route's
// (tools/split_emul.py: 1.65e-7 against 1.52e-7 relative error of the recurrence's output over 40 random tiles).
__device__ __forceinline__ void tile_split8(const float (&x)[8], h8t& hi, h8t& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2v v = {x[2 * i], x[2 * i + 1]};
    const h2t a = __builtin_convertvector(v, h2t);                  // v_cvt_pk_f16_f32, round to nearest
    const unsigned au = __builtin_bit_cast(unsigned, a);
    // the remainders x - hi straight from the packed halves (v_fma_mix_f32 reads an f16 operand in place; the compiler's
    // own form is two conversions + a packed subtraction, and packed f32 instructions are slow beside MFMAs)
    // (written over a COPY of x: the copy is the compiler's instruction, so the wait states behind a matrix instruction that
    //  wrote or read the register it picks are the compiler's to pad; a fresh output of the assembly itself was found 5-6
    //  states behind such a write by tools/check_mfma_hazards.py)
    f2v r = v;
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r[0]) : "v"(au));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[1]) : "v"(au));
    const h2t b = __builtin_convertvector(r, h2t);
    hi[2 * i] = a[0]; hi[2 * i + 1] = a[1];
    lo[2 * i] = b[0]; lo[2 * i + 1] = b[1];
  }
}
__device__ __forceinline__ h8t tile_hi8(const float (&x)[8]) {
  h8t hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2v v = {x[2 * i], x[2 * i + 1]};
    const h2t a = __builtin_convertvector(v, h2t);
    hi[2 * i] = a[0]; hi[2 * i + 1] = a[1];
  }
  return hi;
}
__device__ __forceinline__ f4t tile_mfma3(f4t acc, const h8t ah, const h8t al, const h8t bh, const h8t bl) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}
// power of two that brings a magnitude (given by its bit pattern, sign cleared) to [2^target, 2^(target+1)); the
// exponent of the scale is returned too.  Zero / subnormal magnitudes are left alone; exponents are clamped to +-60 (the
// squares and products of scales formed below then stay inside f32).
__device__ __forceinline__ float tile_pow2_scale(unsigned magbits, int target, int* es_out) {
  const int e = (int)(magbits >> 23) - 127;
  int es = (magbits >> 23) == 0u ? 0 : target - e;
  es = es < -60 ? -60 : (es > 60 ? 60 : es);
  *es_out = es;
  return __uint_as_float((unsigned)(127 + es) << 23);
}

// Every global access of the kernel is `wave-uniform base + 32-bit lane offset in bytes` (the saddr + voffset form of
// global_load / global_store): the host checks that the offsets fit, and no 64-bit address ever lives in vector registers.
template <typename T>
__device__ __forceinline__ T ld_off(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// In-kernel phase stamps (diagnostic builds only, tools/tile_stamps.py): -DMIA_TILE_STAMPS compiles them in; the stamp
// values go to a buffer of their own that nothing else reads.
#ifdef MIA_TILE_STAMPS
constexpr int kStampN = 12, kStampTiles = 8192;
__device__ long long g_tile_stamps[kStampTiles * kStampN];
#define MIA_STAMP(i) do { if (lane0 == 0 && bid < kStampTiles) g_tile_stamps[bid * kStampN + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
// slot 9: where the wavefront ran (HW_ID in the low word: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13; XCC_ID in the high word)
#define MIA_STAMP_HWID() do { if (lane0 == 0 && bid < kStampTiles) g_tile_stamps[bid * kStampN + 9] = \
    (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } while (0)
// slots 10, 11: the constant 100 MHz counter (comparable across CUs, which s_memtime is not) at start and end of the wavefront
#define MIA_STAMP_REAL(i) do { if (lane0 == 0 && bid < kStampTiles) g_tile_stamps[bid * kStampN + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MIA_STAMP_HWID() do { } while (0)
#define MIA_STAMP_REAL(i) do { } while (0)
#define MIA_STAMP(i) do { } while (0)
#endif

#ifndef MIA_TILE_WAVES_UT2
#define MIA_TILE_WAVES_UT2 3
#endif
#ifndef MIA_TILE_WAVES_SPLIT_UT2
#define MIA_TILE_WAVES_SPLIT_UT2 2
#endif

template <int UT, int KT, bool SEG, bool SPL>
// Wavefronts per SIMD.  f32 products, UT <= 2: three (168 registers; four spilled and measured no faster).  Split products:
// TWO -- the three-wave build spilled 24 registers (40 MB of scratch traffic per launch, twice the kernel's own) and was no
// faster alone (58 us either way: the half-precision MFMAs leave the kernel short of vector issue slots, not of waves); with
// two, a wave lives 13.8 us instead of 19 and the registers left over let the preparation kernels of later steps run beside
// it without displacing its waves: pipelined step 0.101 -> 0.094 ms.  (Either way the 106 tiles beyond the last full round
// -- 6250 = 3 x 2048 + 106 -- run alone for the last ~12 us of a stand-alone launch: tools/tile_stamps.py.)
__global__ __launch_bounds__(64, (UT <= 2 ? (SPL ? MIA_TILE_WAVES_SPLIT_UT2 : MIA_TILE_WAVES_UT2) : (UT == 3 ? 2 : 1)))   /* UT >= 4: one wavefront per SIMD, up to 512 registers */ void letkf_tile_kernel(TileParams P) {
  constexpr int UMAX = 16 * UT, NU = 4 * UT;
  constexpr int NB = (KT + 1) / 2, NKB = (UT + 1) / 2;       // SPL: blocks of 32 members / of 32 union slots (= two row blocks)
  constexpr int LOGHS = UT <= 1 ? 6 : (UT <= 2 ? 7 : 8), HS = 1 << LOGHS, HR = HS / 64;
  constexpr int DS = UMAX + 4;
  constexpr int CLraw = (16 * DS + 2 * HS + UMAX + 2 * (UMAX > 64 ? UMAX : 64)) / 32;
  constexpr int CL = CLraw > kTabDeg ? kTabDeg : CLraw;      // degrees whose coefficients fit the union scratch (31 at UT = 2)
  constexpr int CQ = (CL + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane0 = threadIdx.x;
  const int k = P.k, kp = P.kp, pm = P.p_max;
  float* Yw = reinterpret_cast<float*>(smem_raw);            // [UMAX][kp] union records (+ 16 zero floats)
  unsigned char* YwB = smem_raw;                             // SPL: [UMAX] rows of P.rsb bytes, see TileParams
  const unsigned RSB = (unsigned)P.rsb;
  float* Dl = SPL ? reinterpret_cast<float*>(smem_raw + UMAX * P.rsb)
                  : Yw + UMAX * kp + 16;                     // [16][DS]   sqrt(rho) of (point, slot), 0 = not local
  int* H = reinterpret_cast<int*>(Dl + 16 * DS);             // [HS]       hash table of observation indices / index bitmap
  int* Hs = H + HS;                                          // [HS]       slot of a table position
  int* ukey = Hs + HS;                                       // [UMAX]     observation index of a slot, -1 = unused
  constexpr int UC = UMAX > 64 ? UMAX : 64;
  int* comp = ukey + UMAX;                                   // [UC]       compacted keys
  int* cpos = comp + UC;                                     // [UC]       their table positions
  float2* Cl = reinterpret_cast<float2*>(Dl);                // [CL][16]   scaled Chebyshev coefficients (reuses Dl .. cpos)

  // XCD-aware block -> tile map: blocks b, b + 8, ... share an XCD (and its L2) and take consecutive tiles, whose
  // records overlap
  const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  int64_t p0;            // first point of the tile (index into the launch's ng points)
  int npts;
  int64_t oc0;           // output column of the tile's first point
  float* Xab = P.Xa;
  int sg = 0;
  if constexpr (SEG) {
    const int64_t tps = ((int64_t)P.seg_len + 15) >> 4;               // tiles per segment
    const int64_t nseg = (P.ng + P.seg_len - 1) / P.seg_len;
    if (bid >= tps * nseg) return;
    sg = (int)(bid / tps);
    const int64_t lt = bid - (int64_t)sg * tps;
    const int64_t s0 = (int64_t)sg * P.seg_len;
    const int64_t slen = P.ng - s0 < P.seg_len ? P.ng - s0 : P.seg_len;
    p0 = s0 + (lt << 4);
    npts = slen - (lt << 4) < 16 ? (int)(slen - (lt << 4)) : 16;
    if (npts <= 0) return;
    oc0 = lt << 4;
    Xab += (int64_t)sg * P.seg_stride;
  } else {
    const int64_t ntile = (P.ng + 15) >> 4;
    if (bid >= ntile) return;
    const int64_t q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
    const int64_t tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    p0 = tile << 4;
    npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
    oc0 = P.o0 + p0;
  }
#ifdef MIA_TILE_POISON
  // diagnostic builds: every LDS byte starts as NaN (f32 and f16), so that a read of unwritten storage shows
  for (unsigned i = (unsigned)lane0; i < (unsigned)P.lds_bytes / 4u; i += 64u) reinterpret_cast<unsigned*>(smem_raw)[i] = 0xffffffffu;
  MIA_TILE_SYNC();
#endif
  MIA_STAMP(0);
  MIA_STAMP_HWID();
  MIA_STAMP_REAL(10);

  // ---- the tile's neighbour lists: lane (lp, sub) = (lane >> 2, lane & 3) holds entries sub, sub + 4, ... of point lp.
  //      Count and entries are requested before anything is waited for.  The entries are loop-carried: loaded here for
  //      the first pass and, at the bottom of the loop, for the next one (only a split tile has one) -- dead in between.
  const int nl = pm < P.p_cap ? pm : P.p_cap;
  int eidx[NU];
  double ewd[NU];        // weights as loaded (float64 lists); converted where they are used -- a conversion next to its load
                         // would wait for that load before the next one is requested
  auto load_lists = [&](int64_t pt0, int lp, int sub) {
    const unsigned rowb = (unsigned)(lp < npts ? lp : 0) * (unsigned)P.p_cap;
    const int32_t* ib = P.idx + pt0 * P.p_cap;
    const double* wb = reinterpret_cast<const double*>(P.w) + pt0 * P.p_cap;
#pragma unroll
    for (int u = 0; u < NU; ++u) {           // unconditional loads inside the row's storage (no branch, no predicate to keep);
      const int pos = sub + 4 * u;           // masked by mask_lists
      const unsigned e = rowb + (unsigned)(pos < nl ? pos : 0);
      eidx[u] = ld_off<int32_t>(ib, e * 4u);
      ewd[u] = ld_off<double>(wb, e * 8u);
    }
  };
  // entries beyond the point's count (and those of points that are not analysed) become index -1: the ONE validity test of
  // everything that follows
  auto mask_lists = [&](int sub, int cnt_) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (sub + 4 * u >= cnt_) eidx[u] = -1;
  };
  auto entry_weight = [&](int u) -> float { return float(ewd[u]); };
  int lcnt;
  unsigned long long badmask;
  {
    const int lp = lane0 >> 2, sub = lane0 & 3;
    lcnt = ld_off<int32_t>(P.cnt + p0, (unsigned)(lp < npts ? lp : 0) * 4u);
    load_lists(p0, lp, sub);
    const bool pbad = lp < npts && (lcnt > pm || lcnt > P.p_cap || lcnt > UMAX);   // loud failure, never truncate
    if (pbad) {
      if (sub == 0) P.flags[p0 + lp] = MIA_FLAG_OVERFLOW;
      const float nanv = __builtin_nanf("");
      for (int it = sub; it < P.m * k; it += 4) {
        if constexpr (SEG) __hip_atomic_store(&Xab[(int64_t)it * P.ldo + oc0 + lp], nanv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else Xab[(int64_t)it * P.ldo + oc0 + lp] = nanv;
      }
    }
    if (lp >= npts || pbad) lcnt = 0;
    badmask = __ballot(pbad);
    mask_lists(sub, lcnt);
  }

  int lo = 0;
#pragma clang loop unroll(disable)
  while (lo < npts) {
    // The loop body runs once per tile unless the tile had to be split.  Tile origin and lane id go through opaque
    // copies, so that the compiler does not hoist dozens of address / predicate registers out of a loop that does not
    // loop (they lived across the whole body and pushed the matrix phases into scratch).
    int64_t p0v = p0, oc0v = oc0;
    int lane = lane0;
    asm volatile("" : "+s"(p0v), "+s"(oc0v), "+v"(lane));
    const int lr = lane & 15, h = lane >> 4, lp = lane >> 2, sub = lane & 3;
    const bool colok = lr < npts && !((badmask >> (4 * lr)) & 1ull);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int lrc = lr < npts ? lr : npts - 1;                      // a column that exists (clamped, unconditional loads)
    const unsigned ldxb = (unsigned)P.ldx * 4u, ldob = (unsigned)P.ldo * 4u;       // (k ld 4 < 2^31: checked on the host)
    const unsigned xlane = (unsigned)(4 * h) * ldxb + (unsigned)lrc * 4u;
    // ---- a state row of the tile: member (tm, q) of lane group h = 16 tm + 4 h + q, column lr.  Unconditional loads
    //      (clamped to the last member / an existing column), requested now and consumed after the union is built
    auto load_x = [&](int mi, f4t (&xr)[KT]) {
      const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0v;
#pragma unroll
      for (int tm = 0; tm < KT; ++tm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned off = xlane + (unsigned)(16 * tm + q) * ldxb;
          if (tm == KT - 1) {
            const int mem = 16 * tm + 4 * h + q;
            off = (unsigned)(mem < k ? mem : k - 1) * ldxb + (unsigned)lrc * 4u;
          }
          xr[tm][q] = ld_off<float>(xbase, off);
        }
    };
    // SPL: member (b, i) of lane group h = 32 b + 8 h + i (the summation index of a 32-deep product: lane group h supplies
    // eight consecutive values), column lr
    // (ONE lane offset per member block -- the lane group's first member -- and a wave-uniform row pointer that walks the
    //  eight members: sixteen per-lane offsets would be hoisted out of the state-row loop and spilled.  A chunk without
    //  any member reads the block's first chunk instead; an ensemble size that is not a multiple of eight clamps per value)
    auto load_xs = [&](int mi, int hh, float (&xr)[NB][8]) {
      const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0v;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float* pr = xbase + (int64_t)(32 * b) * P.ldx;
        if (b < NB - 1 || (k & 7) == 0) {
          const unsigned vo = (unsigned)(32 * b + 8 * hh < k ? 8 * hh : 0) * ldxb + (unsigned)lrc * 4u;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xr[b][i] = ld_off<float>(pr, vo);
            pr += P.ldx;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int mem = 32 * b + 8 * hh + i;
            xr[b][i] = ld_off<float>(xbase, (unsigned)(mem < k ? mem : k - 1) * ldxb + (unsigned)lrc * 4u);
          }
        }
      }
    };
    f4t xb[KT];
    float xsb[NB][8];
    if constexpr (SPL) load_xs(0, h, xsb);
    else load_x(0, xb);
    MIA_STAMP(1);

    // ---- union of the lists of points [lo, hi), slots by RANK of the observation index (the enumeration order of the
    //      products = ascending rank); shrink the range until the union fits
    int n = 16, hi, U;
    unsigned ymax_bits = 0u;
    int es[NU];            // slot of this lane's entries
    for (;;) {
      hi = lo + n < npts ? lo + n : npts;
      const bool act = lp >= lo && lp < hi;
      for (int i = lane; i < UMAX; i += 64) ukey[i] = -1;
      unsigned mx1 = 0u, mninv = 0u;             // range of the observation indices (wave-uniform after the reductions)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        es[u] = -1;
        const bool v = act && eidx[u] >= 0;
        const unsigned key = (unsigned)eidx[u];
        mx1 = (v && key + 1u > mx1) ? key + 1u : mx1;
        mninv = (v && ~key > mninv) ? ~key : mninv;
      }
      mx1 = tile_wave_max_u32(mx1);
      mninv = tile_wave_max_u32(mninv);
      const int ibase = (int)~mninv;
      if (mx1 == 0u) {
        U = 0;                                       // no local observation anywhere in the range
        MIA_TILE_SYNC();
      } else if ((int)mx1 - 1 - ibase < 256) {
        // -- fast path (1-D / index-sorted networks): the indices span < 256 values.  A 256-bit map in LDS (atomic OR, no
        //    return value) that every lane reads back whole: rank = set bits below, no dedupe, no sort
        unsigned long long* bm = reinterpret_cast<unsigned long long*>(H);
        if (lane < 4) bm[lane] = 0ull;
        MIA_TILE_SYNC();
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (act && eidx[u] >= 0) {
            const unsigned off = (unsigned)(eidx[u] - ibase);
            atomicOr(reinterpret_cast<unsigned*>(H) + (off >> 5), 1u << (off & 31u));
          }
        MIA_TILE_SYNC();
        const unsigned long long m0 = bm[0], m1 = bm[1], m2 = bm[2], m3 = bm[3];
        const int c0 = __popcll(m0), c1 = c0 + __popcll(m1), c2 = c1 + __popcll(m2);
        // (the count is the same in every lane but comes from vector registers: as a scalar, the "step beyond the union"
        //  tests of the products below are s_cmp / s_cbranch instead of eight 64-bit lane masks kept in scalar registers)
        U = __builtin_amdgcn_readfirstlane(c2 + __popcll(m3));
        if (U > UMAX) { n >>= 1; continue; }      // (n = 1 always fits: a single list has at most UMAX entries)
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (act && eidx[u] >= 0) {
            const unsigned off = (unsigned)(eidx[u] - ibase), wsel = off >> 6;
            const unsigned long long mw = wsel == 0 ? m0 : (wsel == 1 ? m1 : (wsel == 2 ? m2 : m3));
            const int pre = wsel == 0 ? 0 : (wsel == 1 ? c0 : (wsel == 2 ? c1 : c2));
            const int rk = pre + __popcll(mw & ((1ull << (off & 63u)) - 1ull));
            es[u] = 16 * (rk >> 4) + 4 * (rk & 3) + ((rk >> 2) & 3);
            ukey[es[u]] = eidx[u];                   // (lanes that share an observation write the same value)
          }
        MIA_TILE_SYNC();
      } else {
        // -- general path: dedupe in a hash table, rank by counting
#pragma unroll
        for (int r = 0; r < HR; ++r) H[lane + 64 * r] = -1;
        MIA_TILE_SYNC();
        int full = 0;
        int epos[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          epos[u] = 0;
          if (act && eidx[u] >= 0) {
            unsigned hh = ((unsigned)eidx[u] * 2654435761u) >> (32 - LOGHS);
            int it = 0;
#pragma clang loop unroll(disable)
            for (; it < HS; ++it) {           // bounded: a table that fills up means the union cannot fit anyway
              const int old = atomicCAS(&H[hh], -1, eidx[u]);
              if (old == -1 || old == eidx[u]) break;
              hh = (hh + 1) & (HS - 1);
            }
            if (it == HS) full = 1;
            epos[u] = (int)hh;
          }
        }
        MIA_TILE_SYNC();
        int mykey[HR], myci[HR], tot = 0;
#pragma unroll
        for (int r = 0; r < HR; ++r) {
          mykey[r] = H[lane * HR + r];
          const bool occ = mykey[r] != -1;
          const unsigned long long mask = __ballot(occ);
          myci[r] = occ ? tot + __popcll(mask & lt_mask) : -1;
          tot += __popcll(mask);
        }
        U = __builtin_amdgcn_readfirstlane(__any(full) ? (1 << 20) : tot);
        if (U > UMAX) { n >>= 1; continue; }
#pragma unroll
        for (int r = 0; r < HR; ++r)
          if (myci[r] >= 0) { comp[myci[r]] = mykey[r]; cpos[myci[r]] = lane * HR + r; }
        MIA_TILE_SYNC();
        for (int i = lane; i < U; i += 64) {
          const int key = comp[i];
          int rk = 0;
          for (int j = 0; j < U; ++j) rk += comp[j] < key ? 1 : 0;
          const int slot = 16 * (rk >> 4) + 4 * (rk & 3) + ((rk >> 2) & 3);
          ukey[slot] = key;
          Hs[cpos[i]] = slot;
        }
        MIA_TILE_SYNC();
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (act && eidx[u] >= 0) es[u] = Hs[epos[u]];
      }
      MIA_STAMP(2);
      // ---- the union's records, unscaled (the sqrt(rho) factors differ per point: they live in D)
      float fin = 0.0f;       // stays 0 while every value is finite (inf * 0 = NaN)
      unsigned mxi = 0u;      // SPL: largest magnitude of the union's records as a bit pattern (NaN > inf > finite)
      {
        const unsigned kpv = (unsigned)kp >> 2;
        const int total = UMAX * (int)kpv;
        float4* Yw4 = reinterpret_cast<float4*>(Yw);
        constexpr int GQ = 4;
        for (int base = 0; base < total; base += 64 * GQ) {
          float4 v[GQ];
          int kk[GQ];
#pragma unroll
          for (int u = 0; u < GQ; ++u) {
            const unsigned it = base + 64 * u + lane < total ? (unsigned)(base + 64 * u + lane) : 0u;
            const unsigned j = (it * (unsigned)P.kpv_magic) >> 20;
            kk[u] = ukey[j];
            // (unconditional load -- a predicated one serialises the GQ requests -- from an address that is valid also for
            //  unused slots and for P = 0, where there is no record array at all: the coefficient table)
            // (the one 64-bit lane address of the kernel: the record array may exceed 4 GB)
            const float4* src = kk[u] < 0 ? reinterpret_cast<const float4*>(P.tab_c)
                                          : reinterpret_cast<const float4*>(P.rec) + ((uint64_t)(unsigned)kk[u] * kpv + (it - j * kpv));
            v[u] = *src;
          }
#pragma unroll
          for (int u = 0; u < GQ; ++u) {
            const int it = base + 64 * u + lane;
            if (it < total) {
              const float4 t = kk[u] < 0 ? float4{0.f, 0.f, 0.f, 0.f} : v[u];
              if constexpr (SPL) {
                const unsigned a0 = __float_as_uint(t.x) & 0x7fffffffu, a1 = __float_as_uint(t.y) & 0x7fffffffu;
                const unsigned a2 = __float_as_uint(t.z) & 0x7fffffffu, a3 = __float_as_uint(t.w) & 0x7fffffffu;
                const unsigned m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3, m4 = m01 > m23 ? m01 : m23;
                mxi = m4 > mxi ? m4 : mxi;
                const unsigned j = ((unsigned)it * (unsigned)P.kpv_magic) >> 20;
                *reinterpret_cast<float4*>(YwB + j * RSB + ((unsigned)it - j * kpv) * 16u) = t;
              } else {
                fin = fmaf(t.x, 0.0f, fmaf(t.y, 0.0f, fmaf(t.z, 0.0f, fmaf(t.w, 0.0f, fin))));
                Yw4[it] = t;
              }
            }
          }
        }
        if constexpr (!SPL) { if (lane < 4) Yw4[total + lane] = float4{0.f, 0.f, 0.f, 0.f}; }
      }
      if constexpr (SPL) {
        mxi = tile_wave_max_u32(mxi);
        fin = mxi >= 0x7f800000u ? __builtin_nanf("") : 0.0f;
        ymax_bits = mxi;
      }
      // A non-finite record would reach EVERY column of the tile through the shared Gram matrix (NaN * 0 = NaN), also
      // the points that do not see that observation.  Such a tile is analysed point by point: the union is then the
      // point's own list and the damage stays where the reference has it (the points that use the observation).
      if (__any(fin != fin) && hi - lo > 1) { n = 1; continue; }
      break;
    }
    MIA_STAMP(3);
    // SPL: the union's records become scaled half pairs, in place (a chunk of eight f32 = 32 bytes -> 8 hi | 8 lo halves).
    // One power of two for the tile, from the largest magnitude the gather saw; two lanes share a row.
    float inv_sy = 1.0f;
    int esy = 0;
    if constexpr (SPL) {
      const float sy = tile_pow2_scale(ymax_bits, 9, &esy);
      inv_sy = __uint_as_float((unsigned)(127 - esy) << 23);
      // (every row: a row the union does not use holds zeros, but the lo halves of its last chunk would be read from the
      //  padding behind the record, which nothing writes)
      const int rows = UMAX, nc = P.nc;
      MIA_TILE_SYNC();
#pragma clang loop unroll(disable)
      for (int r0 = 0; r0 < rows; r0 += 32) {
        const int r = r0 + (lane >> 1);
        if (r < rows) {
#pragma clang loop unroll(disable)
          for (int c = lane & 1; c < nc; c += 2) {
            unsigned char* q = YwB + (unsigned)r * RSB + (unsigned)c * 32u;
            const f4t a = *reinterpret_cast<const f4t*>(q), b = *reinterpret_cast<const f4t*>(q + 16);
            const float x[8] = {a[0] * sy, a[1] * sy, a[2] * sy, a[3] * sy, b[0] * sy, b[1] * sy, b[2] * sy, b[3] * sy};
            h8t hh, ll;
            tile_split8(x, hh, ll);
            *reinterpret_cast<h8t*>(q) = hh;
            *reinterpret_cast<h8t*>(q + 16) = ll;
          }
        }
      }
    }
    for (int i = lane; i < 4 * DS; i += 64) reinterpret_cast<f4t*>(Dl)[i] = f4t{0.f, 0.f, 0.f, 0.f};
    MIA_TILE_SYNC();
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (es[u] >= 0) Dl[lp * DS + es[u]] = entry_weight(u);
    MIA_TILE_SYNC();
    const bool colact = colok && lr >= lo && lr < hi;
    f4t dreg[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) dreg[t] = *reinterpret_cast<const f4t*>(Dl + lr * DS + 16 * t + 4 * h);
    MIA_STAMP(4);

    f4t G[UT][UT];          // G[t1][t2][q] = Gram[16 t1 + 4 h + q][16 t2 + lr]
    h8t GAh[UT][NKB], GAl[UT][NKB];   // SPL: the same matrix as A fragments of the 32-deep products, 2^-16 G as half pairs
    float alpha = 0.0f;
    int deg = 0, tab_idx = 0, degmax = 0, pflag = 0;
    bool decl = false;
    for (int mi = 0; mi < P.m; ++mi) {
      // (lane roles through opaque copies once more: what is invariant in this loop -- addresses, predicates of the
      //  first-row-only phases -- would otherwise be hoisted in front of it and spilled there)
      int hv = h, lrv = lr;
      asm volatile("" : "+v"(hv), "+v"(lrv));
      float xm;
      float inv_sx = 1.0f;          // SPL: 1 / (power of two that scaled this column's x')
      h8t xh[NB], xl[NB];           // SPL: x' of the column as half pairs, member block b
      if constexpr (SPL) {
        if (mi > 0) load_xs(mi, hv, xsb);
        float xs = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool live = colact && (b < NB - 1 || 32 * b + 8 * hv + i < k);       // (only the last member block is ragged)
            xsb[b][i] = live ? xsb[b][i] : 0.0f;
            xs += xsb[b][i];
          }
        xm = tile_add_h(xs) * P.inv_k;
        unsigned xmax = 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool live = colact && (b < NB - 1 || 32 * b + 8 * hv + i < k);
            xsb[b][i] = live ? xsb[b][i] - xm : 0.0f;
            const unsigned a = __float_as_uint(xsb[b][i]) & 0x7fffffffu;
            xmax = a > xmax ? a : xmax;
          }
        xmax = __float_as_uint(tile_max_h(__uint_as_float(xmax)));
        int esx;
        const float sx = tile_pow2_scale(xmax, 9, &esx);
        inv_sx = __uint_as_float((unsigned)(127 - esx) << 23);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float t8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) t8[i] = xsb[b][i] * sx;
          tile_split8(t8, xh[b], xl[b]);
        }
      } else {
        if (mi > 0) load_x(mi, xb);
        float xs = 0.0f;
#pragma unroll
        for (int tm = 0; tm < KT; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool live = colact && (tm < KT - 1 || 16 * tm + 4 * hv + q < k);       // (only the last member block is ragged)
            xb[tm][q] = live ? xb[tm][q] : 0.0f;
            xs += xb[tm][q];
          }
        xm = tile_add_h(xs) * P.inv_k;
#pragma unroll
        for (int tm = 0; tm < KT; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool live = colact && (tm < KT - 1 || 16 * tm + 4 * hv + q < k);
            xb[tm][q] = live ? xb[tm][q] - xm : 0.0f;
          }
      }
      // ---- G = Yw Yw^T (first row only) and Z = Yw X'
      f4t Z[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) Z[t] = f4t{0.f, 0.f, 0.f, 0.f};
      if (mi == 0) {
#pragma unroll
        for (int t1 = 0; t1 < UT; ++t1)
#pragma unroll
          for (int t2 = 0; t2 < UT; ++t2) G[t1][t2] = f4t{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (SPL) {
        // lane (lr, h) reads, of row 16 t + lr, the chunk of members 32 b + 8 h .. + 7: A fragment of row block t and, the
        // product being symmetric, B fragment of column block t
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          h8t ah[UT], al[UT];
          const int chunk = 4 * b + hv;
          const int vc = k - (32 * b + 8 * hv);          // members of this chunk that exist (the rest: innovation, padding)
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            const unsigned ro = (unsigned)(16 * t + lrv) * RSB + (unsigned)(chunk < P.nc ? chunk : 0) * 32u;
            u4t wh = *reinterpret_cast<const u4t*>(YwB + ro), wl = *reinterpret_cast<const u4t*>(YwB + ro + 16);
            if (b == NB - 1) {
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const int lim = vc - 2 * w;
                const unsigned msk = lim >= 2 ? 0xffffffffu : (lim == 1 ? 0x0000ffffu : 0u);
                wh[w] &= msk;
                wl[w] &= msk;
              }
            }
            ah[t] = __builtin_bit_cast(h8t, wh);
            al[t] = __builtin_bit_cast(h8t, wl);
          }
          if (mi == 0) {
#pragma unroll
            for (int t2 = 0; t2 < UT; ++t2)
#pragma unroll
              for (int t1 = 0; t1 < UT; ++t1) G[t1][t2] = tile_mfma3(G[t1][t2], ah[t1], al[t1], ah[t2], al[t2]);
          }
#pragma unroll
          for (int t = 0; t < UT; ++t) Z[t] = tile_mfma3(Z[t], ah[t], al[t], xh[b], xl[b]);
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < KT; ++tm) {
          f4t av[UT];
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            av[t] = *reinterpret_cast<const f4t*>(Yw + (16 * t + lrv) * kp + 16 * tm + 4 * hv);
            if (tm == KT - 1) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (16 * tm + 4 * hv + q >= k) av[t][q] = 0.0f;        // innovation / pad columns, next row's start
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (mi == 0) {
#pragma unroll
              for (int t2 = 0; t2 < UT; ++t2)
#pragma unroll
                for (int t1 = 0; t1 < UT; ++t1)
                  G[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t1][q], av[t2][q], G[t1][t2], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < UT; ++t) Z[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][q], xb[tm][q], Z[t], 0, 0, 0);
          }
        }
      }
      MIA_STAMP(5);
      const int esg = 16 - 2 * esy;        // SPL: (matrix the products see) = 2^-esg (true Gram matrix)
      if (mi == 0) {
        // ---- Gershgorin bound of every point: L_g = max_a w_a sum_b |G_ab| w_b, then degree / interval from the table
        //      (step (tk, q) of a product over the union covers ranks 16 tk + 4 q .. + 3: steps beyond the union are
        //       skipped -- config 2: 28 observations, 7 of 8 steps)
        f4t R[UT];
#pragma unroll
        for (int t = 0; t < UT; ++t) R[t] = f4t{0.f, 0.f, 0.f, 0.f};
        float L = 0.0f;
        if constexpr (SPL) {
          // A fragments of G for the 32-deep products: lane group h supplies slots 16 (2 kb + tt) + 4 h + q, i.e. the values
          // this lane holds of the tiles (2 kb, t) and (2 kb + 1, t) -- no data moves.  |G| times D needs the hi halves only
          // (a bound: the rounding of both operands is covered by the margin below)
#pragma unroll
          for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              float gv[8];
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[4 * tt + q] = 2 * kb + tt < UT ? G[2 * kb + tt < UT ? 2 * kb + tt : 0][t][q] * 0x1p-16f : 0.0f;
              tile_split8(gv, GAh[t][kb], GAl[t][kb]);
            }
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
            if (32 * kb < U) {
              float dv[8];
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) dv[4 * tt + q] = 2 * kb + tt < UT ? dreg[2 * kb + tt < UT ? 2 * kb + tt : 0][q] : 0.0f;
              const h8t dh = tile_hi8(dv);
#pragma unroll
              for (int t = 0; t < UT; ++t) {
                u4t ag = __builtin_bit_cast(u4t, GAh[t][kb]);
                ag &= 0x7fff7fffu;
                R[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8t, ag), dh, R[t], 0, 0, 0);
              }
            }
        } else {
#pragma unroll
          for (int tk = 0; tk < UT; ++tk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * tk + 4 * q < U) {
#pragma unroll
                for (int t = 0; t < UT; ++t)
                  R[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fabsf(G[tk][t][q]), dreg[tk][q], R[t], 0, 0, 0);
              }
        }
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = dreg[t][q] * R[t][q];
            L = (v > L || v != v) ? v : L;
          }
        L = tile_max_h(L);
        if constexpr (SPL) L = fmaxf(L, 1e-37f) * 1.002f;       // (in units of 2^-esg; half-precision operands: 2 x 2^-11)
        else L = fmaxf(L, 1e-30f * P.reg) * 1.0001f;
        if (!(L == L) || !(fabsf(L) < 1e30f)) { pflag |= MIA_FLAG_NONFINITE; L = SPL ? 1.0f : P.reg; }
        if constexpr (SPL) tab_idx = (int)ceilf(float(kTabPerOctave) * (__builtin_amdgcn_logf(L * P.inv_reg) + float(esg))) + kTabIdx0;
        else tab_idx = (int)ceilf(float(kTabPerOctave) * __builtin_amdgcn_logf(L * P.inv_reg)) + kTabIdx0;
        tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
        // header and coefficients are requested together: the coefficients of the 16 points go, scaled, to LDS --
        // [degree][point] pairs in the storage of the union scratch (D, hash table, slot tables: dead until the next
        // pass) -- for every degree that storage holds (the table is zero beyond an entry's degree): one memory round trip
        // in all, none per recurrence step.  (SPL: unscaled -- the universal coefficients are O(1), which keeps the
        // accumulated vectors in the range of the half-precision operands; the route's constants multiply the results)
        const int2 hd = ld_off<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
        {
          const unsigned cb = (unsigned)tab_idx * (unsigned)(kTabDeg * 8);
          const float sphi = SPL ? 1.0f : P.cs_phi, spsi = SPL ? 1.0f : P.cs_psi;
          float2 cst[CQ];
#pragma unroll
          for (int u = 0; u < CQ; ++u) cst[u] = ld_off<float2>(P.tab_c, cb + (unsigned)(hv + 4 * u < kTabDeg ? hv + 4 * u : kTabDeg - 1) * 8u);
#pragma unroll
          for (int u = 0; u < CQ; ++u)
            if (hv + 4 * u < CL) Cl[(hv + 4 * u) * 16 + lrv] = float2{cst[u].x * sphi, cst[u].y * spsi};
        }
        deg = hd.x;
        if constexpr (SPL) alpha = __builtin_ldexpf(__int_as_float(hd.y) * P.inv_reg, esg);
        else alpha = __int_as_float(hd.y) * P.inv_reg;
        decl = colact && (deg > P.dmax || deg > kTabDeg - 1);
        if (decl && hv == 0) {
          P.flags[p0v + lrv] = MIA_FLAG_RETRY;
          atomicAdd(P.retry_count, 1);
        }
        degmax = (int)wave_max_nonneg_dpp((colact && !decl) ? float(deg) : 0.0f);
        MIA_TILE_SYNC();
      }
      MIA_STAMP(6);
      // ---- the recurrence on the 16 columns at once; vectors stay in the result layout.
      //      Run on v with t = D o v (t_j = T_j(A) t_0, A = alpha D G D - I, t_0 = D o Z  <=>  v_0 = Z,
      //      v_{j+1} = 2 (alpha G (D^2 o v_j) - v_j) - v_{j-1}): one multiply per value and step less than on t, D enters as
      //      D^2 in the products' right-hand side and once at the end; slots that are not local to a column (D = 0) carry
      //      bounded junk that D^2 = 0 keeps out of every product.  Two steps per trip, so that the three-term update swaps
      //      roles instead of moving registers.
      //      SPL: the vectors are carried times a power of two per column (|v_0| -> 2^8; |v_j| <= sqrt(U) |v_0| stays far
      //      inside the half-precision range); every step splits the 4 UT values D^2 o v_j of the lane.
      f4t va[UT], vb[UT], aphi[UT], apsi[UT], y[UT], d2[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) d2[t] = dreg[t] * dreg[t];
      // the right-hand side D^2 o tv of a 32-deep product, slots 16 (2 kb + tt) + 4 h + q of this lane's column, as half pairs
      // (SPL carries u = D^2 o v instead of v: u_{j+1} = 2 (alpha D^2 o (G u_j) - u_j) - u_{j-1} -- the vectors ARE the right-hand
      //  sides, and the accumulated sums are what the last two products need)
      auto rhs_split = [&](const f4t (&tv)[UT], int kb, h8t& bh, h8t& bl) {
        float bv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int tk = 2 * kb + tt < UT ? 2 * kb + tt : 0;
            bv[4 * tt + q] = 2 * kb + tt < UT ? tv[tk][q] : 0.0f;
          }
        tile_split8(bv, bh, bl);
      };
      auto product = [&](const f4t (&tv)[UT]) {
#pragma unroll
        for (int t = 0; t < UT; ++t) y[t] = f4t{0.f, 0.f, 0.f, 0.f};
        if constexpr (SPL) {
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
            if (32 * kb < U) {
              h8t bh, bl;
              rhs_split(tv, kb, bh, bl);
#pragma unroll
              for (int t = 0; t < UT; ++t) y[t] = tile_mfma3(y[t], GAh[t][kb], GAl[t][kb], bh, bl);
            }
        } else {
#pragma unroll
          for (int tk = 0; tk < UT; ++tk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * tk + 4 * q < U) {
                const float b = d2[tk][q] * tv[tk][q];
#pragma unroll
                for (int t = 0; t < UT; ++t) y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(G[tk][t][q], b, y[t], 0, 0, 0);
              }
        }
      };
      auto coef = [&](int j) -> float2 {                              // (zero beyond a point's own degree)
        if (j < CL) return Cl[j * 16 + lrv];
        const float2 c = ld_off<float2>(P.tab_c, ((unsigned)tab_idx * (unsigned)kTabDeg + (unsigned)(j < kTabDeg ? j : kTabDeg - 1)) * 8u);
        return SPL ? c : float2{c.x * P.cs_phi, c.y * P.cs_psi};
      };
      // vnew = 2 (alpha y - vcur) - vold, written over vold; the two weight functions accumulate c_j vnew
      f4t ad2[UT];              // SPL: alpha D^2
      if constexpr (SPL) {
#pragma unroll
        for (int t = 0; t < UT; ++t) ad2[t] = alpha * d2[t];
      }
      auto advance = [&](f4t (&vold)[UT], const f4t (&vcur)[UT], const float2 cj) {
        product(vcur);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          if constexpr (SPL) vold[t] = 2.0f * (ad2[t] * y[t] - vcur[t]) - vold[t];
          else vold[t] = 2.0f * (alpha * y[t] - vcur[t]) - vold[t];
          aphi[t] = cj.x * vold[t] + aphi[t];
          apsi[t] = cj.y * vold[t] + apsi[t];
        }
      };
      float inv_s2 = 1.0f;       // SPL: 1 / (power of two the column's vectors are carried at, relative to Z)
      {
        const float2 c0 = coef(0), c1 = coef(1);
        if constexpr (SPL) {
          unsigned zmax = 0u;
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            va[t] = Z[t] * d2[t];                     // u_0
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned a = __float_as_uint(va[t][q]) & 0x7fffffffu;
              zmax = a > zmax ? a : zmax;
            }
          }
          zmax = __float_as_uint(tile_max_h(__uint_as_float(zmax)));
          int es2;
          const float s2 = tile_pow2_scale(zmax, 8, &es2);
          inv_s2 = __uint_as_float((unsigned)(127 - es2) << 23);
#pragma unroll
          for (int t = 0; t < UT; ++t) va[t] *= s2;
        } else {
#pragma unroll
          for (int t = 0; t < UT; ++t) va[t] = Z[t];
        }
        product(va);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          if constexpr (SPL) vb[t] = ad2[t] * y[t] - va[t];
          else vb[t] = alpha * y[t] - va[t];
          aphi[t] = c0.x * va[t] + c1.x * vb[t];
          apsi[t] = c0.y * va[t] + c1.y * vb[t];
        }
      }
      int j = 2;
      for (; j + 1 <= degmax; j += 2) {
        const float2 cj = coef(j), cj1 = coef(j + 1);
        advance(va, vb, cj);          // va = v_j
        advance(vb, va, cj1);         // vb = v_{j+1}
      }
      if (j <= degmax) advance(va, vb, coef(j));
      MIA_STAMP(7);
      // ---- x' w_mean = sum_b d_b (w_b psi_b): one more product, row vector of the innovations (column k of the records)
      //      times D o Psi -- on the matrix cores like everything else, because their enumeration IS the canonical
      //      summation order (a lane-local partial sum would group the observations by rank mod 4, i.e. by tile
      //      composition).  Row 0 of the result tile = lanes (lrv, hv = 0), register 0; handed to the column's other lanes.
      // (x of this row is needed once more, for f0 x': read again, L2-hot, instead of being held in 4 KT registers across
      //  the recurrence; requested before the last two products, which cover its latency)
      f4t xre[KT];
      load_x(mi, xre);
      f4t zacc = {0.f, 0.f, 0.f, 0.f};
      f4t acc[KT];
      float mterm;
      if constexpr (SPL) {
        // the results carry (scale of the records)^2 x (scale of x') x (scale of the vectors); the route's constants, left out of
        // the coefficients, come in here
        const float funs = (inv_s2 * inv_sx) * inv_sy * inv_sy;
        // a record value as a half pair from its chunk: row = slot, value index v (member, or k = the innovation)
        auto slot_of = [&](int kb, int i) -> unsigned { return (unsigned)(16 * (2 * kb + (i >> 2)) + 4 * hv + (i & 3)); };
        auto a_frag = [&](int kb, int v, bool on, h8t& ah, h8t& al) {
          const unsigned vo = (unsigned)(v >> 3) * 32u + (unsigned)(v & 7) * 2u;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool have = 2 * kb + (i >> 2) < UT;
            const _Float16* ph = reinterpret_cast<const _Float16*>(YwB + (have ? slot_of(kb, i) : 0u) * RSB + vo);
            ah[i] = ph[0];
            al[i] = ph[8];
          }
          if (!on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)0.0f; al[i] = (_Float16)0.0f; }
          }
        };
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
          if (32 * kb < U) {
            h8t ah, al, bh, bl;
            a_frag(kb, k, lrv == 0, ah, al);
            rhs_split(apsi, kb, bh, bl);
            zacc = tile_mfma3(zacc, ah, al, bh, bl);
          }
        const float zu = __shfl(zacc[0], lrv, 64) * (P.cs_psi * funs);
        mterm = xm + zu;
        h8t ph_[NKB], pl_[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) rhs_split(aphi, kb, ph_[kb], pl_[kb]);
        const float fo = P.cs_phi * funs;
        const int vlast = 8 * P.nc - 1;
#pragma unroll
        for (int tj = 0; tj < KT; ++tj) {
          acc[tj] = f4t{0.f, 0.f, 0.f, 0.f};
          const int mem = 16 * tj + lrv;                 // the output row this lane supplies to the A operand
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
            if (32 * kb < U) {
              h8t ah, al;
              a_frag(kb, mem < vlast ? mem : vlast, true, ah, al);
              acc[tj] = tile_mfma3(acc[tj], ah, al, ph_[kb], pl_[kb]);
            }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[tj][q] = acc[tj][q] * fo + (mterm + P.f0 * (xre[tj][q] - xm));
            if (!(fabsf(acc[tj][q]) <= 1e30f) && (tj < KT - 1 || 16 * tj + 4 * hv + q < k)) pflag |= MIA_FLAG_NONFINITE;
          }
        }
      } else {
#pragma unroll
        for (int tk = 0; tk < UT; ++tk)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (16 * tk + 4 * q < U) {
              const float a = lrv == 0 ? Yw[(16 * tk + 4 * hv + q) * kp + k] : 0.0f;
              zacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, d2[tk][q] * apsi[tk][q], zacc, 0, 0, 0);
            }
        const float zu = __shfl(zacc[0], lrv, 64);
#pragma unroll
        for (int t = 0; t < UT; ++t) aphi[t] *= d2[t];          // D o phi(S) z = D^2 o (accumulated v): right-hand side of the last product
        mterm = xm + zu;
#pragma unroll
        for (int tj = 0; tj < KT; ++tj) {
          acc[tj] = f4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tk = 0; tk < UT; ++tk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * tk + 4 * q < U)
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(Yw[(16 * tk + 4 * hv + q) * kp + 16 * tj + lrv], aphi[tk][q], acc[tj], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[tj][q] += mterm + P.f0 * (xre[tj][q] - xm);
            if (!(fabsf(acc[tj][q]) <= 1e30f) && (tj < KT - 1 || 16 * tj + 4 * hv + q < k)) pflag |= MIA_FLAG_NONFINITE;
          }
        }
      }
      if (colact && !decl) {
        float* obase = Xab + (int64_t)mi * k * P.ldo + oc0v;
        const unsigned olane = (unsigned)(4 * hv) * ldob + (unsigned)lrv * 4u;
#pragma unroll
        for (int tj = 0; tj < KT; ++tj)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (tj < KT - 1 || 16 * tj + 4 * hv + q < k) {
              float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(obase) + (olane + (unsigned)(16 * tj + q) * ldob));
              // segmented launch: write-through (agent-scope) stores, published by the counter below
              if constexpr (SEG) __hip_atomic_store(dst, acc[tj][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else *dst = acc[tj][q];
            }
      } else {
        pflag = 0;          // (columns that are not written do not report)
      }
    }
    MIA_STAMP(8);
    MIA_STAMP_REAL(11);
    {
      const unsigned long long fb = __ballot(pflag != 0);
      const bool anyf = ((fb >> lr) & 0x0001000100010001ull) != 0ull;
      if (h == 0 && colact && !decl) P.flags[p0v + lr] = (anyf ? MIA_FLAG_NONFINITE : 0) | (deg << 8);
    }
    lo = hi;
    MIA_TILE_SYNC();
    if (lo < npts) { load_lists(p0v, lp, sub); mask_lists(sub, lcnt); }       // a split tile: the entries of the next pass
  }
  if constexpr (SEG) {
    // all output stores of this wavefront have been acknowledged before it counts its points
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane0 == 0)
      __hip_atomic_fetch_add(P.done + ((size_t)sg * 64 + (unsigned)(bid & 63)) * kSlotStride, npts, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifdef MIA_TILE_STAMPS
#ifdef MIA_TILE_TU_SPLIT
extern "C" int mia_debug_tile_split_stamps(long long* host, int n_tiles) {
#else
extern "C" int mia_debug_tile_stamps(long long* host, int n_tiles) {
#endif
  if (n_tiles > kStampTiles) n_tiles = kStampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile_stamps), sizeof(long long) * kStampN * (size_t)n_tiles);
}
#endif

// rsb = 0: f32 records; otherwise the row stride of the split-precision layout
static size_t tile_lds_bytes(int ut, int kp, int rsb) {
  const int umax = 16 * ut, hs = ut <= 1 ? 64 : (ut <= 2 ? 128 : 256);
  const size_t yw = rsb ? (size_t)umax * rsb : ((size_t)umax * kp + 16) * sizeof(float);
  return align_up(yw + 16 * (size_t)(umax + 4) * sizeof(float) +
                  ((size_t)2 * hs + umax + 2 * (size_t)(umax > 64 ? umax : 64)) * sizeof(int), 16);
}

template <int UT, int KT, bool SEG, bool SPL>
static int tile_launch_s(const TileParams& tp, hipStream_t stream) {
  const size_t lds = tile_lds_bytes(UT, tp.kp, SPL ? tp.rsb : 0);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  TileParams tpl = tp;
  tpl.lds_bytes = (int)lds;
  auto kern = letkf_tile_kernel<UT, KT, SEG, SPL>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntile = SEG ? (((int64_t)tp.seg_len + 15) >> 4) * ((tp.ng + tp.seg_len - 1) / tp.seg_len) : (tp.ng + 15) >> 4;
  const int64_t gx = ntile < 65536 ? ntile : 65536;
  const int64_t gy = (ntile + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  hipEvent_t& stop = launch_stop_event();
  if (stop) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64), (unsigned)lds, stream, launch_start_event(), stop, 0,
                          tpl);
    stop = nullptr;        // taken
    launch_start_event() = nullptr;
  } else {
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(tpl);
  }
  ++tile_launch_count();
  note_analysis_kernel("letkf_tile_kernel<%d, %d, %s, %s>", UT, KT, SEG ? "true" : "false", SPL ? "true" : "false");
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Instantiations: UT <= KT + 1 (the dual route has p <= k, so a union of p + slack slots never needs more row blocks than
// that); the segmented variant (step driver with several pieces) up to UT = 4.
template <int UT, int KT, bool SPL>
static int tile_launch_t(const TileParams& tp, hipStream_t stream) {
  if constexpr (UT <= KT + 1) {
    if (tp.seg_len > 0) {
      if constexpr (UT <= 4 && KT <= 4) return tile_launch_s<UT, KT, true, SPL>(tp, stream);
      else return MIA_ERR_UNSUPPORTED;
    }
    return tile_launch_s<UT, KT, false, SPL>(tp, stream);
  } else {
    return MIA_ERR_UNSUPPORTED;
  }
}

template <int UT, bool SPL>
static int tile_launch_u(const TileParams& tp, int kt, hipStream_t stream) {
  switch (kt) {
    case 1: return tile_launch_t<UT, 1, SPL>(tp, stream);
    case 2: return tile_launch_t<UT, 2, SPL>(tp, stream);
    case 3: return tile_launch_t<UT, 3, SPL>(tp, stream);
    case 4: return tile_launch_t<UT, 4, SPL>(tp, stream);
    case 5: return tile_launch_t<UT, 5, SPL>(tp, stream);
    case 6: return tile_launch_t<UT, 6, SPL>(tp, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

// Slots an instantiation offers a tile beyond the longest single list.  Sixteen consecutive points of a regular
// network add ~one observation per second point (config 2: 20 -> 28); below this slack most tiles would be split.
constexpr int kTileSlack = 8;

static bool tile_shape_ok(int m, int k, int p_max) {
  return m >= 1 && k >= 2 && k <= 96 && p_max <= k && p_max + kTileSlack <= 96;
}

// What a launch needs of its sizes (shared with tile_launch_would_serve): the shape, every global access as base + 32-bit
// byte offset (the largest offsets are a column of one state row block -- k ld floats --, one tile's list rows), the LDS
// budget of the instantiation, the grid, the segmented instantiations (UT, KT <= 4)
static bool tile_launch_args_ok(int m, int k, int p_max, int p_cap, int64_t ldx, int64_t ldo, int64_t ng, int seg_len, bool spl) {
  if (!tile_shape_ok(m, k, p_max)) return false;
  if ((int64_t)k * ldx * 4 >= ((int64_t)1 << 31) || (int64_t)k * ldo * 4 >= ((int64_t)1 << 31) ||
      (int64_t)p_cap * 16 * 8 >= ((int64_t)1 << 31))
    return false;
  const int kt = (k + 15) >> 4, ut0 = (p_max + kTileSlack + 15) >> 4, ut = ut0 < 1 ? 1 : ut0;
  if (ut > kt + 1 || ut > 6 || kt > 6) return false;
  if (seg_len > 0 && (ut > 4 || kt > 4)) return false;
  const int kp = (k + 1 + 3) & ~3, rsb = 32 * ((k + 1 + 7) >> 3) + 16;
  if (tile_lds_bytes(ut, kp, spl ? rsb : 0) > kMaxDynamicLds) return false;
  const int64_t ntile = seg_len > 0 ? (((int64_t)seg_len + 15) >> 4) * ((ng + seg_len - 1) / seg_len) : (ng + 15) >> 4;
  return ntile <= (int64_t)65536 * 65535;
}

template <bool SPL>
static int tile_launch_impl(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                            const int32_t* nbr_cnt, const int32_t* nbr_idx, const void* nbr_w, int w_f32, int p_cap,
                            int p_max, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                            int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c, hipStream_t stream,
                            int seg_len, int64_t seg_stride, int32_t* done) {
  if (seg_len > 0 && (!done || ng >= (int64_t)1 << 31)) return MIA_ERR_UNSUPPORTED;
  if (!flags || !retry_count || !tab_hdr || !tab_c || w_f32) return MIA_ERR_UNSUPPORTED;
  if (!tile_launch_args_ok(m, k, p_max, p_cap, ldx, ldo, ng, seg_len, SPL)) return MIA_ERR_UNSUPPORTED;
  TileParams tp;
  tp.X = X; tp.ldx = ldx; tp.m = m; tp.k = k; tp.kp = (k + 1 + 3) & ~3;
  tp.g0 = g0; tp.ng = ng; tp.rec = rec;
  tp.cnt = nbr_cnt; tp.idx = nbr_idx; tp.w = nbr_w; tp.w_f32 = w_f32; tp.p_cap = p_cap; tp.p_max = p_max;
  const double rg = (double)(k - 1) / (double)inf_factor, km = (double)(k - 1);
  tp.reg = (float)rg;
  tp.inv_reg = (float)(1.0 / rg);
  tp.f0 = (float)sqrt(km / rg);
  tp.inv_k = (float)(1.0 / (double)k);
  tp.Xa = Xa; tp.ldo = ldo; tp.o0 = o0; tp.flags = flags; tp.retry_count = retry_count;
  tp.dmax = dmax;
  tp.tab_hdr = tab_hdr; tp.tab_c = tab_c;
  tp.cs_phi = (float)(sqrt(km) / (rg * sqrt(rg)));
  tp.cs_psi = (float)(1.0 / rg);
  tp.kpv_magic = ((1 << 20) + (tp.kp >> 2) - 1) / (tp.kp >> 2);
  tp.seg_len = seg_len; tp.seg_stride = seg_stride; tp.done = done;
  tp.nc = (k + 1 + 7) >> 3;                  // k members + the innovation, in chunks of eight
  tp.rsb = 32 * tp.nc + 16;
  const int kt = (k + 15) >> 4;
  const int ut = (p_max + kTileSlack + 15) >> 4;
  switch (ut < 1 ? 1 : ut) {
    case 1: return tile_launch_u<1, SPL>(tp, kt, stream);
    case 2: return tile_launch_u<2, SPL>(tp, kt, stream);
    case 3: return tile_launch_u<3, SPL>(tp, kt, stream);
    case 4: return tile_launch_u<4, SPL>(tp, kt, stream);
    case 5: return tile_launch_u<5, SPL>(tp, kt, stream);
    case 6: return tile_launch_u<6, SPL>(tp, kt, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

#ifdef MIA_TILE_TU_SPLIT
// (this translation unit = letkf_tile_split.hip: the split-precision instantiations, compiled beside the f32 ones)
int tile_split_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                               const int32_t* nbr_cnt, const int32_t* nbr_idx, const void* nbr_w, int w_f32, int p_cap,
                               int p_max, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                               int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c, hipStream_t stream,
                               int seg_len, int64_t seg_stride, int32_t* done) {
  return tile_launch_impl<true>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, w_f32, p_cap, p_max, inf_factor, Xa, ldo, o0,
                                flags, retry_count, dmax, tab_hdr, tab_c, stream, seg_len, seg_stride, done);
}
#else
// Any number of state rows: the Gram matrix, the union and the coefficients are shared by the rows of a tile, a further row
// costs Z + recurrence + output (C2, 1e5 points: 76 us for the first row, ~37 us per further row -- m = 8 / 16 / 32:
// 0.34 / 0.63 / 1.23 ms against 0.76 / 0.92 / 1.61 ms of the 16-row MFMA batches of letkf_cheb_rows_kernel, which remains the
// route of the shapes this kernel does not cover).
bool tile_route_covers(int m, int k, int p_max) { return tile_shape_ok(m, k, p_max); }

hipEvent_t& launch_stop_event() {
  static thread_local hipEvent_t ev = nullptr;
  return ev;
}
hipEvent_t& launch_start_event() {
  static thread_local hipEvent_t ev = nullptr;
  return ev;
}
unsigned long long& tile_launch_count() {
  static thread_local unsigned long long n = 0;
  return n;
}
// Two members: the split-precision variant of this kernel loses three digits when the analysis mean moves by hundreds of spreads
// (3e-4 against 4e-6 with f32 products, tools/small_k_sweep.py; three members and more: 2-3e-6 either way; the tile route's kernels
// are not affected) -- such ensembles take the f32 products.
static bool tile_split_for(int k) { return option(MIA_OPT_TILE_SPLIT) != 0 && k >= 3; }
bool tile_launch_would_serve(int m, int k, int p_max, int p_cap, int64_t ldx, int64_t ldo, int64_t ng, int seg_len) {
  return option(MIA_OPT_TILE) != 0 && tile_launch_args_ok(m, k, p_max, p_cap, ldx, ldo, ng, seg_len, tile_split_for(k));
}

int tile_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const void* nbr_w, int w_f32, int p_cap,
                         int p_max, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                         int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c, hipStream_t stream,
                         int seg_len, int64_t seg_stride, int32_t* done) {
  if (tile_split_for(k))
    return tile_split_analysis_launch(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, w_f32, p_cap, p_max, inf_factor, Xa, ldo,
                                      o0, flags, retry_count, dmax, tab_hdr, tab_c, stream, seg_len, seg_stride, done);
  return tile_launch_impl<false>(X, ldx, m, k, g0, ng, rec, nbr_cnt, nbr_idx, nbr_w, w_f32, p_cap, p_max, inf_factor, Xa, ldo, o0,
                                 flags, retry_count, dmax, tab_hdr, tab_c, stream, seg_len, seg_stride, done);
}
#endif

}  // namespace mia

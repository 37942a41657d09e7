// Tile lists + split-precision records: the data formats between the localisation kernel, the record packing kernel and
// the sixteen-points-per-wavefront analysis kernel of letkf_tile2.hip (round 3).
//
// Reference path: GaspariCohn.localize_obs per grid point (pytassim/localization/gaspari_cohn.py:97-136) feeding the
// mask + sqrt(rho) scaling of wrapper_localization (pytassim/interface/wrapper.py:86-98).  Round 2 wrote one neighbour
// list per grid point and let every analysis wavefront rebuild, for its 16 points, the UNION of their lists, the rank of
// every observation in it and the 16 x U matrix of sqrt(rho) -- a third of the analysis kernel's instructions and half of
// its latency chain.  Here the localisation kernel emits that tile-shaped form directly:
//
//   hdr [ntile]            int4   {U = slots used (-1: the union does not fit 16 UT slots), longest list of the tile, points, 0}
//   uidx[ntile][16 UT]     int32  observation index of slot s, -1 = unused.  Slots are numbered by RANK of the observation
//                                 index, permuted so that the 32-deep matrix products enumerate them in ascending rank:
//                                 slot(rk) = 16 (rk >> 4) + 4 (rk & 3) + ((rk >> 2) & 3)
//   D   [ntile][UT][64]    float4 lane (lr, h) = (lane & 15, lane >> 4), component q: sqrt(rho) of (point lr, slot
//                                 16 t + 4 h + q), 0 = not local -- the analysis wave's registers, one coalesced 1 KB read per t
//
// Split records (what `arg[..., use] * sqrt(w)` gathers, wrapper.py:94-97, prepared for half-precision matrix cores): an f32
// value x is carried as hi = f16(x), lo = f16(x - hi); record j is first scaled by its OWN power of two 2^e_j that brings its
// largest member magnitude to [2^9, 2^10) -- exact, and undone through the sqrt(rho) matrix (D_hat = D 2^-e), so records of
// very different magnitudes keep their 22-23 bits each:
//
//   rec[j] = nc8 chunks of {8 hi halves | 8 lo halves} (members 8c .. 8c+7, zero padded), then a 16-byte tail
//            {w = d_j 2^e_j (f32), E = 2^-e_j (f32; NaN = the record holds a non-finite value), 0, 0};
//   record P (one past the last) is all zeros: the source of unused slots.
#pragma once
#include "mia_common.h"

namespace mia {

// Slots an instantiation offers a tile beyond the longest single list.  Sixteen consecutive points of a regular
// network add ~one observation per second point (config 2: 20 -> 28); below this slack most tiles would not fit.
constexpr int kTileSlack = 8;

static inline int tile_ut_for(int p_max) {
  const int ut = (p_max + kTileSlack + 15) >> 4;
  return ut < 1 ? 1 : ut;
}

struct TileListLayout { size_t hdr, idx, D, bytes; int ut; int64_t ntile; };
static inline TileListLayout tile_list_layout(int64_t ng, int ut) {
  TileListLayout L;
  L.ut = ut;
  L.ntile = (ng + 15) >> 4;
  size_t o = 0;
  L.hdr = o; o = align_up(o + (size_t)(L.ntile > 0 ? L.ntile : 1) * 16, 256);
  L.idx = o; o = align_up(o + (size_t)(L.ntile > 0 ? L.ntile : 1) * 16 * ut * sizeof(int32_t), 256);
  L.D = o; o = align_up(o + (size_t)(L.ntile > 0 ? L.ntile : 1) * ut * 1024, 256);
  L.bytes = o;
  return L;
}

static inline int split_nc8(int k) { return (k + 7) >> 3; }
static inline int split_rec_bytes(int k) { return 32 * split_nc8(k) + 16; }

// power of two that brings a magnitude (given by its bit pattern, sign cleared) to [2^target, 2^(target+1)); the
// exponent of the scale is returned too.  Zero / subnormal magnitudes are left alone; exponents are clamped to +-60 (the
// squares and products of scales formed in the kernels then stay inside f32).
__device__ __forceinline__ float pow2_scale(unsigned magbits, int target, int* es_out) {
  const int e = (int)(magbits >> 23) - 127;
  int es = (magbits >> 23) == 0u ? 0 : target - e;
  es = es < -60 ? -60 : (es > 60 ? 60 : es);
  *es_out = es;
  return __uint_as_float((unsigned)(127 + es) << 23);
}

using h8v = __attribute__((ext_vector_type(8))) _Float16;
using h2v = __attribute__((ext_vector_type(2))) _Float16;
using f2w = __attribute__((ext_vector_type(2))) float;
using f4w = __attribute__((ext_vector_type(4))) float;
using u4w = __attribute__((ext_vector_type(4))) unsigned;

// x (8 values) -> hi = f16(x), lo = f16(x - hi), both rounded to nearest: three instructions per pair of values (the
// compiler's own form is two conversions, a packed subtraction and a conversion, and packed f32 instructions are slow beside
// MFMAs).
__device__ __forceinline__ void split8(const float (&x)[8], h8v& hi, h8v& lo) {
  u4w hu, lu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2w v = {x[2 * i], x[2 * i + 1]};
    const h2v a = __builtin_convertvector(v, h2v);                  // v_cvt_pk_f16_f32, round to nearest
    hu[i] = __builtin_bit_cast(unsigned, a);
  }
  // lo = f16(x - hi), straight into the two halves of a register: v_fma_mixlo / mixhi read the f16 operand in place and round
  // the (exact) f32 difference once.  ONE statement for the eight values, closed by the two wait states a matrix instruction
  // needs after a vector instruction wrote one of its operands: the compiler pads nothing around inline assembly (without
  // them an MFMA scheduled right behind read a stale operand -- found on the UT = 1 many-rows instantiation).
  // ... and OPENED by three: the outputs are fresh registers of the allocator's choice, and it likes the accumulator input (SrcC) of the
  // matrix instruction issued just before -- the compiler keeps three wait states between such a read and a vector write over it
  // in its own code (write-after-read), and pads nothing in front of inline assembly (tools/check_mfma_hazards.py: WAR rule).
  asm("s_nop 2\n\t"
      "v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "=&v"(lu[0]), "=&v"(lu[1]), "=&v"(lu[2]), "=&v"(lu[3])
      : "v"(hu[0]), "v"(hu[1]), "v"(hu[2]), "v"(hu[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
        "v"(x[6]), "v"(x[7]));
  hi = __builtin_bit_cast(h8v, hu);
  lo = __builtin_bit_cast(h8v, lu);
}
// The same with the lo halves written OVER A COPY of the hi halves (four register moves more): the outputs of split8 are fresh
// registers of the allocator's choice, and it may choose one that a matrix instruction wrote a few cycles earlier -- the
// compiler pads neither side of inline assembly, so the (later) write of the matrix instruction could land on top of the lo
// halves.  tools/check_mfma_hazards.py finds such places in a build; code outside the hottest loops uses this form.
__device__ __forceinline__ void split8_tied(const float (&x)[8], h8v& hi, h8v& lo) {
  u4w hu, lu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2w v = {x[2 * i], x[2 * i + 1]};
    const h2v a = __builtin_convertvector(v, h2v);
    hu[i] = __builtin_bit_cast(unsigned, a);
    lu[i] = hu[i];
  }
  asm("v_fma_mixlo_f16 %0, %0, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %1, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %2, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %2, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %3, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %3, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "+v"(lu[0]), "+v"(lu[1]), "+v"(lu[2]), "+v"(lu[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  hi = __builtin_bit_cast(h8v, hu);
  lo = __builtin_bit_cast(h8v, lu);
}
// The split of -x: the negation rides on source modifiers of the conversions and of the mixed multiply-adds alike (no instruction
// of its own: the compiler's conversion of a negated pair was a v_xor per value in front of it), all inside one statement.
// Its outputs are fresh registers of the allocator's choice and the compiler pads nothing around inline assembly: the leading
// s_nop 3 keeps the first write four wait states behind the instruction in front of it.  That covers a matrix instruction
// that still reads the register as its accumulator input (3), and a matrix instruction's RESULT register: the allocator hands
// it out once its last reader has issued -- a vector reader sits 8 wait states behind the result already, a matrix reader
// stands for 4 more itself (the compiler chains accumulators through different registers, so such registers do come free
// in the middle of a product).  tools/check_mfma_hazards.py checks every build.  12 vector instructions (round 4: 20; tied 24).
__device__ __forceinline__ void split8n(const float (&x)[8], h8v& hi, h8v& lo) {
  u4w hu, lu;
  asm("s_nop 3\n\t"
      "v_cvt_pk_f16_f32 %0, -%8, -%9\n\t"
      "v_cvt_pk_f16_f32 %1, -%10, -%11\n\t"
      "v_cvt_pk_f16_f32 %2, -%12, -%13\n\t"
      "v_cvt_pk_f16_f32 %3, -%14, -%15\n\t"
      "v_fma_mixlo_f16 %4, %0, -1.0, -%8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %4, %0, -1.0, -%9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %5, %1, -1.0, -%10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %5, %1, -1.0, -%11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %6, %2, -1.0, -%12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %6, %2, -1.0, -%13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %7, %3, -1.0, -%14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %7, %3, -1.0, -%15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "=&v"(hu[0]), "=&v"(hu[1]), "=&v"(hu[2]), "=&v"(hu[3]), "=&v"(lu[0]), "=&v"(lu[1]), "=&v"(lu[2]), "=&v"(lu[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  hi = __builtin_bit_cast(h8v, hu);
  lo = __builtin_bit_cast(h8v, lu);
}
__device__ __forceinline__ h8v hi8(const float (&x)[8]) {
  h8v hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2w v = {x[2 * i], x[2 * i + 1]};
    const h2v a = __builtin_convertvector(v, h2v);
    hi[2 * i] = a[0]; hi[2 * i + 1] = a[1];
  }
  return hi;
}

// ---- wave helpers of the tile kernels (letkf_tile2.hip, letkf_tile2w.hip) ----------------------------------------------
#define MIA_T2_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ float t2_add_h(float v) {       // sum over the four lanes (lr, h = 0..3), in every one of them
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ unsigned t2_max_h(unsigned u) {   // maximum of bit patterns over the same four lanes
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  u2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = r.x > r.y ? r.x : r.y;
  r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return r.x > r.y ? r.x : r.y;
}
// x' = x - mean of one state row as scaled half pairs (one power of two per column): the values of member chunk b of this lane
// (members 8 (4 b + sg) .. + 7 of column lr) arrive in xsb and leave as x' (unscaled); xm = the column's mean, inv_sx = 1 / scale.
// SHARED by letkf_tile2_kernel.h and letkf_tile2p.hip so that the one- and the two-wavefront kernels produce the same bits.
// Whole chunks of eight members (k a multiple of 8, wave-uniform test): a chunk beyond the ensemble, or a column beyond the tile, is
// switched off by a FACTOR 0 / 1 per chunk -- one multiply-add per value where round 4 had a compare and two selects per value (a
// select costs three multiply-adds' issue time, tools/micro/valu_rates.hip); the loads of such values are clamped to real data, so
// 0 x is 0.  Any other k: the per-value selects.
template <int NB, bool TIED>
__device__ __forceinline__ void t2_split_x(float (&xsb)[NB][8], const bool colok, const int sg, const int k, const float inv_k,
                                           float& xm, float& inv_sx, h8v (&xh)[NB], h8v (&xl)[NB]) {
#pragma clang fp contract(off)      // (inlined into several kernels that must agree bit for bit: no multiply-add the source does not spell out)
  float xs = 0.0f;
  unsigned xmax = 0u;
  if ((k & 7) == 0) {
    float lm[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) lm[b] = (colok && 8 * (4 * b + sg) < k) ? 1.0f : 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xs = __builtin_fmaf(xsb[b][i], lm[b], xs);
    xm = t2_add_h(xs) * inv_k;
    float xmaxf = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float xml = xm * lm[b];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xsb[b][i] = __builtin_fmaf(xsb[b][i], lm[b], -xml);
        xmaxf = __builtin_fmaxf(xmaxf, __builtin_fabsf(xsb[b][i]));
      }
    }
    xmax = __float_as_uint(xmaxf);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool live = colok && 8 * (4 * b + sg) + i < k;
        xsb[b][i] = live ? xsb[b][i] : 0.0f;
        xs += xsb[b][i];
      }
    xm = t2_add_h(xs) * inv_k;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool live = colok && 8 * (4 * b + sg) + i < k;
        xsb[b][i] = live ? xsb[b][i] - xm : 0.0f;
        const unsigned a = __float_as_uint(xsb[b][i]) & 0x7fffffffu;
        xmax = a > xmax ? a : xmax;
      }
  }
  xmax = t2_max_h(xmax);
  int esx;
  const float sx = pow2_scale(xmax, 9, &esx);
  inv_sx = __uint_as_float((unsigned)(127 - esx) << 23);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = xsb[b][i] * sx;
    // (tied where the split follows matrix instructions closely -- the row loop, the pair kernel: there the fresh outputs of the untied
    //  form could land on an accumulator still in flight, tools/check_mfma_hazards.py)
    if constexpr (TIED) split8_tied(t8, xh[b], xl[b]);
    else split8(t8, xh[b], xl[b]);
  }
}
__device__ __forceinline__ unsigned t2_wave_max_u32(unsigned u) {     // wave-uniform maximum (DPP)
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
__device__ __forceinline__ f4w t2_mfma3(f4w acc, const h8v ah, const h8v al, const h8v bh, const h8v bl) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ T t2_ld(const void* base, unsigned byte_off) {     // wave-uniform base + 32-bit lane offset
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ s4v t2_tr_read(const unsigned char* lds_addr) {     // ds_read_b64_tr_b16 (EXEC must be all ones)
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)lds_addr);
}


// split-record packing job (rides in the tile-list kernel as extra single-wave workgroups, or runs as a launch of its own)
struct SplitPackJob { const float* Yb; const float* d; unsigned char* rec; int k; };

// ---- split records ---------------------------------------------------------------------------------------------------
// One wavefront packs records j0 .. j0 + 63 (indices up to P: record P is the all-zero record).  Lane j reads entry j of
// every row of Yb (256-byte row segments, twenty-four rows requested before any is consumed) into an LDS image [64][ls], ls odd;
// every lane then finds its record's largest member magnitude (-> power of two), writes the tail, and the chunks of the
// 64 records are converted by all lanes, one chunk of eight members (32 bytes out) per lane and trip.
// lds: 64 * ls floats + 64 floats, ls = (k + 1) | 1.
__device__ inline void pack_split_wave(const SplitPackJob& J, int64_t P, int64_t block, float* lds) {
  const int lane = threadIdx.x & 63;
  const int k = J.k, nc8 = (k + 7) >> 3, rb = 32 * nc8 + 16;
  const int ls = (k + 1) | 1;
  float* scl = lds + 64 * ls;
  const int64_t j0 = block * 64;
  const int64_t j = j0 + lane;
  const bool real = j < P;                    // (j == P: the zero record; j > P: nothing)
  const int64_t jc = real ? j : (P > 0 ? P - 1 : 0);
  const float dj = (real && P > 0) ? J.d[jc] : 0.0f;                   // (requested with the first rows)
  constexpr int kRowsInFlight = 40;                                    // (k = 40: ONE round trip -- round 4: 24 rows, two trips)
  for (int i0 = 0; i0 < k; i0 += kRowsInFlight) {
    float v[kRowsInFlight];
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) {
      const int i = i0 + u < k ? i0 + u : k - 1;
      v[u] = (P > 0) ? J.Yb[(int64_t)i * P + jc] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u)
      if (i0 + u < k) lds[lane * ls + i0 + u] = real ? v[u] : 0.0f;
  }
  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0);
  unsigned mx = 0u;
  for (int i = 0; i < k; ++i) {
    const unsigned a = __float_as_uint(lds[lane * ls + i]) & 0x7fffffffu;
    mx = a > mx ? a : mx;
  }
  int es;
  const float sc = pow2_scale(mx, 9, &es);
  const float wd = dj * sc;
  const bool bad = mx >= 0x7f800000u || !(fabsf(wd) < 3.0e38f);
  scl[lane] = sc;
  if (j <= P) {
    const float E = bad ? __builtin_nanf("") : __uint_as_float((unsigned)(127 - es) << 23);
    *reinterpret_cast<f4w*>(J.rec + j * rb + 32 * nc8) = f4w{wd, E, 0.0f, 0.0f};
  }
  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0);
  const int nvalid = P + 1 - j0 < 64 ? (int)(P + 1 - j0) : 64;
  const int nq = nvalid * nc8;
  int r = lane / nc8, c = lane - r * nc8;
  const int dr = 64 / nc8, dc = 64 - dr * nc8;
  for (int q = lane; q < nq; q += 64) {
    const float s = scl[r];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 8 * c + i < k ? lds[r * ls + 8 * c + i] * s : 0.0f;
    h8v hi, lo;
    split8(x, hi, lo);
    unsigned char* o = J.rec + (j0 + r) * rb + 32 * c;
    *reinterpret_cast<h8v*>(o) = hi;
    *reinterpret_cast<h8v*>(o + 16) = lo;
    r += dr; c += dc;
    if (c >= nc8) { c -= nc8; ++r; }
  }
}

static inline size_t split_pack_lds(int k) { return ((size_t)64 * ((k + 1) | 1) + 64) * sizeof(float); }

// The same records from a wavefront that must fit BESIDE the analysis kernel's five 96-register wavefronts per SIMD (32 registers
// are left; the bucket kernel of the step driver, localize.hip): lane = record.  The 64 entries of every row of Yb go straight
// into LDS with global_load_lds_dword (no registers, all k rows in flight), image [k][64]; each lane then reads ITS column
// (conflict-free), finds its scale, and converts and stores its own record chunk by chunk.  Same arithmetic as pack_split_wave:
// the records are the same bit for bit.  lds: 64 k floats.  Two halves so that the caller can put work between request and use.
__device__ __forceinline__ void pack_split_lean_request(const SplitPackJob& J, int64_t P, int64_t block, float* lds) {
  const int lane = threadIdx.x & 63;
  const int64_t j = block * 64 + lane;
  const int64_t jc = j < P ? j : (P > 0 ? P - 1 : 0);
  if (P > 0)
    for (int i = 0; i < J.k; ++i)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(J.Yb + (int64_t)i * P + jc),
                                       (__attribute__((address_space(3))) void*)(lds + i * 64), 4, 0, 0);
}
__device__ __forceinline__ void pack_split_lean_finish(const SplitPackJob& J, int64_t P, int64_t block, const float* lds) {
  const int lane = threadIdx.x & 63;
  const int k = J.k, nc8 = (k + 7) >> 3, rb = 32 * nc8 + 16;
  const int64_t j = block * 64 + lane;
  const bool real = j < P;                    // (j == P: the zero record; j > P: nothing)
  const float dj = (real && P > 0) ? J.d[j] : 0.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  unsigned mx = 0u;
  if (real)
    for (int i = 0; i < k; ++i) {
      const unsigned a = __float_as_uint(lds[i * 64 + lane]) & 0x7fffffffu;
      mx = a > mx ? a : mx;
    }
  int es;
  const float sc = pow2_scale(mx, 9, &es);
  const float wd = dj * sc;
  const bool bad = mx >= 0x7f800000u || !(fabsf(wd) < 3.0e38f);
  if (j > P) return;
  unsigned char* o = J.rec + j * rb;
  const float E = bad ? __builtin_nanf("") : __uint_as_float((unsigned)(127 - es) << 23);
  *reinterpret_cast<f4w*>(o + 32 * nc8) = f4w{wd, E, 0.0f, 0.0f};
  for (int c = 0; c < nc8; ++c) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (real && 8 * c + i < k) ? lds[(8 * c + i) * 64 + lane] * sc : 0.0f;
    h8v hi, lo;
    split8(x, hi, lo);
    *reinterpret_cast<h8v*>(o + 32 * c) = hi;
    *reinterpret_cast<h8v*>(o + 32 * c + 16) = lo;
  }
}
static inline size_t split_pack_lean_lds(int k) { return (size_t)64 * k * sizeof(float); }

// host side (tile_lists.hip)
struct ScanParams;
int split_pack_launch(const float* Yb, const float* d, int k, int64_t P, void* rec, hipStream_t stream);
// tile lists of grid points [g0, g0 + ng) over the index already built in `index_ws` (bucket: by index_bucket_build_impl,
// whose error bits the kernel folds into stats[3] as bits 8 / 16; stats then has four entries); optional packing passenger.
// stats: [0] longest list (running maximum), [1] tiles whose union did not fit (added)
int tile_lists_launch(const double* grid_xyz, int64_t g0, int64_t ng, int64_t P, int n_coord, const int32_t* coord_group,
                      const double* gc_c, int n_r, double gc_eps, int taper, int ut, void* tile_lists, int32_t* stats,
                      void* index_ws, hipStream_t stream, const SplitPackJob* pack, bool bucket = false, const int* counts = nullptr);
// letkf_tile2.hip
// housekeeping the analysis launch does for the bucket index (see Tile2Params)
struct Tile2Params {
  const float* X; int64_t ldx; int m, k;
  int64_t g0, ng;
  const unsigned char* rec; int rb, nc8; int64_t zero_rec;
  const int4* thdr; const int32_t* tidx; const f4w* tD;
  float inv_reg, f0, inv_k, cs_phi, cs_psi;
  float* Xa; int64_t ldo, o0; int32_t* flags; int32_t* retry_count;
  int dmax;
  const int2* tab_hdr; const float2* tab_c;
  // pieces (step driver with an exchange in several pieces): the ng points are seg_len-sized pieces (a multiple of 16: tiles
  // never straddle one), piece s writes its own (m k, seg_len) buffer at Xa + s * seg_stride.  0: one result array
  int seg_len; int64_t seg_stride;
  // housekeeping for the step driver's bucket index (null: none): the launch's first workgroups zero the per-cell counts
  // clr_counts[0 .. *clr_n) -- their only readers, the tile-list kernel, ran before this launch -- and workgroup 0 folds the build's
  // error word into *err_out (bits 8, 16) and clears it
  int* clr_counts; const int* clr_n; unsigned* clr_err; int32_t* err_out;
  int stagger;      // (experiment builds) start delay per wave slot of a SIMD, in units of 64 cycles
};
struct Tile2Housekeeping { int* counts; const int* n; unsigned* err; int32_t* err_out; };
bool tile2_covers(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t ng);
bool tile2_records_addressable(int k, int64_t P);      // (P + 1) split records within 32-bit byte offsets
int tile2_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const void* rec, int64_t P,
                          const void* tile_lists, int ut, float inf_factor, float* Xa, int64_t ldo, int64_t o0,
                          int32_t* flags, int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c,
                          hipStream_t stream, int seg_len = 0, int64_t seg_stride = 0, const Tile2Housekeeping* hk = nullptr,
                          const struct Tile2Loc* loc = nullptr);
// letkf_tile2f.hip: the same analysis with the localisation of each tile done by its own wavefront (loc: mia_localize_dev.h) --
// shapes: unions of at most 32 slots, any number of state rows
bool tile2f_covers(int m, int k, int ut, int n_coord);
// launch coalescing (letkf_tile2f.hip): on the calling thread, tile2f_launch collects the steps' parameters instead of launching
// until tile2f_collect_launch puts them on `stream` as one grid (kT2fBatchMax steps at most; a step that does not fit the batch
// ends the collection and is launched on its own)
constexpr int kT2fBatchMax = 4;
void tile2f_collect_begin();
int tile2f_collected();
bool tile2f_collecting();
int tile2f_collect_launch(hipStream_t stream);
int tile2f_launch(const Tile2Params& tp, const struct Tile2Loc& loc, int ut, int kt, hipStream_t stream);
size_t tile2_lds_bytes(int ut, int k);

// letkf_tile2p.hip: the same analysis with two wavefronts per tile (unions of more than 32 slots); MIA_ERR_UNSUPPORTED for
// shapes it has no instantiation for
int tile2p_launch_any(const Tile2Params& tp, int ut, int kt, hipStream_t stream);
// lketkf_tile.hip: RBF-kernelised analysis from tile lists and the f32 perturbations themselves (no split records)
bool lketkf_tile_covers(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t ng, int64_t P);
int lketkf_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* Yb, const float* d,
                       int64_t P, const void* tile_lists, int ut, float inf_factor, float gamma, float* Xa, int64_t ldo,
                       int64_t o0, int32_t* flags, int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c,
                       hipStream_t stream, int seg_len = 0, int64_t seg_stride = 0, const Tile2Housekeeping* hk = nullptr);

}  // namespace mia

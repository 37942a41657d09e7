// letkf_tile2_kernel's body (see letkf_tile2.hip for the method and the LDS image), as a function template shared by the two
// translation units that instantiate it: letkf_tile2.hip (lists from memory) and letkf_tile2f.hip (the wavefront localises its
// own tile first).
#pragma once
#include <type_traits>
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"
#ifdef MIA_TILE_STAMPS
// (diagnostic builds: sub-phase stamps of the localisation, slots 12 .. 19 of the tile's row -- indexed by TILE there, by block here)
namespace mia { constexpr int kT2StampN = 20, kT2StampTiles = 8192; static __device__ long long g_tile2_stamps[kT2StampTiles * kT2StampN]; }
#define MIA_TL_STAMP(i) do { if (lane == 0 && tile < mia::kT2StampTiles) mia::g_tile2_stamps[tile * mia::kT2StampN + 12 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "mia_tile_localize.h"

namespace mia {

// In-kernel phase stamps (diagnostic builds only, tools/tile2_stamps.py): -DMIA_TILE_STAMPS compiles them in; the stamp values
// go to a buffer of their own that nothing else reads.  Slots 0 .. 8: s_memtime at the phase boundaries; 9: where the wave ran
// (HW_ID low word, XCC_ID high word); 10, 11: the constant 100 MHz counter at start and end (comparable across CUs).
#ifdef MIA_TILE_STAMPS
#define T2_STAMP(i) do { if (lane == 0 && bid < kT2StampTiles) g_tile2_stamps[bid * kT2StampN + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define T2_STAMP_HWID() do { if (lane == 0 && bid < kT2StampTiles) g_tile2_stamps[bid * kT2StampN + 9] = \
    (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } while (0)
#define T2_STAMP_REAL(i) do { if (lane == 0 && bid < kT2StampTiles) g_tile2_stamps[bid * kT2StampN + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T2_STAMP(i) do { } while (0)
#define T2_STAMP_HWID() do { } while (0)
#define T2_STAMP_REAL(i) do { } while (0)
#endif

#ifndef MIA_TILE2_WAVES_UT2
#define MIA_TILE2_WAVES_UT2 4
#endif

// MROWS = false: one state row per grid point (the benchmark configurations), straight-line code -- 86 registers at UT = 2,
// KT = 3, five wavefronts per SIMD.  MROWS = true: any number of rows in a loop that shares the union, the Gram matrix and the
// coefficients.  Left alone the compiler hoists the loop's invariant addresses and predicates in front of it (199-227 registers at
// the same shape, two wavefronts per SIMD); the loop launders the lane-derived indices at its top, so that work stays inside
// the row: 100 registers, four wavefronts per SIMD, 0.0196 instead of 0.0211 ms per row and 1e5 points.
//
// LOC = 0: lists from memory (localize_tiles_kernel ran before).  LOC = 1, 2, 3 (the number of coordinates): the wavefront
// LOCALISES ITS TILE ITSELF over the step's bucket index (tile_localize, mia_tile_localize.h -- the same code and therefore the same
// union, ranks and sqrt(rho) as the list kernel's) before anything else; its scratch shares the LDS of the record image, which is
// filled afterwards.  No tile list is written or read, one launch and one memory round trip less per step.
template <int UT, int KT, bool MROWS, int LOC>
__device__ __forceinline__ void tile2_body(Tile2Params P, const Tile2Loc* loc, const int64_t bid) {
  constexpr int UMAX = 16 * UT, NB = (KT + 1) / 2, NKB = (UT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int lane = threadIdx.x, lr = lane & 15, h = lane >> 4;      // (not const: the row loop launders them, see there)
  const int k = P.k, nc8 = P.nc8;
  const unsigned IMG = (unsigned)(UT * nc8) * 512u;
  unsigned char* zline = smem + IMG;                         // 512 zero bytes: the chunks a record does not have
  int* ukey = reinterpret_cast<int*>(smem + IMG + 512);      // [UMAX] observation index of a slot
  float* wdl = reinterpret_cast<float*>(ukey + UMAX);        // [UMAX] innovation of the slot's record, in its scale
  float* El = wdl + UMAX;                                    // [UMAX] 2^-e of the slot's record

  // XCD-aware block -> tile map: blocks b, b + 8, ... share an XCD (and its L2) and take consecutive tiles, whose
  // records overlap
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
  const unsigned ldxb = (unsigned)P.ldx * 4u, ldob = (unsigned)P.ldo * 4u;       // (k ld 4 < 2^31: checked on the host)
  int lrc = lr < npts ? lr : npts - 1;
  bool colok = lr < npts;

#ifdef MIA_EXPERIMENTS
  // (bits 8..: wave priority by phase -- 1: prologue high, compute normal; 2: prologue normal, compute high)
  const int prio_mode = (P.stagger >> 8) & 0xff, exp_trim = P.stagger >> 16;      // (bits 16..: steps taken off the table's degree)
  P.stagger &= 0xff;
  if (prio_mode == 1) __builtin_amdgcn_s_setprio(3);
  if (prio_mode >= 3) {     // fast lanes: the waves of some slots of a SIMD run at high priority, retire early and hand their slot (and
                            // its priority) to a tile of the second round while the other slots' waves are still at work
    const int wid = (int)(__builtin_amdgcn_s_getreg((4 << 11) | 4) & 0xf);      // HW_ID wave id
    const bool fast = prio_mode == 3 ? (wid & 1) : (prio_mode == 4 ? wid < 2 : (prio_mode == 5 ? wid == 0 : (wid % 3) == 0));
    if (fast) __builtin_amdgcn_s_setprio(3);
  }
  if (P.stagger > 0) {      // waves of one SIMD start their memory phases apart
    const int slot = (int)(__builtin_amdgcn_s_getreg((4 << 11) | 4) & 0xf);      // HW_ID wave id
    for (int i = 0; i < slot * P.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif
  T2_STAMP(0);
  T2_STAMP_HWID();
  T2_STAMP_REAL(10);
  // ---- first round trip: everything that does not depend on the slot table is requested together with it -- header, slot
  //      table (first: the only thing the second round trip waits for), sqrt(rho) matrix, the state row
  int hdU = 0;
  int myidx[(UMAX + 63) / 64];
  f4w dreg[UT];
  if constexpr (LOC == 0) {
    const int4 hd = P.thdr[tile];
    hdU = hd.x;
#pragma unroll
    for (int r = 0; r < (UMAX + 63) / 64; ++r) {
      const int s = lane + 64 * r;
      myidx[r] = s < UMAX ? t2_ld<int32_t>(P.tidx + tile * UMAX, (unsigned)s * 4u) : -1;
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) dreg[t] = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
  } else {
    // the tile's lists, formed here: union members / slot table / sqrt(rho) in LDS scratch -> this lane's slots and D fragments
    const TileLocOut lo = tile_localize<true, LOC, MIA_TAPER_GC>(loc->scan, P.g0, P.ng, UT, tile, smem, lane);
    const TileLocLds LL(smem);
    hdU = lo.overflow ? -1 : lo.U;
    // (two rounds of independent LDS reads, one wait each: slot -> member, then the member's key / weights -- an unused slot reads
    //  the scratch's key -1 / zero row instead of branching: round 4's select-by-branch form was a chain of sixteen dependent reads)
    int mu[(UMAX + 63) / 64], du[UT][4];
#pragma unroll
    for (int r = 0; r < (UMAX + 63) / 64; ++r) {
      const int s = lane + 64 * r;
      mu[r] = LL.uinv[s < UMAX ? s : 0];
      mu[r] = (s < UMAX && !lo.overflow && mu[r] >= 0) ? mu[r] : kTlUmax;          // ukey[kTlUmax] = -1
    }
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        du[t][qq] = LL.uinv[16 * t + 4 * h + qq];
        du[t][qq] = (!lo.overflow && du[t][qq] >= 0) ? du[t][qq] * 16 + lr : LL.zrow_off + lr;   // the zero row
      }
#pragma unroll
    for (int r = 0; r < (UMAX + 63) / 64; ++r) myidx[r] = LL.ukey[mu[r]];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      f4w v;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) v[qq] = LL.Wt[du[t][qq]];
      dreg[t] = v;
    }
    if (lane == 0) {
      // longest list: reported when it exceeds the bound the step was sized for (rare: the host then redoes the step), and by one
      // tile in 64 behind a guard load of the running maximum (see Tile2Loc::longest_bound)
      if (lo.longest > loc->longest_bound) atomicMax(&loc->stats[0], lo.longest);
      else if ((tile & 63) == 0 && lo.longest > __hip_atomic_load(&loc->stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&loc->stats[0], lo.longest);
      if (tile == 0) atomicOr(&loc->stats[3], kStepSampledLongest);
      if (lo.overflow) atomicAdd(&loc->stats[1], 1);
      if (lo.box_overflow) atomicOr(&loc->stats[1], MIA_TILE_BOX_OVERFLOW);
    }
    MIA_T2_SYNC();      // (every lane has read the scratch: the slot table and the record image may take its place)
  }
  // member (b, i) of lane group h = 8 sigma(b, h) + i, sigma = 4 b + 2 (h & 1) + (h >> 1): column lr of the state row
  int sg = 2 * (h & 1) + (h >> 1);
  auto load_xs = [&](int mi, float (&xr)[NB][8]) {
    const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int m0 = 8 * (4 * b + sg);
      if (b < NB - 1 || (k & 7) == 0) {
        const unsigned vo = (unsigned)(m0 < k ? m0 : 0) * ldxb + (unsigned)lrc * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[b][i] = t2_ld<float>(xbase, vo + (unsigned)i * ldxb);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int mem = m0 + i;
          xr[b][i] = t2_ld<float>(xbase, (unsigned)(mem < k ? mem : k - 1) * ldxb + (unsigned)lrc * 4u);
        }
      }
    }
  };
  float xsb[NB][8];
  load_xs(0, xsb);
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int s = lane + 64 * r;
    if (s < UMAX) ukey[s] = myidx[r];
  }
  for (int i = lane; i < 32; i += 64) reinterpret_cast<f4w*>(zline)[i] = f4w{0.f, 0.f, 0.f, 0.f};
  MIA_T2_SYNC();
  const int U = __builtin_amdgcn_readfirstlane(hdU);
  if (U < 0) {                     // the union of this tile did not fit its slots: loud failure, never a truncated analysis
    if (colok && h == 0) P.flags[p0 + lr] = MIA_FLAG_OVERFLOW;
    const float nanv = __builtin_nanf("");
    if (colok)
      for (int it = h; it < P.m * k; it += 4) P.Xa[(int64_t)it * P.ldo + oc0 + lr] = nanv;
    return;
  }
  T2_STAMP(1);        // header and slot table have arrived
  // ---- second round trip: the union's records, straight into the LDS image (load u, lane l = piece line 4 u + (l >> 4), column
  //      l & 15), and their tails (innovation, scale) -- consumed only after the Gram product
  {
    // Every lane's pieces belong to at most 2 UT records: row 16 t + ((lr - 8 par) & 15) of row block t for the chunks of parity
    // par.  Their byte offsets into the record array are formed ONCE (32-bit: (P + 1) rb < 2^32, checked on the host) -- round 4
    // looked the slot up in LDS and multiplied in 64 bits for every one of the five or six loads (25 vector instructions each).
    const int g = lane >> 4, hl = g & 1, gh = g >> 1;
    unsigned roff[UT][2];
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int idx = ukey[16 * t + ((lr - 8 * par) & 15)];
        roff[t][par] = (unsigned)(idx < 0 ? (int)P.zero_rec : idx) * (unsigned)P.rb + 16u * (unsigned)hl;
      }
    // A record has 2 KT - 1 or 2 KT chunks of eight members (k in (16 (KT - 1), 16 KT]): with the count a compile-time constant
    // the (row block, chunk) of every piece line is one too, and the lane's record offset is picked by INDEX -- as a run-time
    // count it was picked by eight selects per load and lane group, forty-odd selects per tile, each the issue time of three
    // multiply-adds (tools/micro/valu_rates.hip).  One wave-uniform branch instead.
    auto gather = [&](auto nc8c) {
      constexpr int NC8 = decltype(nc8c)::value;
      constexpr int NL = (UT * NC8 + 1) / 2;
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        // piece lines 2 u (lane groups 0, 1) and 2 u + 1 (groups 2, 3): (row block, chunk) of each
        const int lA = 2 * u, lB = 2 * u + 1;
        const int tA = lA / NC8, cA = lA % NC8, tB = (lB / NC8) < UT ? lB / NC8 : UT - 1, cB = lB % NC8;
        const unsigned offA = roff[tA][cA & 1] + 32u * (unsigned)cA, offB = roff[tB][cB & 1] + 32u * (unsigned)cB;
        const unsigned off = gh ? offB : offA;
        const bool valid = 2 * u + gh < UT * NC8;
        if (valid)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(P.rec + off),
                                           (__attribute__((address_space(3))) void*)(smem + u * 1024), 16, 0, 0);
      }
    };
    if (nc8 == 2 * KT - 1) gather(std::integral_constant<int, 2 * KT - 1>{});
    else gather(std::integral_constant<int, 2 * KT>{});
  }
  f2w tails[(UMAX + 63) / 64];
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const unsigned j = (unsigned)(myidx[r] < 0 ? (int)P.zero_rec : myidx[r]);
    tails[r] = *reinterpret_cast<const f2w*>(P.rec + (j * (unsigned)P.rb + 32u * (unsigned)nc8));
  }
  T2_STAMP(2);        // gather requested

  // byte offset of this lane's A / B fragment of row block t, member block b: row 16 t + lr, chunk sigma(b, h)
  auto frag_off = [&](int t, int b) -> unsigned {
    const int c = 4 * b + sg;
    const unsigned col = (unsigned)((lr + 8 * (c & 1)) & 15) * 16u;
    return c < nc8 ? (unsigned)(t * nc8 + c) * 512u + col : IMG + col;
  };

  h8v GAh[UT][NKB], GAl[UT][NKB];      // 2^-16 Ghat as A fragments of the 32-deep products
  float alpha = 0.0f;
  int deg = 0, tab_idx = 0, degmax = 0, pflag = 0;
  bool decl = false;
  // ---- x' = x - mean as scaled half pairs (one power of two per column)
  auto split_x = [&](float (&xsb)[NB][8], float& xm, float& inv_sx, h8v (&xh)[NB], h8v (&xl)[NB]) {
      t2_split_x<NB, MROWS>(xsb, colok, sg, k, P.inv_k, xm, inv_sx, xh, xl);
    };
  // ---- first state row: Gram matrix, interval and degree of every point (shared by all rows), its own Z
  float xm0, inv_sx0;
  f4w Z0[UT];
  {
    h8v xh[NB], xl[NB];
    split_x(xsb, xm0, inv_sx0, xh, xl);
#pragma unroll
    for (int t = 0; t < UT; ++t) Z0[t] = f4w{0.f, 0.f, 0.f, 0.f};
    f4w (&Z)[UT] = Z0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // records in LDS (x, D, tails in registers)
    __builtin_amdgcn_wave_barrier();
    T2_STAMP(3);        // ... and landed
#ifdef MIA_EXPERIMENTS
    if (prio_mode == 1) __builtin_amdgcn_s_setprio(0);
    if (prio_mode == 2) __builtin_amdgcn_s_setprio(3);
#endif
    f4w G[UT][UT];          // G[t1][t2][q] = Gram[16 t1 + 4 h + q][16 t2 + lr]
#pragma unroll
    for (int t1 = 0; t1 < UT; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < UT; ++t2) G[t1][t2] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      h8v ah[UT], al[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const unsigned o = frag_off(t, b);
        ah[t] = *reinterpret_cast<const h8v*>(smem + o);
        al[t] = *reinterpret_cast<const h8v*>(smem + o + 256);
      }
#pragma unroll
      for (int t2 = 0; t2 < UT; ++t2)
#pragma unroll
        for (int t1 = 0; t1 < UT; ++t1) G[t1][t2] = t2_mfma3(G[t1][t2], ah[t1], al[t1], ah[t2], al[t2]);
#pragma unroll
      for (int t = 0; t < UT; ++t) Z[t] = t2_mfma3(Z[t], ah[t], al[t], xh[b], xl[b]);
    }
    T2_STAMP(4);    // x' split, Gram + Z issued
    // the records' tails: innovation (in the record's scale) and scale per slot.  A tile with a non-finite record: through
    // the shared Gram matrix it would reach all 16 columns (NaN * 0 = NaN), also the points that do not see that observation --
    // every point of such a tile is handed to the eigensolver kernel (MIA_FLAG_RETRY), which works point by point and leaves
    // the damage where the reference has it.
    {
      bool badrec = false;
#pragma unroll
      for (int r = 0; r < (UMAX + 63) / 64; ++r) {
        const int s = lane + 64 * r;
        if (s < UMAX) {
          wdl[s] = tails[r][0];
          El[s] = tails[r][1];
          badrec = badrec || !(tails[r][1] == tails[r][1]);
        }
      }
      if (__any(badrec)) {
        if (colok && h == 0) { P.flags[p0 + lr] = MIA_FLAG_RETRY; atomicAdd(P.retry_count, 1); }
        return;
      }
      MIA_T2_SYNC();
      // D_hat = D E: the records' own scales enter through the sqrt(rho) matrix
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const f4w e4 = *reinterpret_cast<const f4w*>(El + 16 * t + 4 * h);
        dreg[t] *= e4;
      }
    }
    // A fragments of G for the 32-deep products: lane group h supplies slots 16 (2 kb + tt) + 4 h + q, i.e. the values
    // this lane holds of the tiles (2 kb, t) and (2 kb + 1, t) -- no data moves
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float gv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) gv[4 * tt + q] = 2 * kb + tt < UT ? G[2 * kb + tt < UT ? 2 * kb + tt : 0][t][q] * 0x1p-16f : 0.0f;
        split8(gv, GAh[t][kb], GAl[t][kb]);
      }
    // ---- Gershgorin bound of every point: L_g = max_a w_a sum_b |G_ab| w_b (hi halves only: a bound, margin below),
    //      then degree / interval from the table.  D_hat spans the records' scales: one power of two for the wave
    f4w R[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) R[t] = f4w{0.f, 0.f, 0.f, 0.f};
    unsigned dmx = 0u;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const unsigned a = __float_as_uint(dreg[t][q]); dmx = a > dmx ? a : dmx; }
    dmx = t2_wave_max_u32(dmx);
    int esd;
    const float sd = pow2_scale(dmx, 0, &esd);
    const float inv_sd = __uint_as_float((unsigned)(127 - esd) << 23);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      if (kb == 0 || 32 * kb < U) {     // (kb = 0 unconditionally: no branch between this product and the use of its result)
        float dv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) dv[4 * tt + q] = 2 * kb + tt < UT ? dreg[2 * kb + tt < UT ? 2 * kb + tt : 0][q] * sd : 0.0f;
        const h8v dh = hi8(dv);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          u4w ag = __builtin_bit_cast(u4w, GAh[t][kb]);
          ag &= 0x7fff7fffu;
          R[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, ag), dh, R[t], 0, 0, 0);
        }
      }
    // (a plain maximum: a non-finite record was sent to the eigensolver kernel above, so nothing here is NaN; round 4's NaN-keeping
    //  form was two comparisons and a select per value)
    float L = 0.0f;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) L = __builtin_fmaxf(L, dreg[t][q] * R[t][q]);
    L = __uint_as_float(t2_max_h(__float_as_uint(L))) * inv_sd;
    L = fmaxf(L, 1e-37f) * 1.002f;       // (in units of 2^-16; half-precision operands: 2 x 2^-11)
    if (!(L == L) || !(fabsf(L) < 1e30f)) { pflag |= MIA_FLAG_NONFINITE; L = 1.0f; }
    tab_idx = (int)ceilf(float(kTabPerOctave) * (__builtin_amdgcn_logf(L * P.inv_reg) + 16.0f)) + kTabIdx0;
    tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
    const int2 th = t2_ld<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
    deg = th.x;
#ifdef MIA_EXPERIMENTS
    deg = deg - exp_trim > 3 ? deg - exp_trim : (deg < 3 ? deg : 3);
#endif
    alpha = __builtin_ldexpf(__int_as_float(th.y) * P.inv_reg, 16);
    decl = colok && (deg > P.dmax || deg > kTabDeg - 1);
    if (decl && h == 0) {
      P.flags[p0 + lr] = MIA_FLAG_RETRY;
      atomicAdd(P.retry_count, 1);
    }
    degmax = (int)wave_max_nonneg_dpp((colok && !decl) ? float(deg) : 0.0f);
  }
  // ---- per state row: recurrence, x' w_mean, output product, stores
  auto row_tail = [&](const int mi, f4w (&Z)[UT], const float xm, const float inv_sx) {
    // ---- the recurrence on the 16 columns at once, on u = D^2 o v (the vectors ARE the right-hand sides of the products):
    //      u_{j+1} = 2 (alpha D^2 o (G u_j) - u_j) - u_{j-1}, u_0 = D^2 o Z; the two weight functions accumulate c_j u_j.
    //      Vectors are carried times a power of two per column (|u_0| -> 2^8; |u_j| <= sqrt(U) |u_0| stays far inside the
    //      half-precision range); the universal coefficients are used unscaled, the route's constants multiply the results.
    T2_STAMP(5);      // Gershgorin, table header requested and used
    const unsigned cbase = (unsigned)tab_idx * (unsigned)(kTabDeg * 8);
    auto coef = [&](int j) -> float2 {                              // (zero beyond a point's own degree)
      return t2_ld<float2>(P.tab_c, cbase + (unsigned)(j < kTabDeg ? j : kTabDeg - 1) * 8u);
    };
    const float2 c0 = coef(0), c1 = coef(1);
    float2 cn0 = coef(2), cn1 = coef(3);
    f4w va[UT], vb[UT], aphi[UT], apsi[UT], ad2[UT];
    float inv_s2;
    {
      float zmaxf = 0.0f;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const f4w d2 = dreg[t] * dreg[t];
        ad2[t] = alpha * d2;
        va[t] = Z[t] * d2;                     // u_0
#pragma unroll
        for (int q = 0; q < 4; ++q) zmaxf = __builtin_fmaxf(zmaxf, __builtin_fabsf(va[t][q]));
      }
      const unsigned zmax = t2_max_h(__float_as_uint(zmaxf));
      int es2;
      const float s2 = pow2_scale(zmax, 8, &es2);
      inv_s2 = __uint_as_float((unsigned)(127 - es2) << 23);
#pragma unroll
      for (int t = 0; t < UT; ++t) va[t] *= s2;
    }
    auto rhs_split = [&](const f4w (&tv)[UT], int kb, h8v& bh, h8v& bl) {
      float bv[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tk = 2 * kb + tt < UT ? 2 * kb + tt : 0;
          bv[4 * tt + q] = 2 * kb + tt < UT ? tv[tk][q] : 0.0f;
        }
      split8(bv, bh, bl);
    };
    // (kb = 0 runs unconditionally -- a tile without any observation left above -- so that the accumulators start from the
    //  MFMA's zero operand instead of eight register moves per step)
    f4w y[UT];
    auto product = [&](const f4w (&tv)[UT]) {
      {
        h8v bh, bl;
        rhs_split(tv, 0, bh, bl);
#pragma unroll
        for (int t = 0; t < UT; ++t) y[t] = t2_mfma3(f4w{0.f, 0.f, 0.f, 0.f}, GAh[t][0], GAl[t][0], bh, bl);
      }
#pragma unroll
      for (int kb = 1; kb < NKB; ++kb)
        if (32 * kb < U) {
          h8v bh, bl;
          rhs_split(tv, kb, bh, bl);
#pragma unroll
          for (int t = 0; t < UT; ++t) y[t] = t2_mfma3(y[t], GAh[t][kb], GAl[t][kb], bh, bl);
        }
    };
    // u_new = 2 (alpha D^2 o y - u_cur) - u_old, written over u_old; the two weight functions accumulate c_j u_new.  Scalar
    // fused multiply-adds on purpose (this file is compiled without the SLP vectoriser): v_pk_fma_f32 costs more than two
    // v_fma_f32 beside MFMAs, and the packed form needs separate negations
    auto advance = [&](f4w (&vold)[UT], const f4w (&vcur)[UT], const float2 cj) {
      product(vcur);
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tq = __builtin_fmaf(ad2[t][q], y[t][q], -vcur[t][q]);
          const float vn = __builtin_fmaf(2.0f, tq, -vold[t][q]);
          vold[t][q] = vn;
          aphi[t][q] = __builtin_fmaf(cj.x, vn, aphi[t][q]);
          apsi[t][q] = __builtin_fmaf(cj.y, vn, apsi[t][q]);
        }
    };
    product(va);
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vq = __builtin_fmaf(ad2[t][q], y[t][q], -va[t][q]);
        vb[t][q] = vq;
        aphi[t][q] = __builtin_fmaf(c1.x, vq, c0.x * va[t][q]);
        apsi[t][q] = __builtin_fmaf(c1.y, vq, c0.y * va[t][q]);
      }
    int j = 2;
    for (; j + 1 <= degmax; j += 2) {
      const float2 cj = cn0, cj1 = cn1;
      cn0 = coef(j + 2); cn1 = coef(j + 3);       // (two steps ahead: the loads' latency hides behind the products)
      advance(va, vb, cj);          // va = u_j
      advance(vb, va, cj1);         // vb = u_{j+1}
    }
    if (j <= degmax) advance(va, vb, cn0);
    // ---- output.  With y_b = E_b yhat_b (true record = its scale times the normalised record), D_hat = D E and the
    //      recurrence run on u_hat = E u:   x' w_mean = sum_b d_b (D psi(S) z)_b = sum_b (d_b / E_b) apsi_hat_b, and
    //      Xa' = sum_b y_b (D phi(S) z)_b = sum_b yhat_b aphi_hat_b: no per-slot factor is left.  The results carry
    //      (scale of x') x (scale of the vectors); the route's constants, left out of the coefficients, come in here.
    T2_STAMP(6);      // recurrence
    const float funs = inv_s2 * inv_sx;
    // (x of this row once more, in the RESULT layout -- member 16 tj + 4 h + q -- for f0 x': L2-hot, requested before the
    //  last products, which cover its latency)
    f4w xre[KT];
    {
      const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0;
      const unsigned xo0 = (unsigned)(4 * h) * ldxb + (unsigned)lrc * 4u;            // member 4 h, this lane's column
      const unsigned xolast = (unsigned)(k - 1) * ldxb + (unsigned)lrc * 4u;       // (clamp for the ragged last block)
#pragma unroll
      for (int tj = 0; tj < KT; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned o = xo0 + (unsigned)(16 * tj + q) * ldxb;
          if (tj == KT - 1) o = o < xolast ? o : xolast;
          xre[tj][q] = t2_ld<float>(xbase, o);
        }
    }
    float zu = 0.0f;
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const f4w w4 = *reinterpret_cast<const f4w*>(wdl + 16 * t + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) zu = fmaf(w4[q], apsi[t][q], zu);
    }
    zu = t2_add_h(zu) * (P.cs_psi * funs);
    const float mterm = xm + zu;
    h8v ph_[NKB], pl_[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) rhs_split(aphi, kb, ph_[kb], pl_[kb]);
    const float fo = P.cs_phi * funs;
    // A operand of the output product: rows = members 16 tj + lr, summation over slots -- the records are stored member-
    // contiguous, so this is a transposed read: lane 4 q + p of group h addresses slot 16 (2 kb + tt) + 4 h + q, members
    // 16 tj + 4 p .. + 3 (chunk 2 tj + (p >> 1), byte 8 (p & 1)) and receives, for its member, the four slots q = 0 .. 3
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    f4w acc[KT];
    unsigned amax = 0u;        // largest |value| of this lane's results as a bit pattern: NaN > inf > every finite value
#pragma unroll
    for (int tj = 0; tj < KT; ++tj) {
      acc[tj] = f4w{0.f, 0.f, 0.f, 0.f};
      const int c = 2 * tj + (tp >> 1);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (32 * kb < U) {
          s4v a4[2][2];       // [tt][hi / lo]
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int tb = 2 * kb + tt < UT ? 2 * kb + tt : 0;
            const unsigned col = (unsigned)((4 * h + tq + 8 * (c & 1)) & 15) * 16u + 8u * (unsigned)(tp & 1);
            const unsigned o = (c < nc8 && 2 * kb + tt < UT) ? (unsigned)(tb * nc8 + c) * 512u + col : IMG + col;
            a4[tt][0] = t2_tr_read(smem + o);
            a4[tt][1] = t2_tr_read(smem + o + 256);
          }
          typedef short s8v __attribute__((__vector_size__(8 * sizeof(short))));
          const s8v ahs = __builtin_shufflevector(a4[0][0], a4[1][0], 0, 1, 2, 3, 4, 5, 6, 7);
          const s8v als = __builtin_shufflevector(a4[0][1], a4[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[tj] = t2_mfma3(acc[tj], __builtin_bit_cast(h8v, ahs), __builtin_bit_cast(h8v, als), ph_[kb], pl_[kb]);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[tj][q] = acc[tj][q] * fo + (mterm + P.f0 * (xre[tj][q] - xm));
        // (one integer maximum per value and ONE comparison per row instead of a comparison per value -- twelve comparisons were the
        //  issue time of forty multiply-adds; rows beyond the ensemble hold the clamped member k - 1: the same magnitudes)
        const unsigned ab = __float_as_uint(acc[tj][q]) & 0x7fffffffu;
        amax = ab > amax ? ab : amax;
      }
    }
    if (amax > 0x7149f2cau) pflag |= MIA_FLAG_NONFINITE;       // |value| > 1e30, infinite or NaN
    if (colok && !decl) {
      float* obase = P.Xa + (int64_t)mi * k * P.ldo + oc0;
      const unsigned olane = (unsigned)(4 * h) * ldob + (unsigned)lr * 4u;
#pragma unroll
      for (int tj = 0; tj < KT; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (tj < KT - 1 || 16 * tj + 4 * h + q < k)
            *reinterpret_cast<float*>(reinterpret_cast<char*>(obase) + (olane + (unsigned)(16 * tj + q) * ldob)) = acc[tj][q];
    } else {
      pflag = 0;          // (columns that are not written do not report)
    }
  };
  row_tail(0, Z0, xm0, inv_sx0);
  if constexpr (MROWS) {
    for (int mi = 1; mi < P.m; ++mi) {
      // Everything a row needs beyond its own values -- fragment offsets, output addresses, predicates -- derives from the lane
      // number; left alone the compiler computes all of it once in front of the loop and carries it through (227 registers,
      // two wavefronts per SIMD).  Laundering the lane-derived indices here makes that work part of each row: a few dozen
      // integer instructions per row against half the registers.
      asm volatile("" : "+v"(lane), "+v"(lr), "+v"(h));
      lrc = lr < npts ? lr : npts - 1;
      colok = lr < npts;
      sg = 2 * (h & 1) + (h >> 1);
      load_xs(mi, xsb);
      float xm, inv_sx;
      h8v xh[NB], xl[NB];
      split_x(xsb, xm, inv_sx, xh, xl);
      f4w Z[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) Z[t] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          const unsigned o = frag_off(t, b);
          const h8v ah = *reinterpret_cast<const h8v*>(smem + o), al = *reinterpret_cast<const h8v*>(smem + o + 256);
          Z[t] = t2_mfma3(Z[t], ah, al, xh[b], xl[b]);
        }
      }
      row_tail(mi, Z, xm, inv_sx);
    }
  }
  T2_STAMP(7);        // output products and stores issued
  {
    const unsigned long long fb = __ballot(pflag != 0);
    const bool anyf = ((fb >> lr) & 0x0001000100010001ull) != 0ull;
    if (h == 0 && colok && !decl) P.flags[p0 + lr] = (anyf ? MIA_FLAG_NONFINITE : 0) | (deg << 8);
    if constexpr (LOC > 0) {      // (the step's status word says so too: a host that trusts it need not scan 1e5 flags per step)
      if (fb != 0ull && lane == 0) atomicOr(&loc->stats[3], kStepNonfinite);
    }
  }
  T2_STAMP(8);
  T2_STAMP_REAL(11);
}


}  // namespace mia

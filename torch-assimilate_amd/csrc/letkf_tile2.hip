// Fused LETKF analysis, sixteen grid points per wavefront, FROM TILE LISTS AND SPLIT RECORDS (round 3; formats: mia_tiles.h).
//
// Same mathematics as letkf_tile.hip (reference: core/etkf.py:57-103 + interface/wrapper.py:86-98 + base.py:257-278; matrix
// functions of the local matrix S_g = D_g G D_g applied by a Chebyshev recurrence, G = Yw Yw^T shared by the 16 points of
// a tile, every contraction a v_mfma_f32_16x16x32_f16 triple on operands carried as pairs of halves), but the wavefront no
// longer builds anything: the union of the tile's lists, the rank order of its observations and the sqrt(rho) matrix come
// from localize_tiles_kernel in the layout of this kernel's registers, and the records arrive already scaled and split
// into halves.  What is left of the prologue is three memory round trips (tile header -> records -> coefficient table):
//
//   uidx, D, x   requested together; the records of the union go STRAIGHT INTO LDS (global_load_lds_dwordx4, no registers)
//   Gram + Z     36 MFMAs (config 2) on ds_read_b128 fragments
//   Gershgorin   interval + degree per point from the table; coefficients are read per step, two steps ahead
//   recurrence   u_{j+1} = 2 (alpha D^2 o (G u_j) - u_j) - u_{j-1} on the 16 columns at once, vectors in the result layout
//   output       x' w_mean on the vector unit, Xa' = Yw^T (D o Phi) with transposed LDS reads (ds_read_b64_tr_b16)
//
// Per-record scaling.  Record j is normalised by its own power of two (y_j = E_j yhat_j, E_j = 2^-e_j): with
// D_hat = D E the local matrix is S = D_hat Ghat D_hat exactly, the recurrence runs on u_hat = E u, and both final
// products come out in true units without any per-slot correction (see the derivation at the output phase).  Observations of
// very different magnitudes inside one tile therefore keep their 22-23 bits each (round 2 scaled a tile's records by ONE
// power of two).
//
// LDS image of the union's records (conflict-free for both kinds of read): 16-byte piece (row r, chunk c, hi / lo) lives at
//   ((r >> 4) nc8 + c) * 512 + hl * 256 + ((r + 8 (c & 1)) & 15) * 16
// i.e. the sixteen rows of a row block side by side in one 256-byte bank line per (chunk, half); odd chunks are rotated by
// eight rows so that a transposed read (two chunks of a row at once) touches every bank once, and the member blocks of the
// products take chunks in the order sigma(b, h) = 4 b + 2 (h & 1) + (h >> 1) so that the lanes of one ds_read_b128 group
// (h = 0 / 1, or 2 / 3) read chunks of equal parity.  The image is lane-linear for the LDS-DMA: lane l of load u writes
// bytes [1024 u + 16 l, +16), and fetches the piece that belongs there.
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"

namespace mia {


// In-kernel phase stamps (diagnostic builds only, tools/tile2_stamps.py): -DMIA_TILE_STAMPS compiles them in; the stamp values
// go to a buffer of their own that nothing else reads.  Slots 0 .. 8: s_memtime at the phase boundaries; 9: where the wave ran
// (HW_ID low word, XCC_ID high word); 10, 11: the constant 100 MHz counter at start and end (comparable across CUs).
#ifdef MIA_TILE_STAMPS
constexpr int kT2StampN = 12, kT2StampTiles = 8192;
__device__ long long g_tile2_stamps[kT2StampTiles * kT2StampN];
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

// MROWS = false: one state row per grid point (the benchmark configurations), straight-line code -- 102 registers at UT = 2,
// KT = 3, four wavefronts per SIMD.  MROWS = true: any number of rows in a loop that shares the union, the Gram matrix and the
// coefficients; the compiler hoists the loop's invariant addresses and predicates in front of it (227 registers at the same
// shape), so these instantiations run at two wavefronts per SIMD.
template <int UT, int KT, bool MROWS, int WAVES>
__global__ __launch_bounds__(64, WAVES)
void letkf_tile2_kernel(Tile2Params P) {
  constexpr int UMAX = 16 * UT, NB = (KT + 1) / 2, NKB = (UT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x, lr = lane & 15, h = lane >> 4;
  const int k = P.k, nc8 = P.nc8;
  const unsigned IMG = (unsigned)(UT * nc8) * 512u;
  unsigned char* zline = smem + IMG;                         // 512 zero bytes: the chunks a record does not have
  int* ukey = reinterpret_cast<int*>(smem + IMG + 512);      // [UMAX] observation index of a slot
  float* wdl = reinterpret_cast<float*>(ukey + UMAX);        // [UMAX] innovation of the slot's record, in its scale
  float* El = wdl + UMAX;                                    // [UMAX] 2^-e of the slot's record

  // XCD-aware block -> tile map: blocks b, b + 8, ... share an XCD (and its L2) and take consecutive tiles, whose
  // records overlap
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
  const unsigned ldxb = (unsigned)P.ldx * 4u, ldob = (unsigned)P.ldo * 4u;       // (k ld 4 < 2^31: checked on the host)
  const int lrc = lr < npts ? lr : npts - 1;
  const bool colok = lr < npts;

#ifdef MIA_EXPERIMENTS
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
  const int4 hd = P.thdr[tile];
  int myidx[(UMAX + 63) / 64];
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int s = lane + 64 * r;
    myidx[r] = s < UMAX ? t2_ld<int32_t>(P.tidx + tile * UMAX, (unsigned)s * 4u) : -1;
  }
  f4w dreg[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) dreg[t] = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
  // member (b, i) of lane group h = 8 sigma(b, h) + i, sigma = 4 b + 2 (h & 1) + (h >> 1): column lr of the state row
  const int sg = 2 * (h & 1) + (h >> 1);
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
  const int U = __builtin_amdgcn_readfirstlane(hd.x);
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
    const int g = lane >> 4, hl = g & 1;
    int tc = g >> 1;                                     // (row block, chunk) index of this lane's piece line, load 0
    constexpr int NLmax = (UT * 2 * KT + 1) / 2;
#pragma unroll
    for (int u = 0; u < NLmax; ++u) {
      if (2 * u < UT * nc8) {                            // (wave-uniform)
        int t = 0, c = tc;
#pragma unroll
        for (int i = 1; i < UT; ++i)
          if (c >= nc8) { c -= nc8; ++t; }
        const bool valid = tc < UT * nc8;
        const int r = 16 * t + ((lr - 8 * (c & 1)) & 15);
        const int idx = valid ? ukey[r] : -1;
        const int64_t j = idx < 0 ? P.zero_rec : (int64_t)idx;
        const unsigned char* src = P.rec + j * P.rb + (32 * c + 16 * hl);
        if (valid)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src),
                                           (__attribute__((address_space(3))) void*)(smem + u * 1024), 16, 0, 0);
      }
      tc += 2;
    }
  }
  f2w tails[(UMAX + 63) / 64];
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int64_t j = myidx[r] < 0 ? P.zero_rec : (int64_t)myidx[r];
    tails[r] = *reinterpret_cast<const f2w*>(P.rec + j * P.rb + 32 * nc8);
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
      float xs = 0.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool live = colok && 8 * (4 * b + sg) + i < k;
          xsb[b][i] = live ? xsb[b][i] : 0.0f;
          xs += xsb[b][i];
        }
      xm = t2_add_h(xs) * P.inv_k;
      unsigned xmax = 0u;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool live = colok && 8 * (4 * b + sg) + i < k;
          xsb[b][i] = live ? xsb[b][i] - xm : 0.0f;
          const unsigned a = __float_as_uint(xsb[b][i]) & 0x7fffffffu;
          xmax = a > xmax ? a : xmax;
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
        split8(t8, xh[b], xl[b]);
      }
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
    float L = 0.0f;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = dreg[t][q] * R[t][q];
        L = (v > L || v != v) ? v : L;
      }
    L = __uint_as_float(t2_max_h(__float_as_uint(L))) * inv_sd;
    L = fmaxf(L, 1e-37f) * 1.002f;       // (in units of 2^-16; half-precision operands: 2 x 2^-11)
    if (!(L == L) || !(fabsf(L) < 1e30f)) { pflag |= MIA_FLAG_NONFINITE; L = 1.0f; }
    tab_idx = (int)ceilf(float(kTabPerOctave) * (__builtin_amdgcn_logf(L * P.inv_reg) + 16.0f)) + kTabIdx0;
    tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
    const int2 th = t2_ld<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
    deg = th.x;
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
      unsigned zmax = 0u;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const f4w d2 = dreg[t] * dreg[t];
        ad2[t] = alpha * d2;
        va[t] = Z[t] * d2;                     // u_0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned a = __float_as_uint(va[t][q]) & 0x7fffffffu;
          zmax = a > zmax ? a : zmax;
        }
      }
      zmax = t2_max_h(zmax);
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
        if (!(fabsf(acc[tj][q]) <= 1e30f) && (tj < KT - 1 || 16 * tj + 4 * h + q < k)) pflag |= MIA_FLAG_NONFINITE;
      }
    }
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
  }
  T2_STAMP(8);
  T2_STAMP_REAL(11);
}

#ifdef MIA_TILE_STAMPS
extern "C" int mia_debug_tile2_stamps(long long* host, int n_tiles) {
  if (n_tiles > kT2StampTiles) n_tiles = kT2StampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile2_stamps), sizeof(long long) * kT2StampN * (size_t)n_tiles);
}
#endif

static size_t tile2_lds_bytes(int ut, int k) {
  return (size_t)ut * split_nc8(k) * 512 + 512 + (size_t)16 * ut * 12;
}

// wavefronts per SIMD the instantiations are compiled for (register budget 512 / WAVES)
template <int UT, int KT, bool MROWS>
constexpr int tile2_waves() { return UT <= 2 && KT <= 4 ? (MROWS ? 2 : MIA_TILE2_WAVES_UT2) : (UT <= 3 && KT <= 4 && !MROWS ? 2 : 1); }

template <int UT, int KT, bool MROWS, int WAVES = tile2_waves<UT, KT, MROWS>()>
static int tile2_launch_m(const Tile2Params& tp, hipStream_t stream) {
  size_t lds = tile2_lds_bytes(UT, tp.k);
  {   // (experiment builds: a larger LDS request caps the wavefronts per CU -- occupancy A/B without touching the registers)
    int per_cu = 0;
    MIA_EXP_SET(per_cu, "MIA_TILE2_WAVES_PER_CU", atoi);
    if (per_cu > 0 && (size_t)(160 * 1024) / per_cu > lds) lds = ((size_t)(160 * 1024) / per_cu) & ~(size_t)255;
  }
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  auto kern = letkf_tile2_kernel<UT, KT, MROWS, WAVES>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntile = (tp.ng + 15) >> 4;
  const int64_t gx = ntile < 65536 ? ntile : 65536;
  const int64_t gy = (ntile + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  hipEvent_t& stop = launch_stop_event();
  if (stop) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64), (unsigned)lds, stream, launch_start_event(), stop, 0, tp);
    stop = nullptr;        // taken
    launch_start_event() = nullptr;
  } else {
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(tp);
  }
  ++tile_launch_count();
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <int UT, int KT>
static int tile2_launch_s(const Tile2Params& tp, hipStream_t stream) {
#ifdef MIA_EXPERIMENTS
  if constexpr (UT == 2 && KT == 3) {       // (A/B of the occupancy target, tools/ builds only: MIA_TILE2_WAVES=3|5|6)
    int w = 0;
    MIA_EXP_SET(w, "MIA_TILE2_WAVES", atoi);
    if (tp.m == 1 && w == 5) return tile2_launch_m<UT, KT, false, 5>(tp, stream);
    if (tp.m == 1 && w == 6) return tile2_launch_m<UT, KT, false, 6>(tp, stream);
    if (tp.m == 1 && w == 3) return tile2_launch_m<UT, KT, false, 3>(tp, stream);
  }
#endif
  bool force_rows = false;      // (experiment builds: the many-rows instantiation also for m = 1)
  MIA_EXP_SET(force_rows, "MIA_TILE2_FORCE_MROWS", (bool)atoi);
  return tp.m == 1 && !force_rows ? tile2_launch_m<UT, KT, false>(tp, stream) : tile2_launch_m<UT, KT, true>(tp, stream);
}

template <int UT>
static int tile2_launch_u(const Tile2Params& tp, int kt, hipStream_t stream) {
  switch (kt) {
    case 1: if constexpr (UT <= 2) return tile2_launch_s<UT, 1>(tp, stream); else break;
    case 2: if constexpr (UT <= 3) return tile2_launch_s<UT, 2>(tp, stream); else break;
    case 3: if constexpr (UT <= 4) return tile2_launch_s<UT, 3>(tp, stream); else break;
    case 4: if constexpr (UT <= 5) return tile2_launch_s<UT, 4>(tp, stream); else break;
    case 5: return tile2_launch_s<UT, 5>(tp, stream);
    case 6: return tile2_launch_s<UT, 6>(tp, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

// dual route (p_max <= k), k <= 96, union of at most 96 slots (16 per row block: tile_ut_for(p_max) + extra_blocks of them, at most
// one more than the ensemble has member blocks), every global access as base + 32-bit byte offset
bool tile2_covers(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t ng) {
  if (!(m >= 1 && k >= 2 && k <= 96 && p_max <= k && extra_blocks >= 0)) return false;
  if ((int64_t)k * ldx * 4 >= ((int64_t)1 << 31) || (int64_t)k * ldo * 4 >= ((int64_t)1 << 31)) return false;
  const int kt = (k + 15) >> 4, ut = tile_ut_for(p_max) + extra_blocks;
  if (ut > kt + 1 || ut > 6) return false;
  if (tile2_lds_bytes(ut, k) > kMaxDynamicLds) return false;
  return ((ng + 15) >> 4) <= (int64_t)65536 * 65535;
}

int tile2_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const void* rec, int64_t P,
                          const void* tile_lists, int ut, float inf_factor, float* Xa, int64_t ldo, int64_t o0,
                          int32_t* flags, int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c,
                          hipStream_t stream, int seg_len, int64_t seg_stride, const Tile2Housekeeping* hk) {
  if (seg_len < 0 || (seg_len & 15) || (seg_len > 0 && ng >= ((int64_t)1 << 31))) return MIA_ERR_UNSUPPORTED;
  if (!flags || !retry_count || !tab_hdr || !tab_c || !tile_lists || !rec) return MIA_ERR_UNSUPPORTED;
  const int kt = (k + 15) >> 4;
  if (ut < 1 || ut > 6 || ut > kt + 1 || !tile2_covers(m, k, 0, ut - 1, ldx, ldo, ng)) return MIA_ERR_UNSUPPORTED;
  const TileListLayout L = tile_list_layout(ng, ut);
  const char* base = (const char*)tile_lists;
  Tile2Params tp;
  tp.X = X; tp.ldx = ldx; tp.m = m; tp.k = k; tp.g0 = g0; tp.ng = ng;
  tp.rec = (const unsigned char*)rec; tp.rb = split_rec_bytes(k); tp.nc8 = split_nc8(k); tp.zero_rec = P;
  tp.thdr = (const int4*)(base + L.hdr); tp.tidx = (const int32_t*)(base + L.idx); tp.tD = (const f4w*)(base + L.D);
  const double rg = (double)(k - 1) / (double)inf_factor, km = (double)(k - 1);
  tp.inv_reg = (float)(1.0 / rg);
  tp.f0 = (float)sqrt(km / rg);
  tp.inv_k = (float)(1.0 / (double)k);
  tp.cs_phi = (float)(sqrt(km) / (rg * sqrt(rg)));
  tp.cs_psi = (float)(1.0 / rg);
  tp.Xa = Xa; tp.ldo = ldo; tp.o0 = o0; tp.flags = flags; tp.retry_count = retry_count; tp.dmax = dmax;
  tp.tab_hdr = tab_hdr; tp.tab_c = tab_c;
  tp.seg_len = seg_len; tp.seg_stride = seg_stride;
  tp.clr_counts = hk ? hk->counts : nullptr; tp.clr_n = hk ? hk->n : nullptr; tp.clr_err = hk ? hk->err : nullptr;
  tp.err_out = hk ? hk->err_out : nullptr;
  tp.stagger = 0;
  MIA_EXP_SET(tp.stagger, "MIA_TILE2_STAGGER", atoi);
  // unions of more than 32 slots: two wavefronts per tile (letkf_tile2p.hip) where it has the shape
  if (ut >= 3 && option(MIA_OPT_TILE_PAIR) != 0) {
    const int prc = tile2p_launch_any(tp, ut, kt, stream);
    if (prc != MIA_ERR_UNSUPPORTED) return prc;
  }
#ifdef MIA_TILE2_SINGLE        // (development builds: one instantiation, for register / ISA inspection)
  if (ut == 2 && kt == 3) return tile2_launch_s<2, 3>(tp, stream);
  return MIA_ERR_UNSUPPORTED;
#else
  switch (ut) {
    case 1: return tile2_launch_u<1>(tp, kt, stream);
    case 2: return tile2_launch_u<2>(tp, kt, stream);
    case 3: return tile2_launch_u<3>(tp, kt, stream);
    case 4: return tile2_launch_u<4>(tp, kt, stream);
    case 5: return tile2_launch_u<5>(tp, kt, stream);
    case 6: return tile2_launch_u<6>(tp, kt, stream);
  }
  return MIA_ERR_UNSUPPORTED;
#endif
}

}  // namespace mia

using namespace mia;

extern "C" int mia_letkf_analysis_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                            const void* split_rec, int64_t P, const void* tile_lists, int p_max,
                                            int extra_blocks, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                                            int32_t* retry_count, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 1 || k < 2 || g1 < g0 || g0 < 0 || P < 0 || p_max < 0 || extra_blocks < 0 || !(inf_factor > 0.0f)) return MIA_ERR_SIZE;
  if (g1 == g0) return MIA_OK;
  if (!X || !Xa || !split_rec || !tile_lists || !flags || !retry_count) return MIA_ERR_NULL;
  if (!tile2_covers(m, k, p_max, extra_blocks, ldx, ldo, g1 - g0)) return MIA_ERR_UNSUPPORTED;
  const int2* th = nullptr;
  const float2* tc = nullptr;
  if (!cheb_dual_table(stream, &th, &tc)) return MIA_ERR_UNSUPPORTED;
  return tile2_analysis_launch(X, ldx, m, k, g0, g1 - g0, split_rec, P, tile_lists, tile_ut_for(p_max) + extra_blocks, inf_factor, Xa,
                               ldo, o0, flags, retry_count, option(MIA_OPT_CHEB_DMAX), th, tc, stream, 0, 0, nullptr);
}

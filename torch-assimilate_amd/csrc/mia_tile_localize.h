// The localisation of ONE tile of sixteen consecutive grid points by ONE wavefront, shared by localize_tiles_kernel (tile_lists.hip:
// the lists go to memory) and the analysis kernel's fused variant (letkf_tile2_kernel.h: the lists never leave the wavefront).
// Stands in for 16 evaluations of GaspariCohn.localize_obs (pytassim/localization/gaspari_cohn.py:97-136) + the mask / sqrt(rho)
// of wrapper_localization (pytassim/interface/wrapper.py:88-97); see tile_lists.hip for the method.
#pragma once
#include "mia_common.h"
#include "mia_localize_dev.h"
#include "mia_tiles.h"

namespace mia {

#ifndef MIA_TL_STAMP
#define MIA_TL_STAMP(i) do { } while (0)
#endif
#define MIA_TL_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- tile lists ------------------------------------------------------------------------------------------------------
// Gaspari-Cohn taper in float32 from the float64 squared distance -- the weights enter a float32 analysis, so float32
// accuracy is what they need; what must stay the reference's is the DECISION `w > eps`, and that is re-taken in float64 for the
// (rare) pairs whose float32 weight lies within 1e-4 of eps (localize_tiles_kernel).  Written so that nothing cancels:
//   outer branch  f2(r) = (2 - r)^4 (r^2 / 12 + r / 6 - 1 / 24) / r      (the polynomial of gaspari_cohn.py:87-95 has a fourth-
//   order zero at r = 2; its Horner form loses all digits there in float32), and 2 - r = (4 c^2 - d^2) / (c^2 (2 + r)) with the
//   numerator formed in float64 -- relative error of the weight ~6e-7 up to the edge of the support.
// Floating-point contraction is OFF in here and in the distance loop of tile_localize, the fused multiply-adds are written out:
// the function is inlined into two kernels (the list kernel and the analysis kernel's fused variant) and both must produce the
// same bits -- left to the compiler, 2e3 of 4e6 analysis values differed in their last place (1e-9 relative) between the two.
// Round 5: straight-line, ONE comparison.  Comparisons and selects are the expensive vector instructions here (tools/micro/valu_rates.hip:
// a compare + select pair costs about six multiply-adds' issue time with several waves on the SIMD, a transcendental 3.5, a float64
// operation 2): r comes from v_sqrt_f32 (sqrt(0) = 0: no special case for a zero distance; a NaN distance stays NaN and ends as
// weight NaN, which the caller's `w > eps` rejects like the reference's comparisons do), the outer branch's zero beyond r = 2 from
// clamping 2 - r at zero (v_max_f32) instead of a second comparison, and the compiler keeps no branch in it.
__device__ __forceinline__ float gc_taper_fast(double d2, double four_c2, float inv_c, float c, float inv_c2) {
#pragma clang fp contract(off)
  const float d2f = (float)d2;
  const float r = __builtin_amdgcn_sqrtf(d2f) * inv_c;
  const float rinv = c * __builtin_amdgcn_rsqf(d2f);             // (enters where r >= 1 only; inf at r = 0 is never selected)
  const float h1 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.25f, r, 0.5f), r, 0.625f), r, -5.0f / 3.0f);
  const float f1 = __builtin_fmaf(h1 * r, r, 1.0f);
  float t = (float)(four_c2 - d2) * inv_c2 * __builtin_amdgcn_rcpf(2.0f + r);      // 2 - r
  t = __builtin_fmaxf(t, 0.0f);                                  // r >= 2: zero (strict `<` at 2, gaspari_cohn.py:127-133)
  const float t2 = t * t;
  const float f2 = t2 * t2 * __builtin_fmaf(__builtin_fmaf(r, 1.0f / 12.0f, 1.0f / 6.0f), r, -1.0f / 24.0f) * rinv;
  return r < 1.0f ? f1 : f2;
}

// ONE coordinate: the distance is |dx| itself -- no square, no square root -- and 2 - r = (2 c - |dx|) / c is formed in float64 where
// it cancels and scaled in float32: no reciprocal either (two transcendental instructions per pair instead of four, each the issue
// time of three and a half multiply-adds; two float64 operations instead of three).
__device__ __forceinline__ float gc_taper_fast_1d(double dx, double two_c, float inv_c, float c) {
#pragma clang fp contract(off)
  const float a = __builtin_fabsf((float)dx);
  const float r = a * inv_c;
  const float rinv = c * __builtin_amdgcn_rcpf(a);                // (enters where r >= 1 only; inf at r = 0 is never selected)
  const float h1 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.25f, r, 0.5f), r, 0.625f), r, -5.0f / 3.0f);
  const float f1 = __builtin_fmaf(h1 * r, r, 1.0f);
  float t = (float)(two_c - __builtin_fabs(dx)) * inv_c;          // 2 - r
  t = __builtin_fmaxf(t, 0.0f);                                   // r >= 2: zero (strict `<` at 2, gaspari_cohn.py:127-133)
  const float t2 = t * t;
  const float f2 = t2 * t2 * __builtin_fmaf(__builtin_fmaf(r, 1.0f / 12.0f, 1.0f / 6.0f), r, -1.0f / 24.0f) * rinv;
  return r < 1.0f ? f1 : f2;
}

constexpr int kTlUmax = 96;           // largest union (UT = 6)
constexpr int kTlMaxRows = 64;        // cell rows of a tile's box (outer coordinates); more = scattered points: no tile list

// LDS of one localising wavefront: coordinates and cells of the tile's points, union members, slot table, sqrt(rho) matrix
struct TileLocLds {
  double* gxs;      // [16][3] grid coordinates of the tile's points
  int* cgs;         // [16][3] their cells
  int* ukey;        // [kTlUmax] observation index of union member u
  int* uinv;        // [kTlUmax] member of slot s, -1 = unused
  double* cox;      // [64][3] coordinates of the candidates of one pass (bucket index: gathered once, lane = candidate)
  int* coj;         // [64] their observation indices
  float* Wt;        // [16 ut][16] sqrt(rho) of (member, point), 0 = not local; Wt[zrow_off .. + 15] (in FRONT of it) = zeros
  static constexpr int zrow_off = -16;
  __device__ __forceinline__ explicit TileLocLds(unsigned char* base) {
    gxs = reinterpret_cast<double*>(base);
    cox = gxs + 16 * MIA_MAX_COORD;
    cgs = reinterpret_cast<int*>(cox + 64 * MIA_MAX_COORD);
    ukey = cgs + 16 * MIA_MAX_COORD;                 // [kTlUmax + 16]: ukey[kTlUmax] = -1 (the key of an unused slot)
    uinv = ukey + kTlUmax + 16;
    coj = uinv + kTlUmax;
    Wt = reinterpret_cast<float*>(coj + 64) + 16;
  }
};
static inline size_t tile_loc_lds(int ut) {
  return (16 + 64) * MIA_MAX_COORD * sizeof(double) + 16 * MIA_MAX_COORD * sizeof(int) + (2 * kTlUmax + 16 + 64) * sizeof(int) +
         (size_t)16 * (ut * 16 + 1) * sizeof(float);
}

struct TileLocOut {
  int U;               // members of the union (may exceed 16 ut: then `overflow`)
  int longest;         // longest list among the tile's points
  int npts;
  bool overflow;       // the union does not fit the slots, or the tile's box is too large: no list
  bool box_overflow;   // the tile's points span more cells than the kernel scans (scattered orderings, coarse cells)
};

// BUCKET: the observations sit in fixed-capacity buckets per cell (index_bucket_kernel, localize.hip) instead of the scan-based
// layout: scan.start = per-cell counts, scan.sorted / scan.sxyz = bucket entries, cell c at c * bucket_cap.  The candidates of a
// tile are then the entries of the cells of its box taken as ONE flat sequence (a prefix sum over the box's <= 64 cells).  (The
// per-cell counts and the build's error word are put back to zero by the analysis kernel that follows -- letkf_tile2_kernel's
// first workgroups -- so the workspace is left as the next build needs to find it, without a clearing launch and without a
// completion counter: 6250 atomics on one address cost 75 us.)
// NC = number of coordinates, TAPER = MIA_TAPER_*: compile-time, so that the distance loops unroll and the taper is one
// straight-line polynomial (the generic form spent more scalar instructions on its loops than vector ones on the weights).
template <bool BUCKET, int NC, int TAPER>
__device__ __forceinline__ TileLocOut tile_localize(const ScanParams& q, int64_t g0, int64_t ng, int ut, int64_t tile,
                                                    unsigned char* tl_lds, int lane) {
  const TileLocLds lds_(tl_lds);
  double* gxs = lds_.gxs; int* cgs = lds_.cgs; int* ukey = lds_.ukey; int* uinv = lds_.uinv; float* Wt = lds_.Wt;
  double* cox = lds_.cox; int* coj = lds_.coj;
  const int cl = lane & 15, pg = lane >> 4;
  const int UMAX = 16 * ut;
  const IndexHeader* hd = q.hdr;
  constexpr int nc = NC;
  const int64_t p0 = tile << 4;
  const int npts = ng - p0 < 16 ? (int)(ng - p0) : 16;
  MIA_TL_STAMP(0);
  if (lane < 16) {
    const int64_t pt = g0 + p0 + (lane < npts ? lane : npts - 1);
    for (int c = 0; c < MIA_MAX_COORD; ++c) {
      double gx = 0.0;
      int cg = 0;
      if (c < nc) {
        gx = q.grid[pt * nc + c];
        cg = cell_coord(gx, hd->mn[c], hd->invh[c], hd->n[c]);
      }
      gxs[lane * MIA_MAX_COORD + c] = gx;
      cgs[lane * MIA_MAX_COORD + c] = cg;
    }
  }
  for (int s = lane; s < kTlUmax; s += 64) uinv[s] = -1;
  if (lane < 16) { Wt[TileLocLds::zrow_off + lane] = 0.0f; ukey[kTlUmax + lane] = -1; }
  MIA_TL_SYNC();
  MIA_TL_STAMP(1);
  // the tile's cell box: [min cell - 1, max cell + 1] per coordinate, clipped to the cell grid
  int lo[MIA_MAX_COORD], hi[MIA_MAX_COORD];
  for (int c = 0; c < MIA_MAX_COORD; ++c) {
    // (lane cl holds point cl's cell -- points past the tile's end repeat its last one, see above -- and the sixteen lanes of a row
    //  reduce with four DPP steps each way; the loop over the points in LDS this replaces was a sixth of the kernel's instructions)
    int mn = cgs[cl * MIA_MAX_COORD + c], mx = mn;
#define MIA_TL_ROW_STEP(ctrl) do { const int a_ = __builtin_amdgcn_update_dpp(0, mn, ctrl, 0xf, 0xf, false), b_ = __builtin_amdgcn_update_dpp(0, mx, ctrl, 0xf, 0xf, false); \
                                   mn = a_ < mn ? a_ : mn; mx = b_ > mx ? b_ : mx; } while (0)
    MIA_TL_ROW_STEP(0xB1); MIA_TL_ROW_STEP(0x4E); MIA_TL_ROW_STEP(0x124); MIA_TL_ROW_STEP(0x128);      // quad_perm x 2, row_ror:4, row_ror:8
#undef MIA_TL_ROW_STEP
    lo[c] = mn - 1 < 0 ? 0 : mn - 1;
    hi[c] = mx + 1 > hd->n[c] - 1 ? hd->n[c] - 1 : mx + 1;
    if (c >= nc) { lo[c] = 0; hi[c] = 0; }
    lo[c] = __builtin_amdgcn_readfirstlane(lo[c]);       // (the same in every lane: scalar loop bounds below)
    hi[c] = __builtin_amdgcn_readfirstlane(hi[c]);
  }
  const int last = nc - 1;
  bool empty = false;
  for (int c = 0; c < nc; ++c) empty = empty || lo[c] > hi[c];
  // outer coordinates (all but the last, whose cells are contiguous in the index): rows of the box
  const int n0 = nc >= 2 ? hi[0] - lo[0] + 1 : 1;
  const int n1 = nc == 3 ? hi[1] - lo[1] + 1 : 1;
  const long long nrows = empty ? 0 : (long long)n0 * n1;
  bool overflow = nrows > kTlMaxRows;
  bool box_overflow = overflow;        // the tile's points span more cells than the kernel scans (scattered orderings, coarse cells)
  int ubase = 0;
  // sixteen candidates (one per lane cl, position `pos` of the index arrays, valid where `have`) against the tile's points
  // (the coordinates of this lane's four points -- point group pg -- and the radius groups, in registers)
  double gxr[4][NC];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < NC; ++c) gxr[i][c] = gxs[(4 * pg + i) * MIA_MAX_COORD + c];
  int grp[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) grp[c] = NC == 1 ? 0 : q.group[c];
  const int n_r = NC == 1 ? 1 : q.n_r;
  // (float32 copies of the taper's constants, float64 4 c^2)
  const float epsf = q.eps_f;
  double fc2[MIA_MAX_RADII];
  float icf[MIA_MAX_RADII], ccf[MIA_MAX_RADII], ic2f[MIA_MAX_RADII];
#pragma unroll
  for (int r = 0; r < MIA_MAX_RADII; ++r) { fc2[r] = q.four_c2[r]; icf[r] = q.inv_c_f[r]; ccf[r] = q.c_f[r]; ic2f[r] = q.inv_c2_f[r]; }
  const double twoc0 = 2.0 * q.cc[0];
  // a candidate: its observation index and coordinates (requested one trip ahead of their use by the bucket loop below)
  struct Cand { int oj; double ox[NC]; };
  auto fetch = [&](int64_t pos) {
    Cand cd;
    cd.oj = q.sorted[pos];
#pragma unroll
    for (int c = 0; c < NC; ++c) cd.ox[c] = q.sxyz[pos * NC + c];
    return cd;
  };
  // points of this lane's group that exist (the last tile of a block may be ragged)
  bool ptok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ptok[i] = 4 * pg + i < npts;
  int cntl[4] = {0, 0, 0, 0};          // local observations of point 4 pg + i met by THIS lane (one candidate column); summed at the end
  auto weigh = [&](bool have, const Cand& cd) {
#pragma clang fp contract(off)      // (the same bits in every kernel this is inlined into, see gc_taper_fast)
    const int oj = cd.oj;
    double ox[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ox[c] = cd.ox[c];
    f4w wq = {0.f, 0.f, 0.f, 0.f};
    bool use4[4];
    if constexpr (TAPER == MIA_TAPER_GC) {
      // float32 weights of the four pairs first; the (rare) pairs whose weight is within 1e-4 of eps are collected and weighed again
      // in float64 behind ONE wave-level test per trip (round 4: one test and one branch per pair)
      float wf4[4];
      float dmin = 3.0e38f;          // smallest |weight - eps| of the lane's four pairs: ONE comparison per trip instead of one per pair
      const float amb_thr = 1e-4f * epsf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double d2[MIA_MAX_RADII] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const double dx = ox[c] - gxr[i][c];
#pragma unroll
          for (int r = 0; r < (NC == 1 ? 1 : MIA_MAX_RADII); ++r)
            if (grp[c] == r) d2[r] = __builtin_fma(dx, dx, d2[r]);
        }
        float wf = 1.0f;
        if constexpr (NC == 1) {
          wf = gc_taper_fast_1d(ox[0] - gxr[i][0], twoc0, icf[0], ccf[0]);
        } else {
#pragma unroll
          for (int r = 0; r < MIA_MAX_RADII; ++r)
            if (r < n_r) wf *= gc_taper_fast(d2[r], fc2[r], icf[r], ccf[r], ic2f[r]);
        }
        wf4[i] = wf;
        dmin = __builtin_fminf(dmin, __builtin_fabsf(wf - epsf));
      }
      if (__any(dmin < amb_thr)) {      // (pairs that do not count may trigger it too: the test per pair follows)
        // the decision is the float64 one: where float32 put an ambiguous weight on the wrong side of eps, the weight moves to the
        // float value next to eps on the right side (a change of the order of its own rounding error, ~1e-6 relative, of a weight
        // of 1e-5) -- so that every decision below is ONE comparison of the float32 weight
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (have && ptok[i] && __builtin_fabsf(wf4[i] - epsf) < amb_thr) {
            double wgt = 1.0;
            for (int r = 0; r < n_r; ++r) {
              double d2r = 0.0;
              for (int c = 0; c < NC; ++c)
                if (grp[c] == r) { const double dx = ox[c] - gxr[i][c]; d2r = __builtin_fma(dx, dx, d2r); }
              wgt *= gc_taper_d2(d2r, q.inv_c[r], q.cc[r]);
            }
            const bool u64 = wgt > q.eps;
            if (u64 != (wf4[i] > epsf)) wf4[i] = u64 ? __uint_as_float(__float_as_uint(epsf) + 1u) : epsf;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        use4[i] = wf4[i] > epsf && have && ptok[i];
        wq[i] = use4[i] ? __builtin_amdgcn_sqrtf(wf4[i]) : 0.0f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double d2[MIA_MAX_RADII] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const double dx = ox[c] - gxr[i][c];
#pragma unroll
          for (int r = 0; r < (NC == 1 ? 1 : MIA_MAX_RADII); ++r)
            if (grp[c] == r) d2[r] = __builtin_fma(dx, dx, d2[r]);
        }
        double wgt = 1.0;
#pragma unroll
        for (int r = 0; r < (NC == 1 ? 1 : MIA_MAX_RADII); ++r)
          if (r < n_r) wgt *= gc_inf_taper_d2(d2[r], q.inv_c[r], q.cc[r]);
        use4[i] = have && ptok[i] && wgt > q.eps;
        wq[i] = use4[i] ? (float)(wgt * rsqrt_f64(wgt)) : 0.0f;
      }
    }
    bool anyu = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      anyu = anyu || use4[i];
      cntl[i] += use4[i] ? 1 : 0;
    }
    // a candidate is a member of the union when any of its four lanes (one per point group) uses it
    const unsigned long long anyb = __ballot(anyu);
    const unsigned memb = (unsigned)((anyb | (anyb >> 16) | (anyb >> 32) | (anyb >> 48)) & 0xffffull);
    const bool member = (memb >> cl) & 1u;
    const int u = ubase + __popc(memb & ((1u << cl) - 1u));
    if (member && u < UMAX) {
      if (pg == 0) ukey[u] = oj;
      *reinterpret_cast<f4w*>(Wt + u * 16 + 4 * pg) = wq;
    }
    ubase += __popc(memb);
  };
  if constexpr (BUCKET) {
    const int cap = hd->bucket_cap;
    const int nlast = empty ? 0 : hi[last] - lo[last] + 1;
    const long long ncb = nrows * nlast;                     // cells of the box
    box_overflow = box_overflow || ncb > 64;
    overflow = overflow || ncb > 64;
    int mycid = 0, mycnt = 0;
    if (!overflow && lane < (int)ncb) {
      const int row = NC == 1 ? 0 : lane / nlast, cc = lane - row * nlast;      // (one coordinate: one row of cells, no division)
      int base_cell = 0;
      if (nc == 2) base_cell = (lo[0] + row) * hd->n[1];
      else if (nc == 3) base_cell = ((lo[0] + row / n1) * hd->n[1] + (lo[1] + row % n1)) * hd->n[2];
      mycid = base_cell + lo[last] + cc;
      mycnt = q.start[mycid];
      mycnt = mycnt > cap ? cap : mycnt;
    }
    // inclusive prefix over the box's cells (lane = cell): shifts inside the rows of sixteen lanes (zero comes in from the left),
    // then each row's total into the rows above it -- six DPP additions (round 4: six shuffles through LDS with a compare and a
    // select each)
    int incl = mycnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);      // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);      // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);      // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);      // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3 (lane 63: the total)
    const int total = overflow ? 0 : __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
    int* pref = uinv;                                        // (scratch until the ranks are formed: [<= 64] exclusive prefix | cell id)
    pref[lane] = incl - mycnt;
    MIA_TL_SYNC();
    MIA_TL_STAMP(2);
    const int ncbi = (int)(ncb > 64 ? 64 : ncb);
    // The candidates are gathered ONCE, sixty-four per pass with lane = candidate (round 4 located and fetched the sixteen of a trip in
    // all four point groups alike, every trip): position in the bucket arrays = the cell whose prefix range holds the candidate's
    // place in the flat sequence; index and coordinates go to LDS, from where the trips of sixteen take them.  Same order of
    // candidates as before -- same union, same ranks, same bits -- with one prefix search and one memory round trip per pass.
    for (int qp = 0; qp < total; qp += 64) {
      {
        const int qi = qp + lane;
        const int qc = qi < total ? qi : total - 1;          // (the last candidate again where there is none)
        int sel = 0;                                         // the cell of candidate qc: last cell whose prefix is <= qc
        for (int i = 1; i < ncbi; ++i) sel += pref[i] <= qc ? 1 : 0;
        const int cid = __shfl(mycid, sel, 64), pf = pref[sel];
        const Cand cd = fetch((int64_t)cid * cap + (qc - pf));
        coj[lane] = cd.oj;
#pragma unroll
        for (int c = 0; c < NC; ++c) cox[lane * NC + c] = cd.ox[c];
      }
      MIA_TL_SYNC();
      MIA_TL_STAMP(3);
      const int npass = total - qp < 64 ? total - qp : 64;
      for (int q0 = 0; q0 < npass; q0 += 16) {
        Cand cur;
        cur.oj = coj[q0 + cl];
#pragma unroll
        for (int c = 0; c < NC; ++c) cur.ox[c] = cox[(q0 + cl) * NC + c];
        weigh(q0 + cl < npass, cur);
      }
      MIA_TL_SYNC();                                         // (the next pass writes over the candidates)
    }
    MIA_TL_SYNC();
    for (int s_ = lane; s_ < kTlUmax; s_ += 64) uinv[s_] = -1;
  } else {
    for (int row = 0; row < (overflow ? 0 : (int)nrows); ++row) {
      int base_cell = 0;
      if (nc == 2) base_cell = (lo[0] + row) * hd->n[1];
      else if (nc == 3) base_cell = ((lo[0] + row / n1) * hd->n[1] + (lo[1] + row % n1)) * hd->n[2];
      const int beg = __builtin_amdgcn_readfirstlane(q.start[base_cell + lo[last]]);
      const int end = __builtin_amdgcn_readfirstlane(q.start[base_cell + hi[last] + 1]);
      for (int pos0 = beg; pos0 < end; pos0 += 16) {
        const bool have = pos0 + cl < end;
        weigh(have, fetch(have ? pos0 + cl : end - 1));
      }
    }
  }
  const int U = ubase;
  overflow = overflow || U > UMAX;
  MIA_TL_SYNC();
  MIA_TL_STAMP(4);
  // rank of every member by observation index -> slot
  if (!overflow && U <= 32) {
    // at most 32 members (the fused kernel's shapes): BOTH halves of the wavefront count -- lane u and lane u + 32 take sixteen keys
    // each and add up (one round of LDS reads and sixteen compare-adds instead of two; slots beyond U hold a key that counts for nobody)
    if (lane >= U && lane < 32) ukey[lane] = 0x7fffffff;
    MIA_TL_SYNC();
    const int u = lane & 31, v0 = (lane >> 5) << 4;
    const int key = ukey[u];
    int rk = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) rk += ukey[v0 + v] < key ? 1 : 0;
    typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
    const u2v_ r2 = __builtin_amdgcn_permlane32_swap((unsigned)rk, (unsigned)rk, false, false);
    rk = (int)(r2.x + r2.y);
    if (lane < U) uinv[16 * (rk >> 4) + 4 * (rk & 3) + ((rk >> 2) & 3)] = u;
  } else if (!overflow) {
    for (int u = lane; u < U; u += 64) {
      const int key = ukey[u];
      int rk = 0;
      for (int v = 0; v < U; ++v) rk += ukey[v] < key ? 1 : 0;
      uinv[16 * (rk >> 4) + 4 * (rk & 3) + ((rk >> 2) & 3)] = u;
    }
  }
  MIA_TL_SYNC();
  MIA_TL_SYNC();
  MIA_TL_STAMP(5);
  // longest list of the tile: a point's count is the sum over the sixteen candidate columns (lanes) of its group
  int cnt4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int v = cntl[i];
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);      // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);      // row_ror:8
    cnt4[i] = v;
  }
  int mx = cnt4[0] > cnt4[1] ? cnt4[0] : cnt4[1];
  mx = cnt4[2] > mx ? cnt4[2] : mx;
  mx = cnt4[3] > mx ? cnt4[3] : mx;
  const int m0 = __builtin_amdgcn_readlane(mx, 0), m1 = __builtin_amdgcn_readlane(mx, 16);
  const int m2 = __builtin_amdgcn_readlane(mx, 32), m3 = __builtin_amdgcn_readlane(mx, 48);
  const int m01 = m0 > m1 ? m0 : m1, m23 = m2 > m3 ? m2 : m3;
  const int longest = m01 > m23 ? m01 : m23;
  TileLocOut out;
  out.U = U; out.longest = longest; out.npts = npts; out.overflow = overflow; out.box_overflow = box_overflow;
  return out;
}

}  // namespace mia

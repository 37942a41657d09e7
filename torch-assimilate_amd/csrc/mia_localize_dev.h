// Device-side pieces of the Gaspari-Cohn localisation shared by the stand-alone neighbour-list kernel
// (localize.hip) and the analysis kernels that scan the observation index themselves (fused route).
// Reference: GaspariCohn.localize_obs, pytassim/localization/gaspari_cohn.py:97-136.
#pragma once
#include "mia_common.h"

namespace mia {

struct IndexHeader {        // lives at the start of the workspace, written on device
  // order-preserving integer keys of the coordinate extrema, both kept as running MAXIMA so that a
  // plain zero fill initialises them: kmax = max key(x), kmin_inv = max ~key(x)
  unsigned long long kmax[MIA_MAX_COORD];
  unsigned long long kmin_inv[MIA_MAX_COORD];
  double mn[MIA_MAX_COORD];
  double invh[MIA_MAX_COORD];
  int n[MIA_MAX_COORD];
  int ncell;
  unsigned done_bbox, done_count;   // workgroups finished (the last one runs the serial follow-up stage)
  // ---- what survives from one build to the next on the same workspace (bucket index of the step driver, see index_bucket_kernel)
  double cutoff[MIA_MAX_COORD];     // minimal cell edge (2 c per coordinate) the cell grid was derived for
  int bucket_cap;                   // entries per cell of the bucket arrays (a power of two), 0 = none
  unsigned magic;                   // kIndexMagic: mn / invh / n / ncell describe a cell grid
  unsigned err;                     // bucket build: kIndexErrBox / kIndexErrFull (cleared by the tile-list kernel's last workgroup)
  unsigned done_tiles;              // tile-list workgroups finished (the last one cleans the per-cell counts and err)
};
constexpr unsigned kIndexMagic = 0x6d696131u;
constexpr unsigned kIndexErrBox = 1u;      // an observation lies outside the stored box, or the radii changed: the box must be rebuilt
constexpr unsigned kIndexErrFull = 2u;     // a cell holds more observations than a bucket has entries: scan-based index instead

__device__ inline int cell_coord(double x, double mn, double invh, int n) {
  double f = floor((x - mn) * invh);
  f = f < -2.0 ? -2.0 : f;
  f = f > double(n) + 1.0 ? double(n) + 1.0 : f;
  return (f == f) ? int(f) : -2;  // NaN coordinate -> no cell
}

// 1/sqrt(x) in float64 without the (slow, correctly rounded) library sqrt/div: float seed, two Newton
// steps (relative error ~1e-15, far inside what the `weight > eps` decision can resolve)
__device__ inline double rsqrt_f64(double x) {
  double y = (double)__builtin_amdgcn_rsqf((float)x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// Gaspari-Cohn taper from the squared distance: r = sqrt(d2)/c and 1/r = c/sqrt(d2) both come from
// one reciprocal square root, so the -2/(3r) term of the outer branch needs no division
__device__ inline double gc_taper_d2(double d2, double inv_c, double c) {
  if (!(d2 > 0.0)) return d2 == 0.0 ? 1.0 : 0.0;            // r = 0 -> 1; NaN -> 0 (as gc_taper)
  const double y = rsqrt_f64(d2);
  const double r = d2 * y * inv_c, rinv = c * y;
  const double f1 = (((-0.25 * r + 0.5) * r + 0.625) * r - 5.0 / 3.0) * r * r + 1.0;
  const double f2 = ((((r * (1.0 / 12.0) - 0.5) * r + 0.625) * r + 5.0 / 3.0) * r - 5.0) * r + 4.0 - (2.0 / 3.0) * rinv;
  return r < 1.0 ? f1 : (r < 2.0 ? f2 : 0.0);
}

// the same for the form-factor-infinity taper (GaspariCohnInf, gaspari_cohn.py:139-254)
__device__ inline double gc_inf_taper_d2(double d2, double inv_c, double c) {
  if (!(d2 > 0.0)) return d2 == 0.0 ? 1.0 : 0.0;
  const double y = rsqrt_f64(d2);
  return gc_inf_taper_rinv<double>(d2 * y * inv_c, c * y);
}
__device__ inline double taper_d2(int taper, double d2, double inv_c, double c) {
  return taper == MIA_TAPER_GC_INF ? gc_inf_taper_d2(d2, inv_c, c) : gc_taper_d2(d2, inv_c, c);
}

// what a wavefront needs to find the local observations of one grid point
struct ScanParams {
  const double* grid;   // [G][nc]
  const double* sxyz;   // [P][nc] observation coordinates in cell order (bucket index: [bucket_total][nc], cell c at c * bucket_cap)
  const IndexHeader* hdr;
  const int* start;     // [ncell + 1] (bucket index: the per-cell counts)
  const int* sorted;    // [P] observation index in cell order (bucket index: [bucket_total])
  int nc, n_r;
  int group[MIA_MAX_COORD];
  double inv_c[MIA_MAX_RADII];
  double cc[MIA_MAX_RADII];
  double eps;
  int taper;            // MIA_TAPER_*
  // the float32 tile taper's constants, formed on the host (every wavefront used to convert them itself): 4 c^2, 1 / c, c, 1 / c^2
  double four_c2[MIA_MAX_RADII];
  float inv_c_f[MIA_MAX_RADII], c_f[MIA_MAX_RADII], inv_c2_f[MIA_MAX_RADII], eps_f;
};
// what the analysis kernel's fused variant (letkf_tile2f.hip) needs to localise its tiles itself: the scan over the step's bucket
// index and the step's counters
struct Tile2Loc {
  ScanParams scan;
  int32_t* stats;        // [0] longest list (running maximum), [1] tiles whose union did not fit (+ MIA_TILE_BOX_OVERFLOW)
  // the list bound the step was sized for.  The fused kernel reports a tile's longest list when it EXCEEDS this bound (the step is
  // then redone) and otherwise only for one tile in 64: stats[0] is a SAMPLED maximum on this route (bit kStepSampledLongest of the
  // step's error word says so) -- every tile guarding its atomic with a load of the running maximum was a dependent round trip to
  // ONE address of all resident wavefronts at once, 6 us of the kernel's 43 at config 2 (tools/tile2f_stamps.py)
  int longest_bound;
};
constexpr int kStepSampledLongest = 64;      // status bit in counters[3] / [7] (include/mia_letkf.h: MIA_STEP_STATUS_SAMPLED)
constexpr int kStepNonfinite = 128;          // ... MIA_STEP_STATUS_NONFINITE: some point of the fused kernel's launch carries MIA_FLAG_NONFINITE

// One wavefront scans the 3^d cells around grid point g (the innermost coordinate's three cells are one
// contiguous range), evaluates distance and taper in float64 and compacts the observations whose weight
// exceeds eps with a wave ballot, in cell order, ascending index inside a cell.  The first `cap` survivors
// are written as (index, sqrt(weight)); the return value is the true count (may exceed cap).
template <typename WT>
__device__ inline int scan_neighbours(const ScanParams& p, int64_t g, int lane, int cap, int* oidx, WT* ow) {
  const IndexHeader* h = p.hdr;
  double gx[MIA_MAX_COORD];
  int cg[MIA_MAX_COORD];
  for (int c = 0; c < MIA_MAX_COORD; ++c) { gx[c] = 0.0; cg[c] = 0; }
  for (int c = 0; c < p.nc; ++c) {
    gx[c] = p.grid[g * p.nc + c];
    cg[c] = cell_coord(gx[c], h->mn[c], h->invh[c], h->n[c]);
  }
  int count = 0;
  const int nc = p.nc;
  const int n_outer = nc == 1 ? 1 : (nc == 2 ? 3 : 9);
  const int last = nc - 1;
  int lo_l = cg[last] - 1, hi_l = cg[last] + 1;
  lo_l = lo_l < 0 ? 0 : lo_l;
  hi_l = hi_l > h->n[last] - 1 ? h->n[last] - 1 : hi_l;
  for (int o = 0; o < n_outer; ++o) {
    int base_cell = 0;
    bool ok = lo_l <= hi_l;
    if (nc >= 2) {
      int d0 = (nc == 2) ? (o - 1) : (o / 3 - 1);
      int c0 = cg[0] + d0;
      ok = ok && c0 >= 0 && c0 < h->n[0];
      base_cell = c0;
      if (nc == 3) {
        int c1 = cg[1] + (o % 3 - 1);
        ok = ok && c1 >= 0 && c1 < h->n[1];
        base_cell = base_cell * h->n[1] + c1;
      }
      base_cell *= h->n[last];
    }
    if (!ok) continue;
    const int beg = p.start[base_cell + lo_l], end = p.start[base_cell + hi_l + 1];
    for (int b = beg; b < end; b += 64) {
      const int pos = b + lane;
      bool use = false;
      int j = -1;
      double wgt = 0.0;
      if (pos < end) {
        j = p.sorted[pos];
        double d2[MIA_MAX_RADII] = {0.0, 0.0, 0.0};
        for (int c = 0; c < nc; ++c) {
          double dx = p.sxyz[(int64_t)pos * nc + c] - gx[c];
          d2[p.group[c]] += dx * dx;
        }
        wgt = 1.0;
        for (int r = 0; r < p.n_r; ++r) wgt *= taper_d2(p.taper, d2[r], p.inv_c[r], p.cc[r]);
        use = wgt > p.eps;
      }
      const unsigned long long mask = __ballot(use);
      if (use) {
        const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < cap) { oidx[slot] = j; ow[slot] = WT(wgt * rsqrt_f64(wgt)); }
      }
      count += __popcll(mask);
    }
  }
  return count;
}

// layout of the index workspace (built by mia_letkf_index_build_f64)
struct IndexLayout {
  IndexHeader* hdr; int* start; int* cursor; int* sorted; int* cell_of; int* rank_of; double* sxyz; size_t bytes; size_t cap;
  int* bidx; double* bxyz; size_t bucket_total;      // bucket index: [bucket_total] observation index / [bucket_total][nc] coordinates
};
static inline size_t index_cell_cap(int64_t P) {
  int64_t cap = 2 * P;
  if (cap < 1024) cap = 1024;
  if (cap > (int64_t)1 << 24) cap = (int64_t)1 << 24;
  return (size_t)cap;
}
static inline IndexLayout index_layout(void* ws, int64_t P, int nc) {
  // (offsets first, pointers last: the size query passes a null base, and arithmetic on a null pointer is undefined behaviour --
  //  found by the UBSan build of tests/test_host_sanitizers.py)
  IndexLayout L;
  L.cap = index_cell_cap(P);
  L.bucket_total = (size_t)(P > 0 ? 4 * P : 0) + 64;
  size_t o = 0;
  const size_t o_hdr = o; o += align_up(sizeof(IndexHeader), 256);
  const size_t o_start = o; o += align_up((L.cap + 1) * sizeof(int), 256);
  const size_t o_cursor = o; o += align_up(L.cap * sizeof(int), 256);
  const size_t o_sorted = o; o += align_up((size_t)P * sizeof(int) + 4, 256);
  const size_t o_cell = o; o += align_up((size_t)P * sizeof(int) + 4, 256);
  const size_t o_rank = o; o += align_up((size_t)P * sizeof(int) + 4, 256);
  const size_t o_sxyz = o; o += align_up((size_t)P * (size_t)nc * sizeof(double) + 8, 256);
  const size_t o_bidx = o; o += align_up(L.bucket_total * sizeof(int), 256);
  const size_t o_bxyz = o; o += align_up(L.bucket_total * (size_t)nc * sizeof(double), 256);
  L.bytes = o;
  char* base = (char*)ws;
  auto at = [&](size_t off) -> char* { return base ? base + off : nullptr; };
  L.hdr = (IndexHeader*)at(o_hdr); L.start = (int*)at(o_start); L.cursor = (int*)at(o_cursor); L.sorted = (int*)at(o_sorted);
  L.cell_of = (int*)at(o_cell); L.rank_of = (int*)at(o_rank); L.sxyz = (double*)at(o_sxyz);
  L.bidx = (int*)at(o_bidx); L.bxyz = (double*)at(o_bxyz);
  return L;
}

// host side, defined in localize.hip
struct PackJob;
// up to three small int32 buffers cleared by the first index kernel (only honoured when P > 0: the kernel runs)
struct ZeroJob { int32_t* ptr[3]; int64_t n[3]; };
int index_build_impl(const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                     const double* gc_c, int n_r, void* ws, size_t ws_bytes, hipStream_t stream,
                     const PackJob* pack = nullptr, const ZeroJob* zero = nullptr, bool header_clean = false,
                     bool sort_cells = true);
// neighbour lists of grid points [g0, g1) (mia_letkf_localize_f64 without the argument checks of the C entry);
// pack: float32 record packing job executed inside the first index kernel; stats_zeroed: stats are cleared by
// the caller or listed in `zero`
int localize_impl(const double* grid_xyz, int64_t g0, int64_t g1, const double* obs_xyz, int64_t P, int n_coord,
                  const int32_t* coord_group, const double* gc_c, int n_r, double gc_eps, int p_cap,
                  int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w, int32_t* stats, void* ws, size_t ws_bytes,
                  hipStream_t stream, const PackJob* pack, bool stats_zeroed, const ZeroJob* zero,
                  int taper = MIA_TAPER_GC, bool header_clean = false, bool sort_cells = true);
int localize_lists_impl(const double* grid_xyz, int64_t g0, int64_t g1, int64_t P, int n_coord, const int32_t* coord_group,
                        const double* gc_c, int n_r, double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx,
                        double* nbr_w, int32_t* stats, void* ws, hipStream_t stream, const PackJob* pack, int taper);
// lists of the points flagged MIA_FLAG_RETRY into the order a sorted index gives (see sort_flagged_lists_kernel)
int sort_flagged_lists(const int32_t* flags, const int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w, int64_t ng, int p_cap,
                       void* ws, int64_t P, int n_coord, hipStream_t stream);
int make_scan_params(ScanParams* sp, const double* grid_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                     const double* gc_c, int n_r, double gc_eps, void* ws, int taper = MIA_TAPER_GC, bool bucket = false);
// bucket index (step driver): ONE kernel bins the observations into fixed-capacity cells of the cell grid the workspace's header
// describes (fresh_box: the bounding-box kernel runs first and rewrites the header); errors go to the header (IndexHeader::err)
struct SplitPackJob;
int index_bucket_build_impl(const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group, const double* gc_c, int n_r,
                            void* ws, size_t ws_bytes, hipStream_t stream, const ZeroJob* zero, bool fresh_box,
                            const SplitPackJob* spack = nullptr, int* counts = nullptr);

}  // namespace mia

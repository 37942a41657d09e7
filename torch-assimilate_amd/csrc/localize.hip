// Gaspari-Cohn localisation on gfx950: uniform-cell observation index + per-grid-point
// neighbour lists.  Stands in for GaspariCohn.localize_obs evaluated once per grid point
// (pytassim/localization/gaspari_cohn.py:97-136 called from interface/wrapper.py:88-91).
//
// The reference evaluates the taper against ALL P observations for every grid point
// (O(G*P), :124-134).  Here the observations are binned once into cells whose edge is at
// least the taper's support (2*c per coordinate), and one 64-lane wavefront per grid point
// scans only the 3^d neighbouring cells, evaluates distance + taper in float64 (the
// reference's dtype, so `weight > eps` takes the reference's decision) and compacts the
// survivors with a wave ballot.  Integer/byte work, HBM/L2-bound: no MFMA here.
#include <cstdlib>
#include "mia_common.h"
#include "mia_options.h"
#include "mia_localize_dev.h"
#include "mia_pack_dev.h"
#include "mia_tiles.h"

namespace mia {

// The preparation kernels are short, latency-bound chains.  When consecutive steps are pipelined they share the
// SIMDs with the previous step's analysis kernel (VALU-bound, every wave slot taken): at equal priority the
// round-robin issue arbitration stretched localize_kernel from 35 to 200 us and made the preparation chain the
// critical path.  Raising the waves' issue priority (3) let them through; alone on the GPU it changes nothing.
// With the split-precision analysis kernel at two waves per SIMD the preparation fits beside it and the boost costs the
// analysis kernel more than it gains the chain: 0.0863 -> 0.0844 ms per step with priority 0 (1: 0.0875), now the default.

struct IndexParams {
  const double* obs;  // [P][nc]
  int64_t P;
  int nc;
  double cutoff[MIA_MAX_COORD];  // minimal cell edge per coordinate (= 2 * c of its group)
  int cell_cap;
  IndexHeader* hdr;
  int* start;    // [cell_cap + 1]  counts -> exclusive starts
  int* cursor;   // [cell_cap]  per-cell counts: zero outside the counting kernel .. the scan (which puts them back to zero)
  int* sorted;   // [P]      observation index, cell-major, ascending inside a cell
  int* cell_of;  // [P]
  int* rank_of;  // [P]      arrival rank of an observation inside its cell (the counting kernel's atomic returns it)
  double* sxyz;  // [P][nc]  coordinates in the same order (no second indirection in the scan)
  unsigned nb_bbox;      // workgroups of the first kernel that reduce the bounding box ...
  PackJob pack;          // ... the others (if any) pack observation records (pack.rec != nullptr)
  ZeroJob zero;          // small caller buffers cleared by the first kernel (saves their fill launches)
  int scatter_xyz;       // the cell sort is skipped (see index_build_impl): the scatter lays the coordinates out itself
  long long bucket_total;   // entries of the bucket arrays
  int* bidx; double* bxyz;
  unsigned nb_main;         // index_bucket_kernel: workgroups nb_main, nb_main + 1, ... pack split records (independent passenger)
  SplitPackJob spack;
};

__device__ inline unsigned long long dkey(double x) {     // total order of doubles as unsigned integers
  const long long b = __double_as_longlong(x);
  return b < 0 ? ~(unsigned long long)b : ((unsigned long long)b | 0x8000000000000000ull);
}
__device__ inline double dkey_inv(unsigned long long k) {
  const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// one thread: cell grid dimensions from the bounding box
__device__ inline void index_dims(const IndexParams& p) {
  double ext[MIA_MAX_COORD], h[MIA_MAX_COORD];
  long long n[MIA_MAX_COORD];
  for (int c = 0; c < MIA_MAX_COORD; ++c) { n[c] = 1; h[c] = 1.0; ext[c] = 0.0; p.hdr->mn[c] = 0.0; }
  for (int c = 0; c < p.nc; ++c) {
    double mn = 0.0, mx = 0.0;
    // (the extrema were produced by other workgroups' atomics: read them at the same scope)
    const unsigned long long kmax_c = __hip_atomic_load(&p.hdr->kmax[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long kmin_c = __hip_atomic_load(&p.hdr->kmin_inv[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kmax_c != 0ull) { mx = dkey_inv(kmax_c); mn = dkey_inv(~kmin_c); }
    // (the extrema have been read by their only reader: back to zero, which is what the next build on this workspace
    //  starts from -- the chain leaves the header as it found it and needs no fill launch per step)
    __hip_atomic_store(&p.hdr->kmax[c], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p.hdr->kmin_inv[c], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h[c] = p.cutoff[c] > 0.0 ? p.cutoff[c] : 1.0;
    // (one cell of margin on either side: observations that drift a little between two builds stay inside a box that the
    //  bucket index of the step driver keeps from one step to the next)
    mn -= h[c]; mx += h[c];
    p.hdr->mn[c] = mn;
    ext[c] = mx - mn;
    double nn = floor(ext[c] / h[c]) + 1.0;
    n[c] = nn > 1048576.0 ? 1048576 : (long long)nn;
  }
  // shrink until the table fits: halve the longest axis (cells only grow, so the
  // +-1 cell neighbourhood still covers the taper's support)
  while (n[0] * n[1] * n[2] > (long long)p.cell_cap) {
    int big = 0;
    for (int c = 1; c < p.nc; ++c) if (n[c] > n[big]) big = c;
    n[big] = (n[big] + 1) / 2;
  }
  for (int c = 0; c < p.nc; ++c) {
    double hc = ext[c] / double(n[c]);
    if (hc < h[c]) hc = h[c]; else hc *= (1.0 + 1e-12);
    p.hdr->invh[c] = 1.0 / hc;
    p.hdr->n[c] = int(n[c]);
  }
  for (int c = p.nc; c < MIA_MAX_COORD; ++c) { p.hdr->invh[c] = 1.0; p.hdr->n[c] = 1; }
  p.hdr->ncell = int(n[0] * n[1] * n[2]);
  for (int c = 0; c < MIA_MAX_COORD; ++c) p.hdr->cutoff[c] = c < p.nc ? p.cutoff[c] : 0.0;
  {   // bucket capacity: the largest power of two <= (bucket entries) / cells, at most 64; below 8 the buckets are not used
    long long per = (long long)p.bucket_total / (n[0] * n[1] * n[2]);
    int cap = 64;
    while (cap > per) cap >>= 1;
    p.hdr->bucket_cap = cap >= 8 ? cap : 0;
  }
  p.hdr->magic = kIndexMagic;
  __hip_atomic_store(&p.hdr->done_bbox, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bounding box of the observations: wave + workgroup reduction, one pair of 64-bit atomics per
// workgroup and coordinate; the workgroup that finishes last derives the cell grid from it (a separate
// one-thread launch cost as much as this whole kernel: ~5 us of dispatch latency each)
__global__ __launch_bounds__(256) void index_bbox_dims_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  if (blockIdx.x >= p.nb_bbox) {        // independent passenger: observation records for the analysis kernel
    extern __shared__ __attribute__((aligned(16))) float pack_lds[];       // 64 x (kp + 1), sized by the launch
    pack_obs_wave<float>(p.pack.Yb, p.pack.d, p.pack.k, p.P, p.pack.kp, p.pack.rec, (int64_t)(blockIdx.x - p.nb_bbox), pack_lds);
    return;
  }
  __shared__ unsigned long long sx[4], sn[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t stride = (int64_t)p.nb_bbox * blockDim.x;
  for (int q = 0; q < 3; ++q)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.zero.n[q]; i += stride) p.zero.ptr[q][i] = 0;
  for (int c = 0; c < p.nc; ++c) {
    unsigned long long kx = 0ull, kn = 0ull;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < p.P; j += stride) {
      const double x = p.obs[j * p.nc + c];
      if (x == x) { const unsigned long long k = dkey(x); kx = k > kx ? k : kx; kn = ~k > kn ? ~k : kn; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = __shfl_xor(kx, o, 64), b = __shfl_xor(kn, o, 64);
      kx = a > kx ? a : kx; kn = b > kn ? b : kn;
    }
    if (lane == 0) { sx[wv] = kx; sn[wv] = kn; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < nw; ++w) { kx = sx[w] > kx ? sx[w] : kx; kn = sn[w] > kn ? sn[w] : kn; }
      if (kx != 0ull) { atomicMax(&p.hdr->kmax[c], kx); atomicMax(&p.hdr->kmin_inv[c], kn); }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // the atomics above are performed (returning atomics: their results are back) before this one is issued
    const unsigned done = __hip_atomic_fetch_add(&p.hdr->done_bbox, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == p.nb_bbox - 1) index_dims(p);
  }
}

__device__ inline int obs_cell(const IndexHeader* h, const double* x, int nc) {
  int id = 0;
  for (int c = 0; c < nc; ++c) {
    int cc = cell_coord(x[c], h->mn[c], h->invh[c], h->n[c]);
    cc = cc < 0 ? 0 : (cc > h->n[c] - 1 ? h->n[c] - 1 : cc);
    id = id * h->n[c] + cc;
  }
  return id;
}

// exclusive scan of start[0 .. ncell] (entry ncell receives the total) by ONE wavefront, in tiles of 16 entries per
// lane: the 16 loads of a lane are independent (one memory round trip per tile), the wave scan runs on shuffles and the
// carry stays in a register.  A launch of its own with a single-wave workgroup.  (Not the tail of the counting kernel:
// as the tail of its last 256-thread workgroup it waited for four wave slots on one CU beside a bulk kernel -- 160 us
// instead of 22; as the tail of the last of its single-wave workgroups it had to read the other workgroups' counts with
// agent-scope loads, and count + scan took 52 us beside the analysis kernel instead of 10 + 19.)
__global__ __launch_bounds__(64) void index_scan_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  constexpr int T = 16;
  const int lane = threadIdx.x;
  const int n = p.hdr->ncell + 1;
  int carry = 0;
  for (int base = 0; base < n; base += 64 * T) {
    const int lo = base + lane * T;
    int vals[T];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < T; ++u) {
      const int i = lo + u;
      vals[u] = (i < n - 1) ? p.cursor[i] : 0;     // (counts of the previous kernel: visible at its end)
      if (i < n - 1) p.cursor[i] = 0;              // ... read by their only reader: zero again for the next build
      sum += vals[u];
    }
    int x = sum;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    int off = carry + x - sum;
#pragma unroll
    for (int u = 0; u < T; ++u) {
      if (lo + u < n) p.start[lo + u] = off;
      off += vals[u];
    }
    carry += __shfl(x, 63, 64);
  }
}

__global__ __launch_bounds__(64) void index_clear_kernel(uint32_t* p, size_t n) {
  MIA_PREP_PRIORITY();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}

// cell of every observation + per-cell counts.  256-thread workgroups of 15 VGPRs: they fit beside the analysis
// kernel's waves (see index_scan_kernel).
__global__ __launch_bounds__(64) void index_count_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j < p.P) {
    const int c = obs_cell(p.hdr, p.obs + j * p.nc, p.nc);
    p.cell_of[j] = c;
    p.rank_of[j] = atomicAdd(&p.cursor[c], 1);      // (the count IS the observation's place inside its cell: no second atomic)
  }
}

__global__ void index_scatter_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= p.P) return;
  int c = p.cell_of[j];
  int pos = p.start[c] + p.rank_of[j];
  p.sorted[pos] = int(j);
  if (p.scatter_xyz)
    for (int q = 0; q < p.nc; ++q) p.sxyz[(int64_t)pos * p.nc + q] = p.obs[j * p.nc + q];
}

// Lists built on an index whose cells were NOT sorted (lazy sort of the step driver) hold the right observations in
// arrival order.  The sixteen-points-per-wavefront kernel does not care (it ranks the union by observation index); the
// eigensolver kernel that redoes declined points sums in list order -- so the lists of exactly those points (flag
// MIA_FLAG_RETRY) are put into the order a sorted index would have given them: ascending (cell, observation index).
// One wavefront per flagged point, rank sort in place (lists of up to 128 entries: two per lane).
struct SortListsParams { const int32_t* flags; const int32_t* cnt; int32_t* idx; double* w; const int* cell_of; int64_t ng; int p_cap; };
__global__ __launch_bounds__(64) void sort_flagged_lists_kernel(SortListsParams p) {
  const int lane = threadIdx.x;
  for (int64_t pt = blockIdx.x; pt < p.ng; pt += gridDim.x) {
    if (!(p.flags[pt] & MIA_FLAG_RETRY)) continue;
    int n = p.cnt[pt];
    n = n > p.p_cap ? p.p_cap : n;
    n = n > 128 ? 128 : n;
    int32_t* li = p.idx + pt * p.p_cap;
    double* lw = p.w + pt * p.p_cap;
    unsigned long long key[2];
    int32_t id[2];
    double wv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      id[u] = e < n ? li[e] : 0;
      wv[u] = e < n ? lw[e] : 0.0;
      key[u] = e < n ? (((unsigned long long)(unsigned)p.cell_of[id[u]] << 32) | (unsigned)id[u]) : ~0ull;
    }
    int rank[2] = {0, 0};
    for (int u2 = 0; u2 < 2; ++u2)
      for (int l = 0; l < 64; ++l) {
        if (l + 64 * u2 >= n) break;                       // (wave-uniform)
        const unsigned long long other = __shfl(key[u2], l, 64);
#pragma unroll
        for (int u = 0; u < 2; ++u) rank[u] += other < key[u] ? 1 : 0;
      }
    __builtin_amdgcn_wave_barrier();                      // (every lane holds its entries: the stores below may overwrite)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (lane + 64 * u < n) { li[rank[u]] = id[u]; lw[rank[u]] = wv[u]; }
  }
}

// the scatter order inside a cell depends on atomic arrival: sort each cell's slice by observation
// index so that neighbour lists (and therefore summation order) are reproducible; then lay the
// coordinates out in that order.  One WAVE per cell (grid-stride over the cells): a cell of up to 64
// observations is ranked in registers (ids are distinct: rank = number of smaller ids) with one load and one
// store per lane; the thread-per-cell insertion sort this replaces spent 16 us in dependent global round trips.
__global__ __launch_bounds__(256) void index_sortcell_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  const int ncell = p.hdr->ncell;
  for (int c = wave; c < ncell; c += nwave) {
    const int lo = p.start[c], hi = p.start[c + 1], n = hi - lo;
    if (n <= 0) continue;
    if (n <= 64) {
      const int v = lane < n ? p.sorted[lo + lane] : 0x7fffffff;
      int rank = 0;
      for (int l = 0; l < n; ++l) rank += __shfl(v, l, 64) < v ? 1 : 0;
      if (lane < n) {
        p.sorted[lo + rank] = v;
        for (int q = 0; q < p.nc; ++q) p.sxyz[(int64_t)(lo + rank) * p.nc + q] = p.obs[(int64_t)v * p.nc + q];
      }
    } else {
      // larger cells (dense networks: hundreds of observations per cell): the same rank sort out of place -- ids copied to
      // the cell's slice of cell_of (no longer needed once the scatter has run), every lane ranks its share of the
      // elements against all of them.  O(n^2 / 64) per lane; the serial insertion sort this replaces took 13 ms on a
      // network with ~500 observations per cell.
      int* tmp = p.cell_of + lo;
      for (int i = lane; i < n; i += 64) tmp[i] = p.sorted[lo + i];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < n; i += 64) {
        const int v = tmp[i];
        int rank = 0;
        for (int l = 0; l < n; ++l) rank += tmp[l] < v ? 1 : 0;
        p.sorted[lo + rank] = v;
        for (int q = 0; q < p.nc; ++q) p.sxyz[(int64_t)(lo + rank) * p.nc + q] = p.obs[(int64_t)v * p.nc + q];
      }
    }
  }
}

struct LocalizeParams {
  ScanParams scan;
  int64_t g0, ng;
  int p_cap;
  int* cnt; int* idx; double* w; int* stats;
  // independent passenger: workgroups nb_main, nb_main + 1, ... pack the observation records for the analysis kernel
  // (single-wave workgroups, pack_obs_wave).  It used to ride in the FIRST kernel of the chain, whose successors then
  // waited for 17 MB of traffic they do not depend on; only the analysis kernel needs the records.
  unsigned nb_main;
  PackJob pack;
  int64_t P;
};
#define MIA_LOCALIZE_PASSENGER(p)                                                                                          \
  if (blockIdx.x >= (p).nb_main) {                                                                                         \
    extern __shared__ __attribute__((aligned(16))) float pack_lds_[];                                                       \
    pack_obs_wave<float>((p).pack.Yb, (p).pack.d, (p).pack.k, (p).P, (p).pack.kp, (p).pack.rec,                             \
                         (int64_t)(blockIdx.x - (p).nb_main), pack_lds_);                                                   \
    return;                                                                                                                \
  }

// One THREAD per grid point.  A wavefront-per-point version of this kernel (the scan_neighbours device
// function, still used by the fused analysis route) took 83 us for 1e5 points on MI355X although it
// executes only ~400 instructions per wavefront: every wavefront runs a chain of four dependent memory
// round trips (header -> cell range -> index -> coordinates) of ~1.5 us each and only 32 wavefronts fit
// a CU.  With one point per lane the same chain is amortised over 64 points and the candidate loop
// (~30 candidates x ~50 float64 operations) runs with all lanes busy.
// (106 VGPRs.  Capping it at 80, so that ONE retiring wave of a co-running analysis kernel -- 7 x 72 registers per
//  SIMD -- makes room, cost 24 spilled registers and gained nothing: 68 us beside the analysis kernel either way)
__global__ __launch_bounds__(64) void localize_kernel(LocalizeParams p) {
  MIA_PREP_PRIORITY();
  MIA_LOCALIZE_PASSENGER(p);
  const int64_t pt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int count = 0;
  if (pt < p.ng) {
    const ScanParams& q = p.scan;
    const IndexHeader* h = q.hdr;
    const int nc = q.nc;
    double gx[MIA_MAX_COORD];
    int cg[MIA_MAX_COORD];
    for (int c = 0; c < MIA_MAX_COORD; ++c) { gx[c] = 0.0; cg[c] = 0; }
    for (int c = 0; c < nc; ++c) {
      gx[c] = q.grid[(p.g0 + pt) * nc + c];
      cg[c] = cell_coord(gx[c], h->mn[c], h->invh[c], h->n[c]);
    }
    int* my_idx = p.idx + pt * p.p_cap;
    double* my_w = p.w + pt * p.p_cap;
    const int n_outer = nc == 1 ? 1 : (nc == 2 ? 3 : 9);
    const int last = nc - 1;
    int lo_l = cg[last] - 1, hi_l = cg[last] + 1;
    lo_l = lo_l < 0 ? 0 : lo_l;
    hi_l = hi_l > h->n[last] - 1 ? h->n[last] - 1 : hi_l;
    for (int o = 0; o < n_outer; ++o) {
      int base_cell = 0;
      bool ok = lo_l <= hi_l;
      if (nc >= 2) {
        const int d0 = (nc == 2) ? (o - 1) : (o / 3 - 1);
        const int c0 = cg[0] + d0;
        ok = ok && c0 >= 0 && c0 < h->n[0];
        base_cell = c0;
        if (nc == 3) {
          const int c1 = cg[1] + (o % 3 - 1);
          ok = ok && c1 >= 0 && c1 < h->n[1];
          base_cell = base_cell * h->n[1] + c1;
        }
        base_cell *= h->n[last];
      }
      if (!ok) continue;
      const int beg = q.start[base_cell + lo_l], end = q.start[base_cell + hi_l + 1];
      // four candidates per trip: their (independent) loads are issued together, so the ~1.5 us memory
      // latency is paid once per four candidates instead of once per candidate
      for (int pos0 = beg; pos0 < end; pos0 += 4) {
        double wg[4];
        int oj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pos = pos0 + u < end ? pos0 + u : end - 1;
          oj[u] = q.sorted[pos];
          double d2[MIA_MAX_RADII] = {0.0, 0.0, 0.0};
          for (int c = 0; c < nc; ++c) {
            const double dx = q.sxyz[(int64_t)pos * nc + c] - gx[c];
            d2[q.group[c]] += dx * dx;
          }
          double wgt = 1.0;
          for (int r = 0; r < q.n_r; ++r) wgt *= taper_d2(q.taper, d2[r], q.inv_c[r], q.cc[r]);
          wg[u] = pos0 + u < end ? wgt : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (wg[u] > q.eps) {
            if (count < p.p_cap) { my_idx[count] = oj[u]; my_w[count] = wg[u] * rsqrt_f64(wg[u]); }
            ++count;
          }
        }
      }
    }
    for (int s_ = count; s_ < p.p_cap; ++s_) { my_idx[s_] = -1; my_w[s_] = 0.0; }
    p.cnt[pt] = count;
  }
  // statistics: one (conditional) atomic per wavefront
  int mx = count, over = (pt < p.ng && count > p.p_cap) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(mx, o, 64); mx = a > mx ? a : mx; over += __shfl_xor(over, o, 64); }
  if ((threadIdx.x & 63) == 0) {
    if (mx > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], mx);
    if (over) atomicAdd(&p.stats[1], over);
  }
}

// FOUR LANES per grid point (sixteen points per wavefront): the candidate loop of localize_kernel, which evaluates four
// candidates per trip in every lane, spread over a quad -- one candidate per lane and trip, survivors compacted in lane
// order with a ballot, so the lists come out identical (cell order, ascending index inside a cell, same weights).  Four
// times as many wavefronts, each a quarter as long and half the registers: alone on the GPU the thread-per-point kernel
// runs at 1.5 waves per SIMD (1563 waves for 1e5 points) and is all latency; beside the analysis kernel of an earlier step
// (pipelined steps) its long-lived 106-register waves each kept a 168-register analysis wave out of its SIMD.
__global__ __launch_bounds__(64) void localize_quad_kernel(LocalizeParams p) {
  MIA_PREP_PRIORITY();
  MIA_LOCALIZE_PASSENGER(p);
  const int lane = threadIdx.x, sub = lane & 3;
  const int64_t pt = blockIdx.x * (int64_t)16 + (lane >> 2);
  const unsigned qshift = (unsigned)(lane & ~3);
  int count = 0;
  if (pt < p.ng) {
    const ScanParams& q = p.scan;
    const IndexHeader* h = q.hdr;
    const int nc = q.nc;
    double gx[MIA_MAX_COORD];
    int cg[MIA_MAX_COORD];
    for (int c = 0; c < MIA_MAX_COORD; ++c) { gx[c] = 0.0; cg[c] = 0; }
    for (int c = 0; c < nc; ++c) {
      gx[c] = q.grid[(p.g0 + pt) * nc + c];
      cg[c] = cell_coord(gx[c], h->mn[c], h->invh[c], h->n[c]);
    }
    int* my_idx = p.idx + pt * p.p_cap;
    double* my_w = p.w + pt * p.p_cap;
    const int n_outer = nc == 1 ? 1 : (nc == 2 ? 3 : 9);
    const int last = nc - 1;
    int lo_l = cg[last] - 1, hi_l = cg[last] + 1;
    lo_l = lo_l < 0 ? 0 : lo_l;
    hi_l = hi_l > h->n[last] - 1 ? h->n[last] - 1 : hi_l;
    for (int o = 0; o < n_outer; ++o) {
      int base_cell = 0;
      bool ok = lo_l <= hi_l;
      if (nc >= 2) {
        const int d0 = (nc == 2) ? (o - 1) : (o / 3 - 1);
        const int c0 = cg[0] + d0;
        ok = ok && c0 >= 0 && c0 < h->n[0];
        base_cell = c0;
        if (nc == 3) {
          const int c1 = cg[1] + (o % 3 - 1);
          ok = ok && c1 >= 0 && c1 < h->n[1];
          base_cell = base_cell * h->n[1] + c1;
        }
        base_cell *= h->n[last];
      }
      if (!ok) continue;
      const int beg = q.start[base_cell + lo_l], end = q.start[base_cell + hi_l + 1];
      // two candidates per lane and trip (positions pos0 + sub and pos0 + 4 + sub: their loads are requested together,
      // half as many dependent memory round trips per point); the survivors of the first four positions are placed
      // before those of the second four, i.e. still in position order
      for (int pos0 = beg; pos0 < end; pos0 += 8) {          // (the same bounds in the four lanes of a quad)
        bool have[2];
        int oj[2];
        double wgt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          have[u] = pos0 + 4 * u + sub < end;
          const int pos = have[u] ? pos0 + 4 * u + sub : end - 1;
          oj[u] = q.sorted[pos];
          double d2[MIA_MAX_RADII] = {0.0, 0.0, 0.0};
          for (int c = 0; c < nc; ++c) {
            const double dx = q.sxyz[(int64_t)pos * nc + c] - gx[c];
            d2[q.group[c]] += dx * dx;
          }
          wgt[u] = 1.0;
          for (int r = 0; r < q.n_r; ++r) wgt[u] *= taper_d2(q.taper, d2[r], q.inv_c[r], q.cc[r]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool use = have[u] && wgt[u] > q.eps;
          const unsigned quad = (unsigned)(__ballot(use) >> qshift) & 0xfu;
          if (use) {
            const int slot = count + __popc(quad & ((1u << sub) - 1u));
            if (slot < p.p_cap) { my_idx[slot] = oj[u]; my_w[slot] = wgt[u] * rsqrt_f64(wgt[u]); }
          }
          count += __popc(quad);
        }
      }
    }
    for (int s_ = count + sub; s_ < p.p_cap; s_ += 4) { my_idx[s_] = -1; my_w[s_] = 0.0; }
    if (sub == 0) p.cnt[pt] = count;
  }
  // statistics: one (conditional) atomic per wavefront
  int mx = count, over = (pt < p.ng && sub == 0 && count > p.p_cap) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(mx, o, 64); mx = a > mx ? a : mx; over += __shfl_xor(over, o, 64); }
  if (lane == 0) {
    if (mx > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], mx);
    if (over) atomicAdd(&p.stats[1], over);
  }
}

// One WAVEFRONT per grid point, for long lists (dense observation networks): the thread-per-point kernel above walks a
// list of several hundred candidates serially in every lane (p ~ 1000: 13 ms for 2e4 points, 20x the analysis itself);
// here 64 candidates are evaluated per trip and compacted with a ballot (scan_neighbours, the routine the fused
// analysis route uses).  Same order (cell order, ascending index inside a cell), same weights: identical lists.
__global__ __launch_bounds__(64) void localize_wave_kernel(LocalizeParams p) {
  MIA_PREP_PRIORITY();
  MIA_LOCALIZE_PASSENGER(p);
  const int lane = threadIdx.x;
  const int64_t pt = blockIdx.x;
  if (pt >= p.ng) return;
  int* my_idx = p.idx + pt * p.p_cap;
  double* my_w = p.w + pt * p.p_cap;
  const int count = scan_neighbours<double>(p.scan, p.g0 + pt, lane, p.p_cap, my_idx, my_w);
  for (int s_ = (count < p.p_cap ? count : p.p_cap) + lane; s_ < p.p_cap; s_ += 64) { my_idx[s_] = -1; my_w[s_] = 0.0; }
  if (lane == 0) {
    p.cnt[pt] = count;
    if (count > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], count);
    if (count > p.p_cap) atomicAdd(&p.stats[1], 1);
  }
}

struct FromDistParams {
  const double* dist; const int* cand; int64_t n_pts; int p_cap; int n_r;
  double inv_c[MIA_MAX_RADII]; double eps; int taper;
  int* cnt; int* idx; double* w; int* stats;
};

// caller-evaluated distances: taper, mask, compact (one wavefront per grid point)
__global__ __launch_bounds__(256) void localize_from_dist_kernel(FromDistParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t pt = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pt >= p.n_pts) return;
  int count = 0;
  int* my_idx = p.idx + pt * p.p_cap;
  double* my_w = p.w + pt * p.p_cap;
  // two-pass per 64-chunk so that in-place compaction never overwrites unread input:
  // slot <= source position always holds, and a chunk is fully read before it is written
  for (int b = 0; b < p.p_cap; b += 64) {
    const int s = b + lane;
    bool use = false; int j = -1; double wgt = 0.0;
    if (s < p.p_cap) {
      j = p.cand[pt * p.p_cap + s];
      if (j >= 0) {
        wgt = 1.0;
        for (int r = 0; r < p.n_r; ++r) {
          const double rr = p.dist[((int64_t)r * p.n_pts + pt) * p.p_cap + s] * p.inv_c[r];
          wgt *= p.taper == MIA_TAPER_GC_INF ? gc_inf_taper<double>(rr) : gc_taper<double>(rr);
        }
        use = wgt > p.eps;
      }
    }
    const unsigned long long mask = __ballot(use);
    __builtin_amdgcn_wave_barrier();
    if (use) {
      const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
      my_idx[slot] = j; my_w[slot] = sqrt(wgt);
    }
    count += __popcll(mask);
  }
  for (int s = count + lane; s < p.p_cap; s += 64) { my_idx[s] = -1; my_w[s] = 0.0; }
  if (lane == 0) {
    p.cnt[pt] = count;
    if (count > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], count);
  }
}

template <typename T, int TAPER>
__global__ void gc_kernel(const T* r, int64_t n, T* w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) w[i] = TAPER == MIA_TAPER_GC_INF ? gc_inf_taper<T>(r[i]) : gc_taper<T>(r[i]);
}

template <typename T, int TAPER>
static int taper_launch(const T* r, int64_t n, T* w, hipStream_t stream) {
  if (n < 0) return MIA_ERR_SIZE;
  if (n == 0) return MIA_OK;
  if (!r || !w) return MIA_ERR_NULL;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  gc_kernel<T, TAPER><<<dim3((unsigned)nb), dim3(256), 0, stream>>>(r, n, w);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// builds the cell index of the observations in ws (all kernels enqueued on stream)
int index_build_impl(const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                     const double* gc_c, int n_r, void* ws, size_t ws_bytes, hipStream_t stream, const PackJob* pack,
                     const ZeroJob* zero, bool header_clean, bool sort_cells) {
  if (P < 0 || n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (P > 2000000000LL) return MIA_ERR_UNSUPPORTED;
  if (!coord_group || !gc_c) return MIA_ERR_NULL;
  for (int c = 0; c < n_coord; ++c) if (coord_group[c] < 0 || coord_group[c] >= n_r) return MIA_ERR_SIZE;
  for (int r = 0; r < n_r; ++r) if (!(gc_c[r] > 0.0)) return MIA_ERR_SIZE;
  if (P == 0) return MIA_OK;
  if (!obs_xyz || !ws) return MIA_ERR_NULL;
  if (((uintptr_t)ws) & 255) return MIA_ERR_ALIGN;
  const IndexLayout L = index_layout(ws, P, n_coord);
  if (ws_bytes < L.bytes) return MIA_ERR_WORKSPACE;
  IndexParams ip;
  ip.obs = obs_xyz; ip.P = P; ip.nc = n_coord; ip.cell_cap = (int)L.cap;
  for (int c = 0; c < MIA_MAX_COORD; ++c) ip.cutoff[c] = c < n_coord ? 2.0 * gc_c[coord_group[c]] : 1.0;
  ip.hdr = L.hdr; ip.start = L.start; ip.cursor = L.cursor; ip.sorted = L.sorted; ip.cell_of = L.cell_of; ip.rank_of = L.rank_of;
  ip.sxyz = L.sxyz;
  ip.bucket_total = (long long)L.bucket_total; ip.bidx = L.bidx; ip.bxyz = L.bxyz;
  // SINGLE-WAVE workgroups throughout the chain: when steps are pipelined these kernels run beside the previous step's
  // analysis kernel, which fills every SIMD's register file (7 waves x 72 VGPRs at C2).  A lone wave takes the slot of
  // the next analysis wave that retires; a 4-wave workgroup needs one to retire on each SIMD of one CU at the same
  // moment and waited for the analysis grid to drain (160 us instead of 25, measured).
  constexpr unsigned kPrepThreads = 64;
  const unsigned nbP = (unsigned)((P + kPrepThreads - 1) / kPrepThreads);
  ip.nb_bbox = nbP < 256 ? nbP : 256;
  // The chain leaves its workspace as it needs to find it: the header's extrema / completion counters are running maxima /
  // counts from zero, put back to zero by their only reader; the per-cell counts are zeroed by the scan that reads them; the
  // starts are overwritten whole.  So a workspace whose last use was a complete build (header_clean: the step driver's
  // slots) needs no fill launch and no clearing pass at all; any other gets one fill of header + tables (single-wave
  // workgroups, not hipMemsetAsync: the runtime's fill kernel has 256-thread workgroups and waited 30-60 us for a CU
  // beside a bulk kernel).
  if (!header_clean) {
    const size_t words = ((char*)ip.sorted - (char*)ip.hdr) / sizeof(uint32_t);
    const unsigned nb = (unsigned)((words + kPrepThreads * 16 - 1) / (kPrepThreads * 16));
    index_clear_kernel<<<dim3(nb < 1024 ? (nb ? nb : 1) : 1024), dim3(kPrepThreads), 0, stream>>>(
        reinterpret_cast<uint32_t*>(ip.hdr), words);
    MIA_LAUNCH_CHECK();
  }
  ip.pack = PackJob{nullptr, nullptr, nullptr, 0, 0};
  ip.zero = zero ? *zero : ZeroJob{{nullptr, nullptr, nullptr}, {0, 0, 0}};
  unsigned nb_pack = 0;
  if (pack && pack->rec) {
    if ((P + 63) / 64 > 2000000000LL) return MIA_ERR_UNSUPPORTED;
    ip.pack = *pack;
    nb_pack = (unsigned)((P + 63) / 64);
  }
  const size_t pack_lds = nb_pack ? (size_t)64 * (ip.pack.kp + 1) * sizeof(float) : 0;      // (kp <= 132: 34 KB)
  index_bbox_dims_kernel<<<dim3(ip.nb_bbox + nb_pack), dim3(kPrepThreads), pack_lds, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  index_count_kernel<<<dim3(nbP), dim3(kPrepThreads), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  index_scan_kernel<<<dim3(1), dim3(64), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  ip.scatter_xyz = sort_cells ? 0 : 1;
  index_scatter_kernel<<<dim3(nbP), dim3(kPrepThreads), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  if (!sort_cells) return MIA_OK;      // (lazy sort: the scatter has laid the coordinates out; see sort_flagged_lists_kernel)
  const size_t sort_blocks = L.cap < 8192 ? (L.cap ? L.cap : 1) : 8192;      // one wave (cell) per workgroup
  index_sortcell_kernel<<<dim3((unsigned)sort_blocks), dim3(kPrepThreads), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Bucket index (step driver, tile route): the cell grid -- origin, cell edges, dimensions -- is the one the workspace's header
// already holds (from an earlier build on this workspace: the bounding box with its one-cell margin changes slowly or not at
// all between two assimilation steps), and every cell owns bucket_cap entries, so ONE kernel bins the observations: the cell's
// atomic count is the observation's place in its bucket.  No bounding-box pass, no scan, no scatter pass.  Everything the box
// is reused FOR is recomputed from this call's coordinates; that it still holds is checked here per observation (and the
// radii against the ones the grid was derived for): kIndexErrBox sends the step back through the bounding-box kernel,
// kIndexErrFull (a cell with more observations than a bucket holds) to the scan-based index.  The per-cell counts are zero on
// entry: the tile-list kernel's last workgroup puts them back (localize_tiles_kernel).
__global__ __launch_bounds__(64) void index_bucket_kernel(IndexParams p) {
  MIA_PREP_PRIORITY();
  MIA_PREP_PRIORITY();
  // Workgroup b bins observations 64 b .. 64 b + 63 AND packs their split records (the analysis kernel's, only it needs them): the
  // rows of Yb are requested first, straight into LDS; the binning runs while they travel.  One lean wavefront per 64 observations
  // -- at most 32 registers, k x 256 bytes of LDS -- so that it is placed BESIDE the five wavefronts per SIMD of the previous
  // steps' analysis kernels (96 of 512 registers each) instead of waiting for one of them to retire: with two kinds of workgroup,
  // 50 registers and 10.7 KB each, a preparation kernel took 24-45 us in the loop instead of 11 and the launch thread waited for it
  // a third of the time (profiles/r05_step_trace.txt)
  extern __shared__ __attribute__((aligned(16))) float bk_lds[];
  const bool packs = p.spack.rec != nullptr;
  if (packs) pack_split_lean_request(p.spack, p.P, (int64_t)blockIdx.x, bk_lds);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int q = 0; q < 3; ++q)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.zero.n[q]; i += stride) p.zero.ptr[q][i] = 0;
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j < p.P) {
    IndexHeader* h = p.hdr;
    const bool grid_ok = h->magic == kIndexMagic;
    bool ok = grid_ok && h->bucket_cap > 0;
    for (int c = 0; c < p.nc; ++c) ok = ok && h->cutoff[c] == p.cutoff[c];
    const int cap = h->bucket_cap;
    int id = 0;
    bool skip = false;
    double x[MIA_MAX_COORD] = {0.0, 0.0, 0.0};
    for (int c = 0; c < p.nc; ++c) {
      x[c] = p.obs[j * p.nc + c];
      const double f = floor((x[c] - h->mn[c]) * h->invh[c]);
      if (!(x[c] == x[c])) { skip = true; continue; }         // NaN coordinate: no cell (its weight is 0 everywhere)
      if (!(f >= 0.0 && f < double(h->n[c]))) { ok = false; continue; }
      id = id * h->n[c] + int(f);
    }
    if (!ok) {
      atomicOr(&h->err, (grid_ok && cap == 0) ? kIndexErrFull : kIndexErrBox);
    } else if (!skip) {
      const int pos = atomicAdd(&p.cursor[id], 1);
      if (pos >= cap) {
        atomicOr(&h->err, kIndexErrFull);
      } else {
        const int64_t e = (int64_t)id * cap + pos;
        p.bidx[e] = int(j);
        for (int c = 0; c < p.nc; ++c) p.bxyz[e * p.nc + c] = x[c];
      }
    }
  }
  if (packs) pack_split_lean_finish(p.spack, p.P, (int64_t)blockIdx.x, bk_lds);
}

int index_bucket_build_impl(const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group, const double* gc_c, int n_r,
                            void* ws, size_t ws_bytes, hipStream_t stream, const ZeroJob* zero, bool fresh_box,
                            const SplitPackJob* spack, int* counts) {
  if (P <= 0 || n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (P > 500000000LL) return MIA_ERR_UNSUPPORTED;
  if (!coord_group || !gc_c || !obs_xyz || !ws) return MIA_ERR_NULL;
  const IndexLayout L = index_layout(ws, P, n_coord);
  if (ws_bytes < L.bytes) return MIA_ERR_WORKSPACE;
  IndexParams ip;
  ip.obs = obs_xyz; ip.P = P; ip.nc = n_coord; ip.cell_cap = (int)L.cap;
  for (int c = 0; c < MIA_MAX_COORD; ++c) ip.cutoff[c] = c < n_coord ? 2.0 * gc_c[coord_group[c]] : 1.0;
  ip.hdr = L.hdr; ip.start = L.start; ip.cursor = L.cursor; ip.sorted = L.sorted; ip.cell_of = L.cell_of; ip.rank_of = L.rank_of;
  ip.sxyz = L.sxyz;
  ip.bucket_total = (long long)L.bucket_total; ip.bidx = L.bidx; ip.bxyz = L.bxyz;
  if (counts) ip.cursor = counts;        // (the per-cell counts of this build: one of the layout's two arrays, the caller alternates)
  ip.pack = PackJob{nullptr, nullptr, nullptr, 0, 0};
  ip.zero = ZeroJob{{nullptr, nullptr, nullptr}, {0, 0, 0}};
  ip.scatter_xyz = 0;
  constexpr unsigned kPrepThreads = 64;
  const unsigned nbP = (unsigned)((P + kPrepThreads - 1) / kPrepThreads);
  ip.nb_bbox = nbP < 256 ? nbP : 256;
  if (fresh_box) {
    // header + per-cell tables back to zero (a workspace of unknown history), then the bounding box and the cell grid
    const size_t words = ((char*)ip.sorted - (char*)ip.hdr) / sizeof(uint32_t);
    const unsigned nb = (unsigned)((words + kPrepThreads * 16 - 1) / (kPrepThreads * 16));
    index_clear_kernel<<<dim3(nb < 1024 ? (nb ? nb : 1) : 1024), dim3(kPrepThreads), 0, stream>>>(reinterpret_cast<uint32_t*>(ip.hdr), words);
    MIA_LAUNCH_CHECK();
    index_bbox_dims_kernel<<<dim3(ip.nb_bbox), dim3(kPrepThreads), 0, stream>>>(ip);
    MIA_LAUNCH_CHECK();
  }
  ip.zero = zero ? *zero : ZeroJob{{nullptr, nullptr, nullptr}, {0, 0, 0}};
  ip.nb_main = nbP;
  ip.spack = SplitPackJob{nullptr, nullptr, nullptr, 0};
  unsigned nb = nbP;
  size_t lds = 0;
  if (spack && spack->rec) {
    ip.spack = *spack;
    nb = (unsigned)((P + 1 + 63) / 64);      // (record P is the all-zero record)
    lds = split_pack_lean_lds(spack->k);
    if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)index_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  index_bucket_kernel<<<dim3(nb), dim3(kPrepThreads), lds, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// scan parameters over an already built index
int make_scan_params(ScanParams* sp, const double* grid_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                     const double* gc_c, int n_r, double gc_eps, void* ws, int taper, bool bucket) {
  if (taper != MIA_TAPER_GC && taper != MIA_TAPER_GC_INF) return MIA_ERR_SIZE;
  if (n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (!coord_group || !gc_c || !grid_xyz || !ws) return MIA_ERR_NULL;
  const IndexLayout L = index_layout(ws, P, n_coord);
  sp->grid = grid_xyz; sp->sxyz = L.sxyz; sp->hdr = L.hdr; sp->start = L.start; sp->sorted = L.sorted;
  if (bucket) { sp->sxyz = L.bxyz; sp->start = L.cursor; sp->sorted = L.bidx; }
  sp->nc = n_coord; sp->n_r = n_r;
  for (int c = 0; c < MIA_MAX_COORD; ++c) sp->group[c] = c < n_coord ? coord_group[c] : 0;
  for (int r = 0; r < MIA_MAX_RADII; ++r) { sp->inv_c[r] = r < n_r ? 1.0 / gc_c[r] : 1.0; sp->cc[r] = r < n_r ? gc_c[r] : 1.0; }
  sp->eps = gc_eps;
  sp->taper = taper;
  for (int r = 0; r < MIA_MAX_RADII; ++r) {
    sp->four_c2[r] = 4.0 * sp->cc[r] * sp->cc[r];
    sp->inv_c_f[r] = (float)sp->inv_c[r]; sp->c_f[r] = (float)sp->cc[r]; sp->inv_c2_f[r] = (float)(sp->inv_c[r] * sp->inv_c[r]);
  }
  sp->eps_f = (float)gc_eps;
  return MIA_OK;
}

int localize_impl(const double* grid_xyz, int64_t g0, int64_t g1, const double* obs_xyz, int64_t P, int n_coord,
                  const int32_t* coord_group, const double* gc_c, int n_r, double gc_eps, int p_cap,
                  int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w, int32_t* stats, void* ws, size_t ws_bytes,
                  hipStream_t stream, const PackJob* pack, bool stats_zeroed, const ZeroJob* zero, int taper,
                  bool header_clean, bool sort_cells) {
  if (g1 < g0 || g0 < 0 || P < 0) return MIA_ERR_SIZE;
  if (n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII || p_cap < 1) return MIA_ERR_SIZE;
  if (!coord_group || !gc_c || !stats) return MIA_ERR_NULL;
  for (int c = 0; c < n_coord; ++c) if (coord_group[c] < 0 || coord_group[c] >= n_r) return MIA_ERR_SIZE;
  for (int r = 0; r < n_r; ++r) if (!(gc_c[r] > 0.0)) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (!stats_zeroed) MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (ng == 0) return MIA_OK;
  if (!nbr_cnt || !nbr_idx || !nbr_w || !grid_xyz) return MIA_ERR_NULL;
  if (P == 0) {  // no observations at all: every list is empty (-> prior weights downstream)
    MIA_HIP_TRY(hipMemsetAsync(nbr_cnt, 0, ng * sizeof(int32_t), stream));
    MIA_HIP_TRY(hipMemsetAsync(nbr_idx, 0xff, ng * (size_t)p_cap * sizeof(int32_t), stream));
    MIA_HIP_TRY(hipMemsetAsync(nbr_w, 0, ng * (size_t)p_cap * sizeof(double), stream));
    return MIA_OK;
  }
#ifndef MIA_PACK_IN_LOCALIZE
#define MIA_PACK_IN_LOCALIZE 1
#endif
  int rc = index_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, ws, ws_bytes, stream, MIA_PACK_IN_LOCALIZE ? nullptr : pack,
                            zero, header_clean, sort_cells);
  if (rc != MIA_OK) return rc;
  return localize_lists_impl(grid_xyz, g0, g1, P, n_coord, coord_group, gc_c, n_r, gc_eps, p_cap, nbr_cnt, nbr_idx, nbr_w, stats, ws,
                             stream, MIA_PACK_IN_LOCALIZE ? pack : nullptr, taper);
}

// neighbour lists of grid points [g0, g1) over an index that already exists in `ws` (second half of localize_impl; the
// step driver's tile route calls it alone when declined points need per-point lists); P > 0
int localize_lists_impl(const double* grid_xyz, int64_t g0, int64_t g1, int64_t P, int n_coord, const int32_t* coord_group,
                        const double* gc_c, int n_r, double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx,
                        double* nbr_w, int32_t* stats, void* ws, hipStream_t stream, const PackJob* pack, int taper) {
  const int64_t ng = g1 - g0;
  if (ng <= 0 || P <= 0) return MIA_OK;
  int rc;
  LocalizeParams lp;
  rc = make_scan_params(&lp.scan, grid_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, ws, taper);
  if (rc != MIA_OK) return rc;
  lp.g0 = g0; lp.ng = ng; lp.p_cap = p_cap;
  lp.cnt = nbr_cnt; lp.idx = nbr_idx; lp.w = nbr_w; lp.stats = stats;
  lp.pack = PackJob{nullptr, nullptr, nullptr, 0, 0};
  lp.P = P;
  unsigned nb_pack = 0;
  size_t pack_lds = 0;
  if (pack && pack->rec) {
    lp.pack = *pack;
    nb_pack = (unsigned)((P + 63) / 64);
    pack_lds = (size_t)64 * (pack->kp + 1) * sizeof(float);      // (kp <= 132: 34 KB)
  }
  // one-wave workgroups: beside a bulk kernel that holds every wave slot (pipelined steps) a single freed slot is
  // enough to place one, whereas a 4-wave workgroup waited for four slots on one CU (200 us instead of 35)
  if (p_cap >= 64 && ng <= 2147483647LL && !MIA_EXP_FLAG("MIA_LOCALIZE_THREAD")) {   // long lists: one wavefront per grid point
    if (ng + nb_pack > 2147483647LL) return MIA_ERR_UNSUPPORTED;
    lp.nb_main = (unsigned)ng;
    localize_wave_kernel<<<dim3((unsigned)ng + nb_pack), dim3(64), pack_lds, stream>>>(lp);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  // four lanes per grid point: short lists only.  It trades 4x the wavefronts for a quarter of the serial candidate loop --
  // a gain while the kernel is latency-bound (C2, ~30 candidates per point: 33 -> 43 us beside the analysis kernel against
  // 66), a loss once the candidate loop itself is the work (C4 geometry, ~200 candidates: 316 us against 116)
  if (mia::option(MIA_OPT_LOCALIZE_QUAD) && p_cap <= 32) {
    const int64_t nbq = (ng + 15) / 16;
    if (nbq + nb_pack > 2147483647LL) return MIA_ERR_UNSUPPORTED;
    lp.nb_main = (unsigned)nbq;
    localize_quad_kernel<<<dim3((unsigned)nbq + nb_pack), dim3(64), pack_lds, stream>>>(lp);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  const int64_t nb = (ng + 63) / 64;
  if (nb + nb_pack > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  lp.nb_main = (unsigned)nb;
  localize_kernel<<<dim3((unsigned)nb + nb_pack), dim3(64), pack_lds, stream>>>(lp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

int sort_flagged_lists(const int32_t* flags, const int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w, int64_t ng, int p_cap,
                       void* ws, int64_t P, int n_coord, hipStream_t stream) {
  if (ng <= 0 || P <= 0) return MIA_OK;
  if (p_cap > 128) return MIA_ERR_UNSUPPORTED;
  const IndexLayout L = index_layout(ws, P, n_coord);
  SortListsParams sp{flags, nbr_cnt, nbr_idx, nbr_w, L.cell_of, ng, p_cap};
  const int64_t nb = ng < 4096 ? ng : 4096;
  sort_flagged_lists_kernel<<<dim3((unsigned)nb), dim3(64), 0, stream>>>(sp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_gaspari_cohn_f64(const double* r, int64_t n, double* w, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return taper_launch<double, MIA_TAPER_GC>(r, n, w, (hipStream_t)stream);
}
extern "C" int mia_gaspari_cohn_f32(const float* r, int64_t n, float* w, void* stream) {
  (void)hipGetLastError();
  return taper_launch<float, MIA_TAPER_GC>(r, n, w, (hipStream_t)stream);
}
extern "C" int mia_gaspari_cohn_inf_f64(const double* r, int64_t n, double* w, void* stream) {
  (void)hipGetLastError();
  return taper_launch<double, MIA_TAPER_GC_INF>(r, n, w, (hipStream_t)stream);
}
extern "C" int mia_gaspari_cohn_inf_f32(const float* r, int64_t n, float* w, void* stream) {
  (void)hipGetLastError();
  return taper_launch<float, MIA_TAPER_GC_INF>(r, n, w, (hipStream_t)stream);
}

extern "C" int mia_letkf_localize_workspace_bytes(int64_t P, int n_coord, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (P < 0 || n_coord < 1 || n_coord > MIA_MAX_COORD) return MIA_ERR_SIZE;
  if (P > 2000000000LL) return MIA_ERR_UNSUPPORTED;
  *bytes = index_layout(nullptr, P, n_coord).bytes;
  return MIA_OK;
}

extern "C" int mia_letkf_index_build_f64(const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                                         const double* gc_c, int n_r, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return index_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mia_letkf_localize_f64(const double* grid_xyz, int64_t g0, int64_t g1,
                                      const double* obs_xyz, int64_t P, int n_coord,
                                      const int32_t* coord_group, const double* gc_c, int n_r,
                                      double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx,
                                      double* nbr_w, int32_t* stats, void* ws, size_t ws_bytes,
                                      void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return mia::localize_impl(grid_xyz, g0, g1, obs_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, p_cap, nbr_cnt,
                            nbr_idx, nbr_w, stats, ws, ws_bytes, (hipStream_t)stream, nullptr, false, nullptr);
}

extern "C" int mia_letkf_localize_taper_f64(int taper, const double* grid_xyz, int64_t g0, int64_t g1,
                                            const double* obs_xyz, int64_t P, int n_coord,
                                            const int32_t* coord_group, const double* gc_c, int n_r,
                                            double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx,
                                            double* nbr_w, int32_t* stats, void* ws, size_t ws_bytes,
                                            void* stream) {
  (void)hipGetLastError();
  if (taper != MIA_TAPER_GC && taper != MIA_TAPER_GC_INF) return MIA_ERR_SIZE;
  return mia::localize_impl(grid_xyz, g0, g1, obs_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, p_cap, nbr_cnt,
                            nbr_idx, nbr_w, stats, ws, ws_bytes, (hipStream_t)stream, nullptr, false, nullptr, taper);
}

extern "C" int mia_letkf_localize_from_dist_f64(const double* dist, const int32_t* cand_idx,
                                                int64_t n_pts, int p_cap, const double* gc_c, int n_r,
                                                double gc_eps, int32_t* nbr_cnt, int32_t* nbr_idx,
                                                double* nbr_w, int32_t* stats, void* stream_) {
  return mia_letkf_localize_from_dist_taper_f64(MIA_TAPER_GC, dist, cand_idx, n_pts, p_cap, gc_c, n_r, gc_eps, nbr_cnt,
                                                nbr_idx, nbr_w, stats, stream_);
}

extern "C" int mia_letkf_localize_from_dist_taper_f64(int taper, const double* dist, const int32_t* cand_idx,
                                                      int64_t n_pts, int p_cap, const double* gc_c, int n_r,
                                                      double gc_eps, int32_t* nbr_cnt, int32_t* nbr_idx,
                                                      double* nbr_w, int32_t* stats, void* stream_) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  hipStream_t stream = (hipStream_t)stream_;
  if (taper != MIA_TAPER_GC && taper != MIA_TAPER_GC_INF) return MIA_ERR_SIZE;
  if (n_pts < 0 || p_cap < 1 || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (!gc_c || !stats) return MIA_ERR_NULL;
  for (int r = 0; r < n_r; ++r) if (!(gc_c[r] > 0.0)) return MIA_ERR_SIZE;
  MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (n_pts == 0) return MIA_OK;
  if (!dist || !cand_idx || !nbr_cnt || !nbr_idx || !nbr_w) return MIA_ERR_NULL;
  FromDistParams fp;
  fp.dist = dist; fp.cand = cand_idx; fp.n_pts = n_pts; fp.p_cap = p_cap; fp.n_r = n_r;
  for (int r = 0; r < MIA_MAX_RADII; ++r) fp.inv_c[r] = r < n_r ? 1.0 / gc_c[r] : 1.0;
  fp.eps = gc_eps; fp.taper = taper; fp.cnt = nbr_cnt; fp.idx = nbr_idx; fp.w = nbr_w; fp.stats = stats;
  const int64_t nb = (n_pts + 3) / 4;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  localize_from_dist_kernel<<<dim3((unsigned)nb), dim3(256), 0, stream>>>(fp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

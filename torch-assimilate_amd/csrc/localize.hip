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
#include "mia_common.h"

namespace mia {

struct IndexHeader {        // lives at the start of the workspace, written on device
  double mn[MIA_MAX_COORD];
  double invh[MIA_MAX_COORD];
  int n[MIA_MAX_COORD];
  int ncell;
};

struct IndexParams {
  const double* obs;  // [P][nc]
  int64_t P;
  int nc;
  double cutoff[MIA_MAX_COORD];  // minimal cell edge per coordinate (= 2 * c of its group)
  int cell_cap;
  IndexHeader* hdr;
  int* start;    // [cell_cap + 1]  counts -> exclusive starts
  int* cursor;   // [cell_cap]
  int* sorted;   // [P]
  int* cell_of;  // [P]
};

__device__ inline int cell_coord(double x, double mn, double invh, int n) {
  double f = floor((x - mn) * invh);
  f = f < -2.0 ? -2.0 : f;
  f = f > double(n) + 1.0 ? double(n) + 1.0 : f;
  return (f == f) ? int(f) : -2;  // NaN coordinate -> no cell
}

// one workgroup: bounding box of the observations, then the cell grid dimensions
__global__ __launch_bounds__(1024) void index_bbox_kernel(IndexParams p) {
  __shared__ double smn[MIA_MAX_COORD][16], smx[MIA_MAX_COORD][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int c = 0; c < p.nc; ++c) {
    double mn = 1e300, mx = -1e300;
    for (int64_t j = tid; j < p.P; j += blockDim.x) {
      double x = p.obs[j * p.nc + c];
      if (x == x) { mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      double a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
      mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    if (lane == 0) { smn[c][wv] = mn; smx[c][wv] = mx; }
  }
  __syncthreads();
  if (tid == 0) {
    double ext[MIA_MAX_COORD], h[MIA_MAX_COORD];
    long long n[MIA_MAX_COORD];
    for (int c = 0; c < MIA_MAX_COORD; ++c) { n[c] = 1; h[c] = 1.0; ext[c] = 0.0; p.hdr->mn[c] = 0.0; }
    for (int c = 0; c < p.nc; ++c) {
      double mn = 1e300, mx = -1e300;
      for (int w = 0; w < 16; ++w) { mn = smn[c][w] < mn ? smn[c][w] : mn; mx = smx[c][w] > mx ? smx[c][w] : mx; }
      if (mn > mx) { mn = 0.0; mx = 0.0; }
      p.hdr->mn[c] = mn;
      ext[c] = mx - mn;
      h[c] = p.cutoff[c] > 0.0 ? p.cutoff[c] : 1.0;
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

__global__ void index_count_kernel(IndexParams p) {
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= p.P) return;
  int c = obs_cell(p.hdr, p.obs + j * p.nc, p.nc);
  p.cell_of[j] = c;
  atomicAdd(&p.start[c], 1);
}

// one workgroup: in-place exclusive scan of start[0 .. ncell]
__global__ __launch_bounds__(1024) void index_scan_kernel(IndexParams p) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = p.hdr->ncell + 1;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = (i < n - 1) ? p.start[i] : 0;   // entry ncell holds the total
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (i < n) p.start[i] = off + x - v;
    __syncthreads();
    if (tid == 1023) carry = off + x;
    __syncthreads();
  }
}

__global__ void index_scatter_kernel(IndexParams p) {
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= p.P) return;
  int c = p.cell_of[j];
  int pos = p.start[c] + atomicAdd(&p.cursor[c], 1);
  p.sorted[pos] = int(j);
}

// the scatter order inside a cell depends on atomic arrival: sort each cell's slice by
// observation index so that neighbour lists (and therefore summation order) are reproducible
__global__ void index_sortcell_kernel(IndexParams p) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.hdr->ncell) return;
  int lo = p.start[c], hi = p.start[c + 1];
  for (int i = lo + 1; i < hi; ++i) {
    int v = p.sorted[i], j = i - 1;
    while (j >= lo && p.sorted[j] > v) { p.sorted[j + 1] = p.sorted[j]; --j; }
    p.sorted[j + 1] = v;
  }
}

struct LocalizeParams {
  const double* grid;  // [G][nc]
  const double* obs;   // [P][nc]
  int64_t g0, ng, P;
  int nc, n_r;
  int group[MIA_MAX_COORD];
  double inv_c[MIA_MAX_RADII];
  double eps;
  int p_cap;
  const IndexHeader* hdr;
  const int* start;
  const int* sorted;
  int* cnt; int* idx; double* w; int* stats;
};

// one wavefront per grid point
__global__ __launch_bounds__(256) void localize_kernel(LocalizeParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t pt = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pt >= p.ng) return;   // whole wave leaves together
  const IndexHeader* h = p.hdr;
  double gx[MIA_MAX_COORD];
  int cg[MIA_MAX_COORD];
  for (int c = 0; c < MIA_MAX_COORD; ++c) { gx[c] = 0.0; cg[c] = 0; }
  for (int c = 0; c < p.nc; ++c) {
    gx[c] = p.grid[(p.g0 + pt) * p.nc + c];
    cg[c] = cell_coord(gx[c], h->mn[c], h->invh[c], h->n[c]);
  }
  int* my_idx = p.idx + pt * p.p_cap;
  double* my_w = p.w + pt * p.p_cap;
  int count = 0;
  // outer coordinates: -1..+1 each; the innermost (fastest) coordinate's three cells are
  // contiguous in the table and are scanned as one range
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
          double dx = p.obs[(int64_t)j * nc + c] - gx[c];
          d2[p.group[c]] += dx * dx;
        }
        wgt = 1.0;
        for (int r = 0; r < p.n_r; ++r) wgt *= gc_taper<double>(sqrt(d2[r]) * p.inv_c[r]);
        use = wgt > p.eps;
      }
      const unsigned long long mask = __ballot(use);
      if (use) {
        const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < p.p_cap) { my_idx[slot] = j; my_w[slot] = sqrt(wgt); }
      }
      count += __popcll(mask);
    }
  }
  for (int s = count + lane; s < p.p_cap; s += 64) { my_idx[s] = -1; my_w[s] = 0.0; }
  if (lane == 0) {
    p.cnt[pt] = count;
    // same-address atomics serialise at ~11 ns each (1e5 points -> 1 ms): the running maximum only
    // grows, so a relaxed look first lets almost every wavefront skip the atomic
    if (count > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], count);
    if (count > p.p_cap) atomicAdd(&p.stats[1], 1);
  }
}

struct FromDistParams {
  const double* dist; const int* cand; int64_t n_pts; int p_cap; int n_r;
  double inv_c[MIA_MAX_RADII]; double eps;
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
        for (int r = 0; r < p.n_r; ++r)
          wgt *= gc_taper<double>(p.dist[((int64_t)r * p.n_pts + pt) * p.p_cap + s] * p.inv_c[r]);
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

template <typename T>
__global__ void gc_kernel(const T* r, int64_t n, T* w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) w[i] = gc_taper<T>(r[i]);
}

static int cell_cap_for(int64_t P) {
  int64_t cap = 2 * P;
  if (cap < 1024) cap = 1024;
  if (cap > (int64_t)1 << 24) cap = (int64_t)1 << 24;
  return int(cap);
}

}  // namespace mia

using namespace mia;

extern "C" int mia_gaspari_cohn_f64(const double* r, int64_t n, double* w, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  if (n < 0) return MIA_ERR_SIZE;
  if (n == 0) return MIA_OK;
  if (!r || !w) return MIA_ERR_NULL;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  gc_kernel<double><<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(r, n, w);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_gaspari_cohn_f32(const float* r, int64_t n, float* w, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  if (n < 0) return MIA_ERR_SIZE;
  if (n == 0) return MIA_OK;
  if (!r || !w) return MIA_ERR_NULL;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  gc_kernel<float><<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(r, n, w);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_letkf_localize_workspace_bytes(int64_t P, int n_coord, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (P < 0 || n_coord < 1 || n_coord > MIA_MAX_COORD) return MIA_ERR_SIZE;
  if (P > 2000000000LL) return MIA_ERR_UNSUPPORTED;
  const size_t cap = (size_t)cell_cap_for(P);
  size_t b = align_up(sizeof(IndexHeader), 256);
  b += align_up((cap + 1) * sizeof(int), 256);  // start
  b += align_up(cap * sizeof(int), 256);        // cursor
  b += align_up((size_t)P * sizeof(int) + 4, 256);  // sorted
  b += align_up((size_t)P * sizeof(int) + 4, 256);  // cell_of
  *bytes = b;
  return MIA_OK;
}

extern "C" int mia_letkf_localize_f64(const double* grid_xyz, int64_t g0, int64_t g1,
                                      const double* obs_xyz, int64_t P, int n_coord,
                                      const int32_t* coord_group, const double* gc_c, int n_r,
                                      double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx,
                                      double* nbr_w, int32_t* stats, void* ws, size_t ws_bytes,
                                      void* stream_) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  hipStream_t stream = (hipStream_t)stream_;
  if (g1 < g0 || g0 < 0 || P < 0) return MIA_ERR_SIZE;
  if (n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII || p_cap < 1) return MIA_ERR_SIZE;
  if (!coord_group || !gc_c || !stats) return MIA_ERR_NULL;
  for (int c = 0; c < n_coord; ++c) if (coord_group[c] < 0 || coord_group[c] >= n_r) return MIA_ERR_SIZE;
  for (int r = 0; r < n_r; ++r) if (!(gc_c[r] > 0.0)) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (ng == 0) return MIA_OK;
  if (!nbr_cnt || !nbr_idx || !nbr_w || !grid_xyz) return MIA_ERR_NULL;
  if (P == 0) {  // no observations at all: every list is empty (-> prior weights downstream)
    MIA_HIP_TRY(hipMemsetAsync(nbr_cnt, 0, ng * sizeof(int32_t), stream));
    MIA_HIP_TRY(hipMemsetAsync(nbr_idx, 0xff, ng * (size_t)p_cap * sizeof(int32_t), stream));
    MIA_HIP_TRY(hipMemsetAsync(nbr_w, 0, ng * (size_t)p_cap * sizeof(double), stream));
    return MIA_OK;
  }
  if (!obs_xyz || !ws) return MIA_ERR_NULL;
  size_t need = 0;
  int rc = mia_letkf_localize_workspace_bytes(P, n_coord, &need);
  if (rc != MIA_OK) return rc;
  if (ws_bytes < need) return MIA_ERR_WORKSPACE;
  if (((uintptr_t)ws) & 255) return MIA_ERR_ALIGN;

  const size_t cap = (size_t)cell_cap_for(P);
  IndexParams ip;
  ip.obs = obs_xyz; ip.P = P; ip.nc = n_coord; ip.cell_cap = (int)cap;
  for (int c = 0; c < MIA_MAX_COORD; ++c) ip.cutoff[c] = c < n_coord ? 2.0 * gc_c[coord_group[c]] : 1.0;
  char* base = (char*)ws;
  ip.hdr = (IndexHeader*)base; base += align_up(sizeof(IndexHeader), 256);
  ip.start = (int*)base; base += align_up((cap + 1) * sizeof(int), 256);
  ip.cursor = (int*)base; base += align_up(cap * sizeof(int), 256);
  ip.sorted = (int*)base; base += align_up((size_t)P * sizeof(int) + 4, 256);
  ip.cell_of = (int*)base;
  // start and cursor are adjacent: one memset clears both
  MIA_HIP_TRY(hipMemsetAsync(ip.start, 0, (char*)ip.sorted - (char*)ip.start, stream));
  index_bbox_kernel<<<dim3(1), dim3(1024), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  const unsigned nbP = (unsigned)((P + 255) / 256);
  index_count_kernel<<<dim3(nbP), dim3(256), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  index_scan_kernel<<<dim3(1), dim3(1024), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  index_scatter_kernel<<<dim3(nbP), dim3(256), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();
  index_sortcell_kernel<<<dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream>>>(ip);
  MIA_LAUNCH_CHECK();

  LocalizeParams lp;
  lp.grid = grid_xyz; lp.obs = obs_xyz; lp.g0 = g0; lp.ng = ng; lp.P = P; lp.nc = n_coord; lp.n_r = n_r;
  for (int c = 0; c < MIA_MAX_COORD; ++c) lp.group[c] = c < n_coord ? coord_group[c] : 0;
  for (int r = 0; r < MIA_MAX_RADII; ++r) lp.inv_c[r] = r < n_r ? 1.0 / gc_c[r] : 1.0;
  lp.eps = gc_eps; lp.p_cap = p_cap; lp.hdr = ip.hdr; lp.start = ip.start; lp.sorted = ip.sorted;
  lp.cnt = nbr_cnt; lp.idx = nbr_idx; lp.w = nbr_w; lp.stats = stats;
  const int64_t nb = (ng + 3) / 4;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  localize_kernel<<<dim3((unsigned)nb), dim3(256), 0, stream>>>(lp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_letkf_localize_from_dist_f64(const double* dist, const int32_t* cand_idx,
                                                int64_t n_pts, int p_cap, const double* gc_c, int n_r,
                                                double gc_eps, int32_t* nbr_cnt, int32_t* nbr_idx,
                                                double* nbr_w, int32_t* stats, void* stream_) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  hipStream_t stream = (hipStream_t)stream_;
  if (n_pts < 0 || p_cap < 1 || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (!gc_c || !stats) return MIA_ERR_NULL;
  for (int r = 0; r < n_r; ++r) if (!(gc_c[r] > 0.0)) return MIA_ERR_SIZE;
  MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (n_pts == 0) return MIA_OK;
  if (!dist || !cand_idx || !nbr_cnt || !nbr_idx || !nbr_w) return MIA_ERR_NULL;
  FromDistParams fp;
  fp.dist = dist; fp.cand = cand_idx; fp.n_pts = n_pts; fp.p_cap = p_cap; fp.n_r = n_r;
  for (int r = 0; r < MIA_MAX_RADII; ++r) fp.inv_c[r] = r < n_r ? 1.0 / gc_c[r] : 1.0;
  fp.eps = gc_eps; fp.cnt = nbr_cnt; fp.idx = nbr_idx; fp.w = nbr_w; fp.stats = stats;
  const int64_t nb = (n_pts + 3) / 4;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  localize_from_dist_kernel<<<dim3((unsigned)nb), dim3(256), 0, stream>>>(fp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Tile-shaped Gaspari-Cohn lists and split-precision observation records (formats: mia_tiles.h).
//
// localize_tiles_kernel: ONE wavefront per tile of 16 consecutive grid points.  It stands in for 16 evaluations of
// GaspariCohn.localize_obs (pytassim/localization/gaspari_cohn.py:97-136) + the mask / sqrt(rho) of
// wrapper_localization (pytassim/interface/wrapper.py:88-97) and writes what the analysis wave needs in the layout of
// its registers: the union of the 16 lists (observation indices by rank) and the 16 x U matrix of sqrt(rho).
// Candidate-major: the candidates of a tile are the observations of the cell box around its points (every point's 3^d
// neighbourhood lies inside), each is loaded ONCE and weighed against the 16 points in float64 -- same distance, taper and
// `w > eps` arithmetic as the per-point kernels of localize.hip, so the masks are the reference's.  Three dependent memory
// round trips per tile (coordinates -> cell starts -> candidates) where the per-point kernels take six per point.
//
// pack_split_wave: [k][P] perturbations + d[P] -> per-observation split records (scaled hi | lo halves), 64 per wavefront.
#include "mia_common.h"
#include "mia_options.h"
#include "mia_localize_dev.h"
#include "mia_tiles.h"
#include "mia_tile_localize.h"

namespace mia {

struct SplitPackParams { SplitPackJob job; int64_t P; };
__global__ __launch_bounds__(64) void pack_split_kernel(SplitPackParams p) {
  extern __shared__ __attribute__((aligned(16))) float pk_lds[];
  pack_split_wave(p.job, p.P, (int64_t)blockIdx.x, pk_lds);
}


int split_pack_launch(const float* Yb, const float* d, int k, int64_t P, void* rec, hipStream_t stream) {
  if (k < 1 || k > 1024 || P < 0) return MIA_ERR_SIZE;
  if (!rec || (P > 0 && (!Yb || !d))) return MIA_ERR_NULL;
  const int64_t nb = (P + 1 + 63) / 64;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  const size_t lds = split_pack_lds(k);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  SplitPackParams sp{{Yb, d, (unsigned char*)rec, k}, P};
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)pack_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  pack_split_kernel<<<dim3((unsigned)nb), dim3(64), lds, stream>>>(sp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- tile lists ------------------------------------------------------------------------------------------------------
struct TileLocParams {
  ScanParams scan;
  int64_t g0, ng;
  int ut;
  int4* hdr; int32_t* uidx; f4w* D;
  int32_t* stats;
  unsigned nb_main;          // workgroups nb_main, nb_main + 1, ... pack split records (independent passenger)
  SplitPackJob pack;
  int64_t P;
};

// The localisation itself: tile_localize<BUCKET, NC, TAPER> (mia_tile_localize.h), shared with the analysis kernel's fused variant.
template <bool BUCKET, int NC, int TAPER>
__global__ __launch_bounds__(64) void localize_tiles_kernel(TileLocParams p) {
  MIA_PREP_PRIORITY();
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_lds[];
  if (blockIdx.x >= p.nb_main) {
    pack_split_wave(p.pack, p.P, (int64_t)(blockIdx.x - p.nb_main), reinterpret_cast<float*>(tl_lds));
    return;
  }
  const int lane = threadIdx.x, cl = lane & 15, pg = lane >> 4;
  const int UMAX = 16 * p.ut;
  const int64_t tile = blockIdx.x;
  const TileLocOut r = tile_localize<BUCKET, NC, TAPER>(p.scan, p.g0, p.ng, p.ut, tile, tl_lds, lane);
  const TileLocLds L(tl_lds);
  const bool overflow = r.overflow;
  for (int s = lane; s < UMAX; s += 64) {
    const int u = overflow ? -1 : L.uinv[s];
    p.uidx[tile * UMAX + s] = u < 0 ? -1 : L.ukey[u];
  }
  for (int t = 0; t < p.ut; ++t) {
    f4w v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int u = overflow ? -1 : L.uinv[16 * t + 4 * pg + qq];
      v[qq] = u < 0 ? 0.0f : L.Wt[u * 16 + cl];
    }
    p.D[(tile * p.ut + t) * 64 + lane] = v;
  }
  if (lane == 0) {
    p.hdr[tile] = make_int4(overflow ? -1 : r.U, r.longest, r.npts, 0);
    if (r.longest > __hip_atomic_load(&p.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.stats[0], r.longest);
    if (overflow) atomicAdd(&p.stats[1], 1);
    // (more slots cannot help a box that is too large: marked, so that the caller goes to per-point lists at once)
    if (r.box_overflow) atomicOr(&p.stats[1], MIA_TILE_BOX_OVERFLOW);
  }
}

int tile_lists_launch(const double* grid_xyz, int64_t g0, int64_t ng, int64_t P, int n_coord, const int32_t* coord_group,
                      const double* gc_c, int n_r, double gc_eps, int taper, int ut, void* tile_lists, int32_t* stats,
                      void* index_ws, hipStream_t stream, const SplitPackJob* pack, bool bucket, const int* counts) {
  if (ng < 0 || P < 0 || ut < 1 || ut > 6) return MIA_ERR_SIZE;
  if (!tile_lists || !stats) return MIA_ERR_NULL;
  const TileListLayout L = tile_list_layout(ng, ut);
  unsigned nb_pack = 0;
  size_t lds = tile_loc_lds(ut);
  TileLocParams tp;
  tp.pack = SplitPackJob{nullptr, nullptr, nullptr, 0};
  if (pack && pack->rec) {
    tp.pack = *pack;
    if ((P + 1 + 63) / 64 > 2000000000LL) return MIA_ERR_UNSUPPORTED;
    nb_pack = (unsigned)((P + 1 + 63) / 64);
    const size_t pl = split_pack_lds(pack->k);
    lds = pl > lds ? pl : lds;
  }
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  char* base = (char*)tile_lists;
  tp.hdr = (int4*)(base + L.hdr); tp.uidx = (int32_t*)(base + L.idx); tp.D = (f4w*)(base + L.D);
  tp.stats = stats; tp.g0 = g0; tp.ng = ng; tp.ut = ut; tp.P = P;
  if (L.ntile + nb_pack > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  if (L.ntile + nb_pack == 0) return MIA_OK;
  if (P == 0) {     // no observations at all: every tile is empty (-> prior weights downstream); no index exists
    MIA_HIP_TRY(hipMemsetAsync(base + L.hdr, 0, (size_t)L.ntile * 16, stream));
    MIA_HIP_TRY(hipMemsetAsync(base + L.idx, 0xff, (size_t)L.ntile * 16 * ut * sizeof(int32_t), stream));
    MIA_HIP_TRY(hipMemsetAsync(base + L.D, 0, (size_t)L.ntile * ut * 1024, stream));
    if (nb_pack) return split_pack_launch(pack->Yb, pack->d, pack->k, P, pack->rec, stream);
    return MIA_OK;
  }
  if (!grid_xyz || !index_ws) return MIA_ERR_NULL;
  int rc = make_scan_params(&tp.scan, grid_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, index_ws, taper, bucket);
  if (rc != MIA_OK) return rc;
  if (bucket && counts) tp.scan.start = counts;      // (which of the layout's two per-cell count arrays this step's build filled)
  tp.nb_main = (unsigned)L.ntile;
  void (*kern)(TileLocParams) = nullptr;
#define MIA_TL_PICK(B, T)                                                                                                  \
  kern = n_coord == 1 ? localize_tiles_kernel<B, 1, T> : (n_coord == 2 ? localize_tiles_kernel<B, 2, T> : localize_tiles_kernel<B, 3, T>)
  if (bucket) { if (taper == MIA_TAPER_GC_INF) MIA_TL_PICK(true, MIA_TAPER_GC_INF); else MIA_TL_PICK(true, MIA_TAPER_GC); }
  else { if (taper == MIA_TAPER_GC_INF) MIA_TL_PICK(false, MIA_TAPER_GC_INF); else MIA_TL_PICK(false, MIA_TAPER_GC); }
#undef MIA_TL_PICK
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<dim3((unsigned)(L.ntile + nb_pack)), dim3(64), lds, stream>>>(tp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_letkf_split_record_bytes(int k, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (k < 1 || k > 1024) return MIA_ERR_SIZE;
  *bytes = (size_t)split_rec_bytes(k);
  return MIA_OK;
}

extern "C" int mia_letkf_pack_split_f32(const float* Yb, const float* d, int k, int64_t P, void* rec, void* stream) {
  (void)hipGetLastError();
  return split_pack_launch(Yb, d, k, P, rec, (hipStream_t)stream);
}

extern "C" int mia_letkf_tile_lists_bytes(int64_t n_points, int p_max, int extra_blocks, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (n_points < 0 || p_max < 0 || extra_blocks < 0) return MIA_ERR_SIZE;
  if (tile_ut_for(p_max) + extra_blocks > 6) return MIA_ERR_UNSUPPORTED;
  *bytes = tile_list_layout(n_points, tile_ut_for(p_max) + extra_blocks).bytes;
  return MIA_OK;
}

extern "C" int mia_letkf_localize_tiles_f64(int taper, const double* grid_xyz, int64_t g0, int64_t g1,
                                            const double* obs_xyz, int64_t P, int n_coord, const int32_t* coord_group,
                                            const double* gc_c, int n_r, double gc_eps, int p_max, int extra_blocks,
                                            void* tile_lists, size_t tile_lists_bytes, int32_t* stats, void* ws, size_t ws_bytes,
                                            void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (taper != MIA_TAPER_GC && taper != MIA_TAPER_GC_INF) return MIA_ERR_SIZE;
  if (g1 < g0 || g0 < 0 || P < 0 || p_max < 0 || extra_blocks < 0) return MIA_ERR_SIZE;
  if (n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  if (!coord_group || !gc_c || !stats || !tile_lists) return MIA_ERR_NULL;
  const int ut = tile_ut_for(p_max) + extra_blocks;
  if (ut > 6) return MIA_ERR_UNSUPPORTED;
  if (tile_lists_bytes < tile_list_layout(g1 - g0, ut).bytes) return MIA_ERR_WORKSPACE;
  MIA_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int32_t), stream));
  if (P > 0) {
    int rc = index_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, ws, ws_bytes, stream, nullptr, nullptr, false, false);
    if (rc != MIA_OK) return rc;
  }
  return tile_lists_launch(grid_xyz, g0, g1 - g0, P, n_coord, coord_group, gc_c, n_r, gc_eps, taper, ut, tile_lists, stats, ws,
                           stream, nullptr, false);
}

// Fused localisation + LETKF analysis: letkf_tile2_kernel's body (letkf_tile2_kernel.h) with LOC = number of coordinates -- every
// wavefront first localises its own tile of sixteen grid points over the step's bucket index (tile_localize, the list kernel's
// code: same union, same ranks, same sqrt(rho), so the analysis is bit for bit the one from lists in memory), then analyses it.
// Reference: GaspariCohn.localize_obs (pytassim/localization/gaspari_cohn.py:97-136) + wrapper_localization
// (pytassim/interface/wrapper.py:86-98) + the ETKF weights and transform (pytassim/core/etkf.py:57-103), per grid point.
// What it removes from a step: the list kernel's launch, 2.5 KB of lists per tile written and read back, and one dependent
// memory round trip of the analysis prologue (header -> slot table).  Shapes: unions of at most 32 slots (larger
// unions go through lists to the two-wavefronts-per-tile kernel); any number of state rows.
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"
#include "letkf_tile2_kernel.h"

namespace mia {

struct Tile2FParams { Tile2Params t; Tile2Loc loc; };

template <int UT, int KT, int NC, bool MROWS, int WAVES>
__global__ __launch_bounds__(64, WAVES)
void letkf_tile2f_kernel(Tile2FParams PF) { tile2_body<UT, KT, MROWS, NC>(PF.t, &PF.loc, (int64_t)blockIdx.y * gridDim.x + blockIdx.x); }

#ifdef MIA_TILE_STAMPS
extern "C" int mia_debug_tile2f_stamps(long long* host, int n_tiles) {
  if (n_tiles > kT2StampTiles) n_tiles = kT2StampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile2_stamps), sizeof(long long) * kT2StampN * (size_t)n_tiles);
}
#endif

// ---- several steps in ONE launch (launch coalescing, csrc/sharded_step.hip).  A launch of 6250 tiles spends 9.6 us filling and
// draining the chip around 26 us of work at full occupancy (tools/fused_scaling.py: 31.2 / 39.4 / 53.0 us for 5120 / 6250 / 10240
// tiles); kernels of different streams overlap badly (three concurrent ones: 63 us per step, profiles/r05_hwq_ab.txt).  Steps in
// flight are independent, so the step driver's launch thread hands the tiles of up to kT2fBatchMax ready steps to one grid: block b
// belongs to the step whose range of blocks holds it and takes THAT step's parameters -- read from the kernel-argument segment at a
// computed offset (scalar loads; indexing the by-value argument would copy it to scratch).  Same code per tile, same bits.
struct Tile2FBatch { int n; int first[kT2fBatchMax + 1]; Tile2FParams s[kT2fBatchMax]; };

template <int UT, int KT, int NC, bool MROWS, int WAVES>
__global__ __launch_bounds__(64, WAVES)
void letkf_tile2fb_kernel(Tile2FBatch B) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the stub)
  const auto* kb = (const __attribute__((address_space(4))) Tile2FBatch*)__builtin_amdgcn_kernarg_segment_ptr();
  const int bid = (int)((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
  int sel = 0;
#pragma unroll
  for (int i = 1; i < kT2fBatchMax; ++i) sel += (i < kb->n && bid >= kb->first[i]) ? 1 : 0;
  const Tile2FParams PF = kb->s[sel];
  tile2_body<UT, KT, MROWS, NC>(PF.t, &PF.loc, (int64_t)(bid - kb->first[sel]));
#else
  (void)B;
#endif
}

// the collector of the launch thread: while one is set (tile2f_batch_collect), tile2f_launch stores the step's parameters and its
// start / stop events instead of launching
struct Tile2fCollector {
  int n = 0, ut = 0, kt = 0, nc = 0, k = 0;
  bool mrows = false;
  Tile2FParams s[kT2fBatchMax];
  hipEvent_t start[kT2fBatchMax], stop[kT2fBatchMax];
};
static thread_local Tile2fCollector* t_collect = nullptr;

bool tile2f_covers(int m, int k, int ut, int n_coord) {
  const int kt = (k + 15) >> 4;
  return m >= 1 && ut >= 1 && ut <= 2 && kt >= 1 && kt <= 6 && n_coord >= 1 && n_coord <= MIA_MAX_COORD;
}

template <int UT, int KT, int NC, bool MROWS>
static int tile2f_launch_m(const Tile2FParams& pf, hipStream_t stream) {
  const size_t a = tile2_lds_bytes(UT, pf.t.k), b = tile_loc_lds(UT);
  const size_t lds = a > b ? a : b;
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  if (Tile2fCollector* c = t_collect) {
    // collected, not launched: the same instantiation and ensemble size as what the collector holds, room left, a grid that fits
    const int64_t nt = (pf.t.ng + 15) >> 4;
    const bool fits = !MROWS && c->n < kT2fBatchMax && nt < ((int64_t)1 << 24) &&      // (one state row: the instantiations that exist)
                      (c->n == 0 || (c->ut == UT && c->kt == KT && c->nc == NC && c->mrows == MROWS && c->k == pf.t.k));
    if (fits) {
      c->ut = UT; c->kt = KT; c->nc = NC; c->mrows = MROWS; c->k = pf.t.k;
      c->s[c->n] = pf;
      c->start[c->n] = launch_start_event(); c->stop[c->n] = launch_stop_event();
      launch_start_event() = nullptr; launch_stop_event() = nullptr;      // taken
      ++c->n;
      ++tile_launch_count();
      return MIA_OK;
    }
    t_collect = nullptr;      // (does not fit: this step is launched on its own, and the collector takes no more)
  }
  auto kern = letkf_tile2f_kernel<UT, KT, NC, MROWS, MROWS ? 2 : MIA_TILE2_WAVES_UT2>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntile = (pf.t.ng + 15) >> 4;
  const int64_t gx = ntile < 65536 ? ntile : 65536;
  const int64_t gy = (ntile + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  hipEvent_t& stop = launch_stop_event();
  if (stop) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64), (unsigned)lds, stream, launch_start_event(), stop, 0, pf);
    stop = nullptr;        // taken
    launch_start_event() = nullptr;
  } else {
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64), lds, stream>>>(pf);
  }
  ++tile_launch_count();
  note_analysis_kernel("letkf_tile2f_kernel<%d, %d, %d, %s, %d>", UT, KT, NC, MROWS ? "true" : "false", MROWS ? 2 : MIA_TILE2_WAVES_UT2);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <int UT, int KT, int NC>
static int tile2f_launch_n(const Tile2FParams& pf, hipStream_t stream) {        // one state row: straight-line code; more: the row loop
  return pf.t.m == 1 ? tile2f_launch_m<UT, KT, NC, false>(pf, stream) : tile2f_launch_m<UT, KT, NC, true>(pf, stream);
}

template <int UT, int KT>
static int tile2f_launch_k(const Tile2FParams& pf, hipStream_t stream) {
  switch (pf.loc.scan.nc) {
    case 1: return tile2f_launch_n<UT, KT, 1>(pf, stream);
    case 2: return tile2f_launch_n<UT, KT, 2>(pf, stream);
    case 3: return tile2f_launch_n<UT, KT, 3>(pf, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

template <int UT>
static int tile2f_launch_u(const Tile2FParams& pf, int kt, hipStream_t stream) {
  switch (kt) {
    case 1: return tile2f_launch_k<UT, 1>(pf, stream);
    case 2: return tile2f_launch_k<UT, 2>(pf, stream);
    case 3: return tile2f_launch_k<UT, 3>(pf, stream);
    case 4: return tile2f_launch_k<UT, 4>(pf, stream);
    case 5: return tile2f_launch_k<UT, 5>(pf, stream);
    case 6: return tile2f_launch_k<UT, 6>(pf, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

template <int UT, int KT, int NC, bool MROWS>
static int tile2f_batch_launch_m(const Tile2fCollector& c, hipStream_t stream) {
  Tile2FBatch B;
  B.n = c.n;
  int64_t tot = 0;
  for (int i = 0; i < kT2fBatchMax; ++i) {
    B.first[i] = (int)tot;
    if (i < c.n) { B.s[i] = c.s[i]; tot += (c.s[i].t.ng + 15) >> 4; } else B.s[i] = c.s[0];
  }
  B.first[kT2fBatchMax] = (int)tot;
  const size_t a = tile2_lds_bytes(UT, c.k), b = tile_loc_lds(UT);
  const size_t lds = a > b ? a : b;
  auto kern = letkf_tile2fb_kernel<UT, KT, NC, MROWS, MROWS ? 2 : MIA_TILE2_WAVES_UT2>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t gx = tot < 65536 ? tot : 65536;
  const int64_t gy = (tot + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  // the first step's events ride in the dispatch packet (a timed step always opens its launch); the others' completion events are
  // recorded behind the launch
  hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64), (unsigned)lds, stream, c.start[0], c.stop[0], 0, B);
  note_analysis_kernel("letkf_tile2fb_kernel<%d, %d, %d, %s, %d>", UT, KT, NC, MROWS ? "true" : "false", MROWS ? 2 : MIA_TILE2_WAVES_UT2);
  MIA_LAUNCH_CHECK();
  for (int i = 1; i < c.n; ++i)
    if (c.stop[i]) MIA_HIP_TRY(hipEventRecord(c.stop[i], stream));
  return MIA_OK;
}

template <int UT, int KT, int NC>
static int tile2f_batch_launch_n(const Tile2fCollector& c, hipStream_t stream) {
  return c.mrows ? MIA_ERR_UNSUPPORTED : tile2f_batch_launch_m<UT, KT, NC, false>(c, stream);
}
template <int UT, int KT>
static int tile2f_batch_launch_k(const Tile2fCollector& c, hipStream_t stream) {
  switch (c.nc) {
    case 1: return tile2f_batch_launch_n<UT, KT, 1>(c, stream);
    case 2: return tile2f_batch_launch_n<UT, KT, 2>(c, stream);
    case 3: return tile2f_batch_launch_n<UT, KT, 3>(c, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}
template <int UT>
static int tile2f_batch_launch_u(const Tile2fCollector& c, hipStream_t stream) {
  switch (c.kt) {
    case 1: return tile2f_batch_launch_k<UT, 1>(c, stream);
    case 2: return tile2f_batch_launch_k<UT, 2>(c, stream);
    case 3: return tile2f_batch_launch_k<UT, 3>(c, stream);
    case 4: return tile2f_batch_launch_k<UT, 4>(c, stream);
    case 5: return tile2f_batch_launch_k<UT, 5>(c, stream);
    case 6: return tile2f_batch_launch_k<UT, 6>(c, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

// launch-thread interface (sharded_step.hip): begin collecting on this thread; the steps collected so far; launch them as one grid
// and stop collecting
static thread_local Tile2fCollector t_collector_storage;
void tile2f_collect_begin() { t_collector_storage.n = 0; t_collect = &t_collector_storage; }
int tile2f_collected() { return t_collector_storage.n; }
bool tile2f_collecting() { return t_collect != nullptr; }
int tile2f_collect_launch(hipStream_t stream) {
  t_collect = nullptr;
  Tile2fCollector& c = t_collector_storage;
  if (c.n == 0) return MIA_OK;
  // (one step too: the launch thread's steps all run under ONE kernel name, which is what a profile of steps in flight averages over)
  const int rc = c.ut == 1 ? tile2f_batch_launch_u<1>(c, stream) : tile2f_batch_launch_u<2>(c, stream);
  c.n = 0;
  return rc;
}

int tile2f_launch(const Tile2Params& tp, const Tile2Loc& loc, int ut, int kt, hipStream_t stream) {
  if (!tile2f_covers(tp.m, tp.k, ut, loc.scan.nc) || !loc.stats) return MIA_ERR_UNSUPPORTED;
  Tile2FParams pf;
  pf.t = tp;
  pf.loc = loc;
#ifdef MIA_TILE2_SINGLE        // (development builds: one instantiation, for register / ISA inspection)
  if (ut == 2 && kt == 3 && loc.scan.nc == 1) return tile2f_launch_n<2, 3, 1>(pf, stream);
  return MIA_ERR_UNSUPPORTED;
#else
  return ut == 1 ? tile2f_launch_u<1>(pf, kt, stream) : tile2f_launch_u<2>(pf, kt, stream);
#endif
}

}  // namespace mia

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

bool tile2f_covers(int m, int k, int ut, int n_coord) {
  const int kt = (k + 15) >> 4;
  return m >= 1 && ut >= 1 && ut <= 2 && kt >= 1 && kt <= 6 && n_coord >= 1 && n_coord <= MIA_MAX_COORD;
}

template <int UT, int KT, int NC, bool MROWS>
static int tile2f_launch_m(const Tile2FParams& pf, hipStream_t stream) {
  const size_t a = tile2_lds_bytes(UT, pf.t.k), b = tile_loc_lds(UT);
  const size_t lds = a > b ? a : b;
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
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

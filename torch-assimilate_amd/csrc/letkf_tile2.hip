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
#include "letkf_tile2_kernel.h"

namespace mia {


template <int UT, int KT, bool MROWS, int WAVES>
__global__ __launch_bounds__(64, WAVES)
void letkf_tile2_kernel(Tile2Params P) { tile2_body<UT, KT, MROWS, 0>(P, nullptr, (int64_t)blockIdx.y * gridDim.x + blockIdx.x); }

#ifdef MIA_TILE_STAMPS
extern "C" int mia_debug_tile2_stamps(long long* host, int n_tiles) {
  if (n_tiles > kT2StampTiles) n_tiles = kT2StampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile2_stamps), sizeof(long long) * kT2StampN * (size_t)n_tiles);
}
#endif

size_t tile2_lds_bytes(int ut, int k) {
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
  note_analysis_kernel("letkf_tile2_kernel<%d, %d, %s, %d>", UT, KT, MROWS ? "true" : "false", WAVES);
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

// the kernels address a record as base + 32-bit byte offset: (P + 1) records must fit 4 GB (2.4e7 observations at k = 40)
bool tile2_records_addressable(int k, int64_t P) { return (P + 1) * (int64_t)split_rec_bytes(k) < ((int64_t)1 << 32); }

int tile2_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const void* rec, int64_t P,
                          const void* tile_lists, int ut, float inf_factor, float* Xa, int64_t ldo, int64_t o0,
                          int32_t* flags, int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c,
                          hipStream_t stream, int seg_len, int64_t seg_stride, const Tile2Housekeeping* hk, const Tile2Loc* loc) {
  if (seg_len < 0 || (seg_len & 15) || (seg_len > 0 && ng >= ((int64_t)1 << 31))) return MIA_ERR_UNSUPPORTED;
  if (!flags || !retry_count || !tab_hdr || !tab_c || !tile_lists || !rec) return MIA_ERR_UNSUPPORTED;
  const int kt = (k + 15) >> 4;
  if (ut < 1 || ut > 6 || ut > kt + 1 || !tile2_covers(m, k, 0, ut - 1, ldx, ldo, ng)) return MIA_ERR_UNSUPPORTED;
  if (!tile2_records_addressable(k, P)) return MIA_ERR_UNSUPPORTED;
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
  { int pm = 0, tr = 0; MIA_EXP_SET(pm, "MIA_TILE2_PRIO", atoi); MIA_EXP_SET(tr, "MIA_TILE2_TRIM", atoi); tp.stagger |= (pm & 0xff) << 8 | tr << 16; }
  // the wavefronts localise their tiles themselves (the caller built no lists: it asked tile2f_covers first)
  if (loc) return tile2f_launch(tp, *loc, ut, kt, stream);
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

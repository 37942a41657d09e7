// Internal launchers shared between translation units (not part of the C ABI).
#pragma once
#include "mia_common.h"
#include "mia_localize_dev.h"

namespace mia {

// a segment's 64 slot counters sit kSlotStride ints apart (one per 256 bytes): atomics that share a cache line
// are serialised by the L2 channel that owns it (25 000 on one line cost more than the whole analysis)
constexpr int kSlotStride = 64;

// IEnKS update (tau = 1) through the weights variant of the matfun kernel: variant 1 transform (points whose Wp is not the
// identity are declined), 2 bundle (inv_eps = 1 / epsilon)
struct IenksOpts { int variant; const float* W_in; int64_t w_stride; float inv_eps; };

// letkf_cheb.hip.  MIA_ERR_UNSUPPORTED when the shape is outside the matfun route.
// seg_len > 0: one segmented launch over the ng points (native step driver): segment s = points
// [s*seg_len, (s+1)*seg_len) writes the (m*k, seg_len) buffer at Xa + s*seg_stride (ldo = seg_len) and counts
// its finished points in the 64 slot counters done[(s*64 + j) * kSlotStride] (zeroed by the caller); needs seg_len % 8 == 0.
int cheb_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                         float inf_factor, int kernel_mode, float gamma, float* Xa, int64_t ldo, int64_t o0,
                         int32_t* flags, int32_t* retry_count, const ScanParams* scan, int32_t* stats,
                         hipStream_t stream, int seg_len = 0, int64_t seg_stride = 0, int32_t* done = nullptr,
                         float* W_out = nullptr /* weights-output variant: [ng][k][k] */, const IenksOpts* ienks = nullptr);

// apply_local.hip: the per-grid-point weight transform on tiles of sixteen points (float32, k <= 96; MIA_ERR_UNSUPPORTED otherwise)
int apply_local_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* W, float* Xa,
                            int64_t ldo, int64_t o0, hipStream_t stream);

// ... and with ONE weight matrix for all points (the global ETKF's transform): grid points as the columns of a plain product
int apply_global_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* W, float* Xa,
                             int64_t ldo, int64_t o0, hipStream_t stream);

// Chebyshev coefficient tables of the matfun kernels (letkf_cheb.hip, cheb_coef_table): geometric grid of scaled
// spectral bounds T = L / reg, 32 per octave over 2^-24 .. 2^8; entry i holds {degree, bits of 2 / T} and 64 (phi, psi)
// coefficient pairs, zero beyond the degree.
constexpr int kTabPerOctave = 32, kTabIdx0 = 24 * kTabPerOctave, kTabN = 32 * kTabPerOctave, kTabDeg = 64;

// letkf_tile.hip: the same analysis with sixteen grid points per wavefront (dual route, k <= 64, few state rows).
// tile_route_covers: shape test; tile_analysis_launch: MIA_ERR_UNSUPPORTED outside it.  nbr_w is float64 or (w_f32)
// float32 lists.
bool tile_route_covers(int m, int k, int p_max);
int tile_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                         const int32_t* nbr_cnt, const int32_t* nbr_idx, const void* nbr_w, int w_f32, int p_cap,
                         int p_max, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                         int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c, hipStream_t stream,
                         int seg_len = 0, int64_t seg_stride = 0, int32_t* done = nullptr);
// (letkf_tile_split.hip) the same launch on the split-precision instantiations: every product on v_mfma_f32_16x16x32_f16
// with operands carried as pairs of halves; tile_analysis_launch forwards here when the option tile_split is on
int tile_split_analysis_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* rec,
                               const int32_t* nbr_cnt, const int32_t* nbr_idx, const void* nbr_w, int w_f32, int p_cap,
                               int p_max, float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                               int32_t* retry_count, int dmax, const int2* tab_hdr, const float2* tab_c, hipStream_t stream,
                               int seg_len, int64_t seg_stride, int32_t* done);

// Completion event for the next tile-kernel launch of this thread (set by the step driver around the analysis call of a
// step in flight): the launch then carries the event in its own dispatch packet (hipExtLaunchKernel) instead of the caller
// recording a marker packet behind it -- one packet less between two kernels of the analysis queue.  Cleared by the launch
// that takes it.
hipEvent_t& launch_stop_event();
hipEvent_t& launch_start_event();     // optional timing partner of the above (the dispatch's own start time)
// The analysis kernel a launch function has just put on a stream, under the name rocprofv3 records for it (template arguments
// included): read back by mia_last_analysis_kernel, so that bench lines and tools label their figures with the kernel that
// ran, not with a guess from the route options.  printf-style.
void note_analysis_kernel(const char* fmt, ...);
// launches of the tile kernel made by this thread so far; whether a launch with these sizes would be accepted by it
unsigned long long& tile_launch_count();
bool tile_launch_would_serve(int m, int k, int p_max, int p_cap, int64_t ldx, int64_t ldo, int64_t ng, int seg_len);
// (letkf_cheb.hip) whether mia_letkf_analysis_matfun_f32 with these arguments ends in the tile kernel -- the dispatch rule
// of cheb_analysis_launch, evaluated without launching (builds the coefficient table on `stream` if it does not exist yet)
bool cheb_tile_will_serve(int m, int k, int p_max, int p_cap, float gamma, int64_t ldx, int64_t ldo, int64_t ng, hipStream_t stream);

// (letkf_cheb.hip) coefficient table of the dual route at the default truncation target (built on first use, per device)
bool cheb_dual_table(hipStream_t stream, const int2** hdr, const float2** c);
// ... and of the primal route (the RBF-kernelised tile kernel, lketkf_tile.hip)
bool cheb_primal_table(hipStream_t stream, const int2** hdr, const float2** c);

// one-wave kernel on `stream` that returns once the 64 slot counters at done64[j * kSlotStride] sum to `expected` (bounded
// polling: after ~seconds it sets bit 0 of *err and returns, so the grid always drains)
int segment_wait_launch(const int32_t* done64, int expected, int32_t* err, hipStream_t stream);

}  // namespace mia

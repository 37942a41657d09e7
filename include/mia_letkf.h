/* mia_letkf.h -- C ABI of the MI355X-native LETKF local-analysis engine.
 *
 * The reference (tobifinn/torch-assimilate, "pytassim" v0.2.1) is pure Python and has no
 * native/FFI boundary; its seam for this path is the per-grid-point closure
 *   wrapper_localization(wrapper_bridge(ETKFModule))      pytassim/interface/wrapper.py:29-99
 * driven by LETKF.estimate_weights                         pytassim/interface/letkf.py:104-148
 * and followed by BaseAssimilation._apply_weights          pytassim/interface/base.py:257-278.
 * Each entry point below replaces one stage of that seam; the comment on every function
 * cites the reference code it stands in for.  INTEGRATION.md shows the ctypes stub a
 * pytassim maintainer would add.
 *
 * Conventions
 *  - all array pointers are DEVICE pointers (gfx950 HBM); the library never allocates,
 *    frees or retains caller memory; scratch comes from the `ws` argument whose size is
 *    obtained from the matching *_workspace_bytes query;
 *  - `stream` is a hipStream_t passed as void*; everything is enqueued on it, nothing
 *    synchronises the host;
 *  - functions never throw; return value 0 = ok, <0 = invalid argument (MIA_ERR_*),
 *    >0 = hipError_t of the failing runtime call;
 *  - dense layouts are row-major with the LAST index fastest;
 *  - ensemble-space results use the reference's orientation:
 *    weights[i][j] = w_mean[i] + W[i][j],  i = 'ensemble', j = 'ensemble_new'
 *    (pytassim/core/etkf.py:102, interface/letkf.py:145-146).
 */
#ifndef MIA_LETKF_H
#define MIA_LETKF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIA_VERSION 100 /* 0.1.0 */

#define MIA_OK 0
#define MIA_ERR_NULL (-1)        /* required pointer is NULL */
#define MIA_ERR_SIZE (-2)        /* negative / zero / inconsistent size */
#define MIA_ERR_UNSUPPORTED (-3) /* shape outside what the kernels are built for (e.g. LDS) */
#define MIA_ERR_WORKSPACE (-4)   /* workspace too small */
#define MIA_ERR_ALIGN (-5)       /* pointer not aligned as documented */
#define MIA_ERR_COMM (-6)        /* RCCL / communicator failure: see mia_comm_last_error() */

/* per-grid-point flag bits written to `flags_opt` */
#define MIA_FLAG_OVERFLOW 1 /* more local observations than p_max: point NOT analysed */
#define MIA_FLAG_NOCONV 2   /* Jacobi eigensolver hit its sweep cap (result still returned) */
#define MIA_FLAG_NONFINITE 4 /* non-finite value met in the local block */
#define MIA_FLAG_RETRY 8    /* matfun route declined this point (spectrum too wide): redo with the eigensolver */
/* bits 8-15: Jacobi sweeps started, bits 16-31: tournament rounds that rotated (diagnostics) */
#define MIA_FLAG_MASK 0xff

/* taper of the localisation */
#define MIA_TAPER_GC 0     /* GaspariCohn,    pytassim/localization/gaspari_cohn.py:40-136 */
#define MIA_TAPER_GC_INF 1 /* GaspariCohnInf, pytassim/localization/gaspari_cohn.py:139-254 (one radius) */

#define MIA_MAX_COORD 3
#define MIA_MAX_RADII 3

int mia_version(void);
const char* mia_status_string(int status);

/* ------------------------------------------------------------------------------------
 * Route options: explicit, process-wide switches for the kernel routes a caller (or a test) may want to steer.  No
 * counterpart in the reference (whose only knob on this path is `chunksize`, interface/letkf.py:80); they replace the
 * MIA_* environment variables that round 1 read inside launch code.  value < 0 restores the default.
 *   "cheb_dmax"       62  largest Chebyshev degree the matrix-function kernels accept before they decline a grid point
 *                         (MIA_FLAG_RETRY -> eigensolver); 3 .. 62
 *   "cheb_table"       1  Chebyshev coefficients from the per-device table / 0: computed inside the kernel
 *   "cheb_rowbatch"    1  m >= 8 state rows in 16-row MFMA batches / 0: row by row
 *   "cheb_big"         1  64 < k <= 128 with more than 64 local observations: matrix-function kernel / 0: eigensolver
 *   "tile"             1  sixteen grid points per wavefront where the shape allows (csrc/letkf_tile.hip) / 0: one
 *   "tile_split"       1  the products of that kernel as split-precision half MFMAs (f32 operands carried as pairs of halves,
 *                         f32 accumulation; same accuracy as f32 MFMAs) / 0: f32 MFMAs (results then do
 *                         not depend on which tile a grid point falls into, bit for bit)
 *   "localize_quad"    1  neighbour lists of capacity <= 32: four lanes per grid point / 0: one thread per grid point (same lists)
 *   "step_hostwait"    1  steps handed to the launch threads (mia_letkf_step_submit, MIA_STEP_NO_JOIN): the launch thread
 *                         waits for the step's preparation on the host and enqueues the analysis kernel without a stream
 *                         wait in front of it / 0: the analysis stream waits for the preparation's event
 *   "step_lazy_sort"   1  step driver: when the block's analysis is one launch of the sixteen-points-per-wavefront kernel (which
 *                         ranks observations itself), the observation index is built without its per-cell sort; the lists of
 *                         declined points are sorted before the eigensolver redoes them (same results, bit for bit) / 0: always
 *   "segment_signal"   1  step driver with several pieces: one segmented launch / 0: one launch + event per piece
 *   "tile_lists"       1  step driver: tile-shaped lists + split records + the analysis kernel of csrc/letkf_tile2.hip where the
 *                         shape allows (needs "tile" and "tile_split") / 0: per-point lists and csrc/letkf_tile.hip
 *   "bucket_index"     1  step driver, tile route: the observations are binned by ONE kernel into fixed-capacity buckets of the cell
 *                         grid the step's workspace already holds (bounding box validated per observation, rebuilt when it no
 *                         longer holds) / 0: bounding box + count + scan + scatter kernels every step
 *   "tile_pair"        1  tile route, unions of more than 32 slots and one state row per grid point: two wavefronts per tile,
 *                         each with the Gram fragments and recurrence vectors of its own row blocks (csrc/letkf_tile2p.hip; config 4:
 *                         0.186 -> 0.172 ms per 1e5 points, a 2-D mesh with 64-slot unions 0.092 -> 0.073) (1) or one (0)
 *   "tile_fused"       1  step driver, tile route over the bucket index, unions of at most 32 slots, no
 *                         geometry epoch declared: every analysis wavefront localises its own tile first (csrc/letkf_tile2f.hip -- the
 *                         list kernel's code, bit-identical results; no list kernel, no tile lists in memory) / 0: lists first
 *   "step_coalesce"    0  steps in flight (mia_letkf_step_submit) on the fused kernel: 1 .. 4: the launch thread keeps at most this many
 *                         analysis launches of its own running and hands the tiles of the steps that become ready meanwhile -- up to
 *                         four steps -- to ONE grid (letkf_tile2fb_kernel: every block takes its own step's parameters; same code per
 *                         tile, bit-identical results; a timed step always opens a launch) / 0 (default): one launch per step.  The
 *                         merged launch is cheaper per step (31 against 47 us) but the held-back steps lengthen the pipeline's loop by
 *                         more: 0.050-0.052 against 0.047 ms per step at config 2 (profiles/r05_coalesce.txt)
 * Scope: process-wide defaults, read when a call ENQUEUES its work -- for steps handed to the launch threads
 * (mia_letkf_step_submit) at submission: a step runs with the routes that were in force when it was submitted, whatever is
 * set afterwards.  What differs per runner of one process travels in the call's own arguments (method, step_flags:
 * MIA_STEP_NO_TILE_LISTS, MIA_STEP_SCAN_INDEX, MIA_STEP_TILE_EXTRA).
 * Returns MIA_ERR_UNSUPPORTED for an unknown name, MIA_ERR_SIZE for a value out of range. */
int mia_set_option(const char* name, int value);
int mia_get_option(const char* name, int* value);
/* The analysis kernel launched last by any entry point of this process, under the name rocprofv3 records for it (template
 * arguments included, e.g. "letkf_tile2f_kernel<2, 3, 1, false, 4>"), NUL-terminated into buf[n]; "" before the first launch.
 * Diagnostics only (bench lines and tools label their figures with the kernel that ran): not a route switch. */
int mia_last_analysis_kernel(char* buf, int n);

/* ------------------------------------------------------------------------------------
 * Gaspari-Cohn taper of normalised distances r = dist / c (unit-testable stage).
 * Replaces GaspariCohn._f1/_f2 and the piecewise assembly in
 * pytassim/localization/gaspari_cohn.py:78-95,127-133 (strict `<` at r = 1 and r = 2).
 * ---------------------------------------------------------------------------------- */
int mia_gaspari_cohn_f64(const double* r, int64_t n, double* w, void* stream);
int mia_gaspari_cohn_f32(const float* r, int64_t n, float* w, void* stream);
/* Form-factor-infinity variant: GaspariCohnInf._f1.._f4 and the branch assembly of
 * gaspari_cohn.py:176-215,244-251 (thresholds 0.5, 1, 1.5, 2, strict `<`). */
int mia_gaspari_cohn_inf_f64(const double* r, int64_t n, double* w, void* stream);
int mia_gaspari_cohn_inf_f32(const float* r, int64_t n, float* w, void* stream);

/* ------------------------------------------------------------------------------------
 * Localisation: for every grid point g in [g0, g1) the list of observations with
 * Gaspari-Cohn weight > gc_eps and sqrt(weight) for each.
 * Replaces GaspariCohn.localize_obs (gaspari_cohn.py:97-136) evaluated once per grid
 * point inside wrapper_localization (interface/wrapper.py:88-91), for the built-in
 * metric family: coordinate c belongs to radius group coord_group[c]; the distance of
 * group i is the Euclidean norm over its coordinates; weight = prod_i GC(dist_i/gc_c[i]).
 * (n_coord = 1, one group: |x_g - x_o|, the metric of examples/benchmark_letkf.py:85-87.)
 * Evaluated in float64 like the reference, so the use/skip decision is the reference's.
 * Instead of the reference's O(G*P) all-pairs pass the observations are binned into a
 * uniform cell grid (cell edge >= 2*c per coordinate) built in `ws`.
 *
 *  grid_xyz [G_total][n_coord] f64     obs_xyz [P][n_coord] f64
 *  nbr_cnt  [g1-g0]           i32  true number of local obs (may exceed p_cap)
 *  nbr_idx  [g1-g0][p_cap]    i32  obs indices, ascending cell order, -1 padded
 *  nbr_w    [g1-g0][p_cap]    f64  sqrt(GC weight)           (wrapper.py:91)
 *  stats    [2]               i32  [0] = max count over the shard, [1] = #points with
 *                                  count > p_cap (lists truncated; caller must retry)
 * ---------------------------------------------------------------------------------- */
int mia_letkf_localize_workspace_bytes(int64_t P, int n_coord, size_t* bytes);
int mia_letkf_localize_f64(const double* grid_xyz, int64_t g0, int64_t g1,
                           const double* obs_xyz, int64_t P, int n_coord,
                           const int32_t* coord_group /* host, [n_coord] */,
                           const double* gc_c /* host, [n_r] */, int n_r, double gc_eps,
                           int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w,
                           int32_t* stats, void* ws, size_t ws_bytes, void* stream);

/* The same with the taper selectable (MIA_TAPER_*): GaspariCohnInf.localize_obs, gaspari_cohn.py:217-254. */
int mia_letkf_localize_taper_f64(int taper, const double* grid_xyz, int64_t g0, int64_t g1,
                                 const double* obs_xyz, int64_t P, int n_coord,
                                 const int32_t* coord_group /* host */, const double* gc_c /* host */, int n_r,
                                 double gc_eps, int p_cap, int32_t* nbr_cnt, int32_t* nbr_idx, double* nbr_w,
                                 int32_t* stats, void* ws, size_t ws_bytes, void* stream);

/* The observation cell index alone (what mia_letkf_localize_f64 builds first), for the fused route below. */
int mia_letkf_index_build_f64(const double* obs_xyz, int64_t P, int n_coord,
                              const int32_t* coord_group /* host */, const double* gc_c /* host */, int n_r,
                              void* ws, size_t ws_bytes /* mia_letkf_localize_workspace_bytes */, void* stream);

/* Same lists from caller-evaluated distances (an arbitrary Python dist_func evaluated on
 * the host in batch): dist [n_r][g1-g0][p_cap] f64, cand_idx [g1-g0][p_cap] (-1 = pad).
 * Compacts in place the candidates whose weight product exceeds gc_eps. */
int mia_letkf_localize_from_dist_f64(const double* dist, const int32_t* cand_idx, int64_t n_pts,
                                     int p_cap, const double* gc_c /* host */, int n_r,
                                     double gc_eps, int32_t* nbr_cnt, int32_t* nbr_idx,
                                     double* nbr_w, int32_t* stats, void* stream);

int mia_letkf_localize_from_dist_taper_f64(int taper, const double* dist, const int32_t* cand_idx, int64_t n_pts,
                                           int p_cap, const double* gc_c /* host */, int n_r,
                                           double gc_eps, int32_t* nbr_cnt, int32_t* nbr_idx,
                                           double* nbr_w, int32_t* stats, void* stream);

/* ------------------------------------------------------------------------------------
 * Local analysis for grid points [g0, g1): mask & sqrt(rho)-scale (wrapper.py:91-97),
 * ETKF weights (core/etkf.py:57-103: Gram matrix, symmetric eigensolve with clamp>=0 and
 * (k-1)/inf shift, Pa, w_mean, symmetric square root W; empty list -> sqrt(inf)*I,
 * etkf.py:91-95) and the ensemble transform xa = mean + X' (w_mean 1^T + W)
 * (interface/base.py:257-278), fused.
 *
 *  X   [m][k][ldx]   prior ensemble, grid fastest (pytassim dims var_name*time, ensemble,
 *                    grid); point g is column g of every (m, k) row
 *  Yb  [k][P]        R^-1/2-normalised obs-space perturbations (base.py:359-379)
 *  d   [P]           normalised innovations
 *  nbr_*             lists for this shard as produced above, row stride p_cap
 *  p_max             upper bound on nbr_cnt over the shard (stats[0]); sizes the LDS block
 *  Xa  [m][k][ldo]   analysis; point g is written to column (g - g0 + o0)
 *  W_opt             NULL or [g1-g0][k][k] weights (what estimate_weights returns)
 *  flags_opt         NULL or [g1-g0] MIA_FLAG_* bits
 * ---------------------------------------------------------------------------------- */
int mia_letkf_analysis_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes);
int mia_letkf_analysis_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                           const float* Yb, const float* d, int64_t P,
                           const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                           int p_cap, int p_max, float inf_factor,
                           float* Xa, int64_t ldo, int64_t o0, float* W_opt, int32_t* flags_opt,
                           void* ws, size_t ws_bytes, void* stream);
int mia_letkf_analysis_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                           const double* Yb, const double* d, int64_t P,
                           const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                           int p_cap, int p_max, double inf_factor,
                           double* Xa, int64_t ldo, int64_t o0, double* W_opt, int32_t* flags_opt,
                           void* ws, size_t ws_bytes, void* stream);

/* The same in two steps, so that the packed observation records can be reused across shards /
 * calls and the analysis kernel can be timed on its own: pack [k][P] (+ d[P]) into obs-major
 * records rec [P][kp], kp = round_up(k + 1, 4) (what the per-point mask-gather
 * `arg[..., luse]`, wrapper.py:94-97, reads), then analyse from the records.
 * gamma <= 0 selects the plain ETKF core, gamma > 0 the RBF-kernelised one. */
int mia_letkf_pack_obs_f32(const float* Yb, const float* d, int k, int64_t P, float* rec, void* stream);
int mia_letkf_pack_obs_f64(const double* Yb, const double* d, int k, int64_t P, double* rec, void* stream);
int mia_letkf_analysis_packed_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                  const float* rec, int64_t P,
                                  const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                  int p_cap, int p_max, float inf_factor, float gamma,
                                  float* Xa, int64_t ldo, int64_t o0, float* W_opt, int32_t* flags_opt,
                                  void* stream);
int mia_letkf_analysis_packed_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                  const double* rec, int64_t P,
                                  const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                  int p_cap, int p_max, double inf_factor, double gamma,
                                  double* Xa, int64_t ldo, int64_t o0, double* W_opt, int32_t* flags_opt,
                                  void* stream);

/* Eigensolver-free route (whenever the weights are not requested; faster than the eigensolver kernel at every number
 * of state rows measured, tools/time_rows.py): the two matrix
 * functions the analysis needs, (C+reg)^-1 and (C+reg)^-1/2 (core/etkf.py:67-76), are applied to the state row
 * by a Chebyshev expansion whose degree is fixed per grid point from a Gershgorin bound (SURVEY.md section 7
 * notes such evaluations are valid because W and Pa are functions of A only).  Points that would need a
 * degree above the built-in cap get MIA_FLAG_RETRY in flags[] (required, not optional here), are counted in
 * *retry_count (device int32, zeroed by the caller) and are left untouched in Xa; the caller then runs
 * mia_letkf_analysis_retry_f32 - the eigensolver kernel restricted to the flagged points.
 * Same argument meaning as mia_letkf_analysis_packed_f32. */
int mia_letkf_analysis_matfun_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                  const float* rec, int64_t P,
                                  const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                  int p_cap, int p_max, float inf_factor, float gamma,
                                  float* Xa, int64_t ldo, int64_t o0, int32_t* flags, int32_t* retry_count,
                                  void* stream);
/* The weights themselves without an eigensolver (what LETKF.estimate_weights returns, interface/letkf.py:145-146, next to
 * the analysis): phi(S) as an n x n matrix from the same Chebyshev recurrence run on the identity -- MFMA products
 * S T_j -- then W = w_mean 1^T + f0 I + Yl phi(S) Yl^T.  Dual route (p_max <= k) up to order 32, primal route (p_max > k, or
 * gamma > 0: the RBF filter; W = w_mean 1^T + phi(S) with S the k x k member Gram / centred kernel matrix) up to k = 48; otherwise
 * MIA_ERR_UNSUPPORTED (use mia_letkf_analysis_packed_f32 with W_opt).  Declined points: MIA_FLAG_RETRY, counted in
 * *retry_count, left untouched in Xa and W; mia_letkf_weights_retry_f32 redoes them with the eigensolver kernel. */
int mia_letkf_weights_matfun_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                 const float* rec, int64_t P,
                                 const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                 int p_cap, int p_max, float inf_factor, float gamma /* > 0: RBF-kernelised core */,
                                 float* Xa, int64_t ldo, int64_t o0, float* W /* [g1-g0][k][k] */,
                                 int32_t* flags, int32_t* retry_count, void* stream);
int mia_letkf_weights_retry_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const float* rec, int64_t P,
                                const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                int p_cap, int p_max, float inf_factor, float gamma,
                                float* Xa, int64_t ldo, int64_t o0, float* W, int32_t* flags, void* stream);
/* ------------------------------------------------------------------------------------
 * Tile route (round 3): the same localisation + analysis with the unit of work a TILE of sixteen
 * consecutive grid points.  Replaces, for the built-in metric family, the per-grid-point loop of
 * LETKF.estimate_weights (interface/letkf.py:127-146) over wrapper_localization (interface/wrapper.py:86-98)
 * -> GaspariCohn.localize_obs (localization/gaspari_cohn.py:97-136) -> ETKFModule (core/etkf.py:57-103),
 * followed by _apply_weights (interface/base.py:257-278).
 *
 * mia_letkf_localize_tiles_f64  cell index of the observations (built in `ws`, as mia_letkf_localize_f64) and, per
 *     tile, the UNION of its sixteen lists (observation indices by rank) + the 16 x U matrix of sqrt(GC weight)
 *     (0 = not local) in the layout of the analysis wavefront's registers (csrc/mia_tiles.h).  float64 distance and
 *     taper arithmetic: the use / skip decision is the reference's.  p_max = bound on the local observations of a
 *     point; a tile offers 16 * (ceil((p_max + 8) / 16) + extra_blocks) <= 96 slots, at most 16 more than the ensemble
 *     size rounded up to 16 (sixteen consecutive points of a 1-D network with one observation per grid step see
 *     p_max + 15 observations: extra_blocks = 1 where p_max + 15 exceeds the default).
 *     stats [2] i32: [0] longest list of the shard, [1] tiles whose union did not fit their slots -- such tiles are
 *     NOT analysed by mia_letkf_analysis_tiles_f32 (MIA_FLAG_OVERFLOW, NaN); the caller repeats with more extra_blocks
 *     or with the list route (mia_letkf_localize_f64 + mia_letkf_analysis_matfun_f32).
 * mia_letkf_pack_split_f32  [k][P] perturbations + d[P] -> P + 1 split records of mia_letkf_split_record_bytes(k)
 *     bytes: every record scaled by its own power of two and carried as pairs of halves (hi, lo) for the
 *     half-precision matrix cores at f32 accuracy; record P is the all-zero record.
 * mia_letkf_analysis_tiles_f32  analysis of grid points [g0, g1) from tile lists of exactly that range and split
 *     records; dual route (p_max <= k <= 96, p_max <= 88), any number m of state rows.  flags / retry_count as
 *     mia_letkf_analysis_matfun_f32: declined points (and every point of a tile that holds a non-finite record) get
 *     MIA_FLAG_RETRY and are redone by mia_letkf_analysis_retry_f32 from per-point lists.
 * mia_letkf_weights_tiles_f32  the same analysis AND the weights W [g1-g0][k][k], W[g][i][j] = w_mean_i + W_pert_ij (what
 *     LETKF.estimate_weights returns, interface/letkf.py:127-146): the Chebyshev recurrence on a matrix block per grid point,
 *     four points of a tile per wavefront (csrc/letkf_tile2w.hip).  Unions of at most 32 slots
 *     (ceil((p_max + 8) / 16) + extra_blocks <= 2), otherwise MIA_ERR_UNSUPPORTED (use
 *     mia_letkf_weights_matfun_f32).  Declined points (the analysis kernel's, and here also every point whose Chebyshev degree
 *     exceeds 36: their analysis is written, their weights are not): MIA_FLAG_RETRY, counted, W left untouched;
 *     mia_letkf_weights_retry_f32 redoes them from per-point lists.
 * ---------------------------------------------------------------------------------- */
/* bit set in stats[1] (the count of tiles without a list) when some tile's points span more index cells than the kernel
 * scans (64 cells / 64 rows: scattered point orderings, cells coarse against the grid spacing): more slots (extra_blocks) cannot
 * help such a tile -- use the per-point lists */
#define MIA_TILE_BOX_OVERFLOW (1 << 30)
#define MIA_STEP_STATUS_SAMPLED 64 /* counters[3] / [7] of a step: see mia_letkf_sharded_step_streams_f32 */
#define MIA_STEP_STATUS_NONFINITE 128 /* with MIA_STEP_STATUS_SAMPLED (the fused kernel ran): some grid point's flags carry
                                       * MIA_FLAG_NONFINITE; absent: none does -- no scan of the per-point flags needed */
int mia_letkf_tile_lists_bytes(int64_t n_points, int p_max, int extra_blocks, size_t* bytes);
int mia_letkf_localize_tiles_f64(int taper, const double* grid_xyz, int64_t g0, int64_t g1,
                                 const double* obs_xyz, int64_t P, int n_coord,
                                 const int32_t* coord_group /* host */, const double* gc_c /* host */, int n_r,
                                 double gc_eps, int p_max, int extra_blocks, void* tile_lists, size_t tile_lists_bytes,
                                 int32_t* stats, void* ws, size_t ws_bytes, void* stream);
int mia_letkf_split_record_bytes(int k, size_t* bytes);
int mia_letkf_pack_split_f32(const float* Yb, const float* d, int k, int64_t P, void* split_rec /* (P + 1) records */,
                             void* stream);
int mia_letkf_analysis_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                 const void* split_rec, int64_t P, const void* tile_lists, int p_max, int extra_blocks,
                                 float inf_factor, float* Xa, int64_t ldo, int64_t o0, int32_t* flags,
                                 int32_t* retry_count, void* stream);
int mia_letkf_weights_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const void* split_rec, int64_t P, const void* tile_lists, int p_max, int extra_blocks,
                                float inf_factor, float* Xa, int64_t ldo, int64_t o0, float* W /* [g1-g0][k][k] */,
                                int32_t* flags, int32_t* retry_count, void* stream);

/* The RBF-kernelised filter on tiles: KETKFModule._estimate_weights with RBFKernel(gamma) (core/ketkf.py:65-94,
 * kernels/rbf.py:75-81,110-111) under wrapper_localization (interface/wrapper.py:86-98) + _apply_weights
 * (interface/base.py:257-278), sixteen grid points per wavefront from the tile lists of mia_letkf_localize_tiles_f64 and
 * the float32 perturbations Yb [k][P] / innovations d [P] themselves (no records of any kind).  The squared member
 * distances of all sixteen points are one f32 matrix-core product per tile (pairs x union slots x points,
 * csrc/lketkf_tile.hip).  2 <= k <= 40, a tile's union within 64 slots; MIA_ERR_UNSUPPORTED otherwise
 * (mia_letkf_analysis_matfun_f32 with gamma > 0 takes every shape).  flags / retry_count as mia_letkf_analysis_tiles_f32;
 * declined points are redone by mia_letkf_analysis_retry_f32 (gamma > 0) from per-point lists. */
/* Whether the tile-route kernels take a shape (1) or not (0): the test mia_letkf_analysis_tiles_f32 (gamma <= 0) /
 * mia_lketkf_rbf_analysis_tiles_f32 (gamma > 0) apply before launching -- ensemble size, slots of a tile's union, LDS, and
 * every global access as base + 32-bit byte offset (k * ldx * 4, k * ldo * 4 [, k * P * 4] < 2^31). */
int mia_letkf_tiles_cover(int m, int k, int p_max, int extra_blocks, int64_t ldx, int64_t ldo, int64_t n_points, int64_t P,
                          float gamma);
int mia_lketkf_rbf_analysis_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                      const float* Yb, const float* d, int64_t P, const void* tile_lists, int p_max,
                                      int extra_blocks, float inf_factor, float gamma, float* Xa, int64_t ldo, int64_t o0,
                                      int32_t* flags, int32_t* retry_count, void* stream);

/* matfun route with the Gaspari-Cohn localisation fused in: every wavefront scans the observation index
 * (mia_letkf_index_build_f64) for its grid point itself, so no neighbour lists are written or read.
 * p_max_assumed sizes the launch (e.g. stats[0] of an earlier call on the same geometry); a grid point with
 * more local observations gets MIA_FLAG_OVERFLOW and NaN output, stats[0] = true maximum, stats[1] = number
 * of such points: when stats[1] != 0 the caller redoes the shard with the larger value (or the list route). */
int mia_letkf_analysis_matfun_fused_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                        const float* rec, int64_t P,
                                        const double* grid_xyz, int n_coord, const int32_t* coord_group /* host */,
                                        const double* gc_c /* host */, int n_r, double gc_eps,
                                        void* index_ws, size_t index_ws_bytes,
                                        int p_max_assumed, float inf_factor, float gamma,
                                        float* Xa, int64_t ldo, int64_t o0, int32_t* flags, int32_t* retry_count,
                                        int32_t* stats, void* stream);
int mia_letkf_analysis_retry_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                 const float* rec, int64_t P,
                                 const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                 int p_cap, int p_max, float inf_factor, float gamma,
                                 float* Xa, int64_t ldo, int64_t o0, int32_t* flags, void* stream);

/* Kernelised variant: KETKFModule with RBFKernel(gamma) (core/ketkf.py:65-94,
 * kernels/rbf.py:75-81,110-111), same localisation and transform (LKETKF,
 * interface/lketkf.py:77). */
int mia_lketkf_rbf_analysis_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const float* Yb, const float* d, int64_t P,
                                const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                int p_cap, int p_max, float inf_factor, float gamma,
                                float* Xa, int64_t ldo, int64_t o0, float* W_opt, int32_t* flags_opt,
                                void* ws, size_t ws_bytes, void* stream);
int mia_lketkf_rbf_analysis_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const double* Yb, const double* d, int64_t P,
                                const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                int p_cap, int p_max, double inf_factor, double gamma,
                                double* Xa, int64_t ldo, int64_t o0, double* W_opt, int32_t* flags_opt,
                                void* ws, size_t ws_bytes, void* stream);

/* Any other reference kernel, and the compositions of pytassim/kernels/base_kernels.py:61-161, as a kernel
 * EXPRESSION in reverse Polish form evaluated per element of K(Yb, Yb) and K(Yb, d) inside the analysis kernel.
 * Every reference kernel is a function of three pair statistics of two rows x, y of the localised block:
 *   x.y (kernels/utils.py:38-58), |x - y|_2^2 (utils.py:93-110), |x - y|_1 (utils.py:61-90, norm = 1):
 *   LinearKernel   linear.py:66-67       DOT
 *   PolyKernel     polynomial.py:78-81   DOT CONST(c) ADD CONST(p) POW
 *   TanhKernel     tanh.py:82-86         DOT CONST(a) MUL CONST(c) ADD TANH
 *   GaussKernel    rbf.py:75-81          SQDIST CONST(-1/(2 l^2)) MUL EXP        (scalar lengthscale)
 *   RationalKernel rational.py:81-87     SQDIST CONST(1/(2 a l^2)) MUL CONST(1) ADD CONST(-a) POW
 *   PeriodicKernel periodic.py:81-84     L1DIST CONST(pi/p) MUL SIN CONST(2) POW CONST(-2/l^2) MUL EXP
 *   OrnsteinUhlenbeckKernel orn_uhl.py:72-75   L1DIST CONST(-1/l) MUL EXP
 *   ScaleKernel    scale.py:70-73        CONST(c)
 *   DiagKernel     diag.py:64-72         DIAG(c)   (c where x and y are the same sample of a square Gram, else 0)
 *   Additive / Multiplicative / PowerKernel   base_kernels.py:98-161   <k1> <k2> ADD | MUL | POW
 * Same argument meaning as mia_letkf_analysis_packed_*; prog is a HOST array. */
#define MIA_KOP_DOT 1
#define MIA_KOP_SQDIST 2
#define MIA_KOP_L1DIST 3
#define MIA_KOP_CONST 4 /* push value */
#define MIA_KOP_DIAG 5  /* push value on the diagonal of K(Yb, Yb), 0 elsewhere and in K(Yb, d) */
#define MIA_KOP_ADD 6   /* binary operators pop y, pop x, push x op y */
#define MIA_KOP_MUL 7
#define MIA_KOP_POW 8
#define MIA_KOP_EXP 9   /* unary operators replace the top */
#define MIA_KOP_TANH 10
#define MIA_KOP_SIN 11
#define MIA_KERNEL_MAX_OPS 24
#define MIA_KERNEL_MAX_DEPTH 6
typedef struct { int32_t op; int32_t reserved; double value; } mia_kernel_op_t;
int mia_lketkf_kernel_analysis_packed_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                          const float* rec, int64_t P,
                                          const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                          int p_cap, int p_max, float inf_factor,
                                          const mia_kernel_op_t* prog /* host */, int n_ops,
                                          float* Xa, int64_t ldo, int64_t o0, float* W_opt, int32_t* flags_opt,
                                          void* stream);
int mia_lketkf_kernel_analysis_packed_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                          const double* rec, int64_t P,
                                          const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w,
                                          int p_cap, int p_max, double inf_factor,
                                          const mia_kernel_op_t* prog /* host */, int n_ops,
                                          double* Xa, int64_t ldo, int64_t o0, double* W_opt, int32_t* flags_opt,
                                          void* stream);

/* ------------------------------------------------------------------------------------
 * Global (unlocalised) ETKF: one (k, P) solve, ETKF.estimate_weights
 * (interface/etkf.py:99-120) -> weights [k][k]; and the global ensemble transform
 * _apply_weights with 2-D weights (base.py:257-278).
 * ---------------------------------------------------------------------------------- */
int mia_etkf_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes);
int mia_etkf_weights_f32(const float* Yb, const float* d, int k, int64_t P, float inf_factor,
                         float* W, int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream);
int mia_etkf_weights_f64(const double* Yb, const double* d, int k, int64_t P, double inf_factor,
                         double* W, int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream);
/* Global KERNELISED ETKF for any number of observations: KETKF.estimate_weights (interface/ketkf.py:34-123) -> one
 * KETKFModule.forward on the full (k, P) block (core/ketkf.py:65-94) with the kernel given as an expression
 * (mia_kernel_op_t, see mia_lketkf_kernel_analysis_packed_*; RBFKernel(gamma) = SQDIST CONST(-gamma) MUL EXP).
 * P = 0 returns the inflated prior sqrt(inf) I (core/etkf.py:91-95). */
int mia_ketkf_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes);
int mia_ketkf_weights_f32(const float* Yb, const float* d, int k, int64_t P, float inf_factor,
                          const mia_kernel_op_t* prog /* host */, int n_ops, float* W, int32_t* flags_opt,
                          void* ws, size_t ws_bytes, void* stream);
int mia_ketkf_weights_f64(const double* Yb, const double* d, int k, int64_t P, double inf_factor,
                          const mia_kernel_op_t* prog /* host */, int n_ops, double* W, int32_t* flags_opt,
                          void* ws, size_t ws_bytes, void* stream);
int mia_apply_weights_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                          const float* W, float* Xa, int64_t ldo, int64_t o0, void* stream);
int mia_apply_weights_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                          const double* W, double* Xa, int64_t ldo, int64_t o0, void* stream);

/* Ensemble transform with PER-GRID-POINT weights: _apply_weights with weights of dims (grid, ensemble, ensemble_new)
 * (base.py:257-278; what update_state does with the result of estimate_weights, filter.py:157-164, and what the
 * IEnKS does every iteration and at its end, variational.py:107-135).  W [g1-g0][k][k]. */
int mia_apply_local_weights_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const float* W, float* Xa, int64_t ldo, int64_t o0, void* stream);
int mia_apply_local_weights_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                const double* W, double* Xa, int64_t ldo, int64_t o0, void* stream);

/* ------------------------------------------------------------------------------------
 * Localised IEnKS: one Gauss-Newton update of the ensemble weights per grid point,
 * IEnKSTransformModule / IEnKSBundleModule.forward (core/ienks.py:108-141, 167-173) on the localised block, as
 * LocalizedIEnKSTransform.inner_loop / LocalizedIEnKSBundle.inner_loop run it through wrapper_localization with
 * args_to_skip=(0,) (interface/lienks.py:75-118; the weights are not masked, wrapper.py:92-93).
 *   W_in   [g1-g0][k][k] (w_stride = k*k), or ONE [k][k] matrix for every grid point (w_stride = 0: the prior
 *          weights of the first iteration, base.py:244-254)
 *   rec, nbr_*: packed observation records and local lists, as for mia_letkf_analysis_packed_*
 *   tau in [0, 1]; epsilon <= 0: transform variant (dh/dw = Wp^-1 Yl), epsilon > 0: bundle variant (dh/dw = Yl / eps)
 *   W_out  [g1-g0][k][k];  a grid point without local observations gets its weights back unchanged (ienks.py:135)
 * flags_opt: MIA_FLAG_OVERFLOW / NOCONV / NONFINITE (singular Wp).
 * ---------------------------------------------------------------------------------- */
int mia_lienks_update_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                          const float* rec, int64_t P,
                          const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                          float tau, float epsilon, float* W_out, int32_t* flags_opt, void* stream);
/* tau = 1 without an eigensolver, through the weights variant of the matfun kernel (Wp' = I + D phi(S) D^T,
 * w_mean' = D psi(S) (d + D^T w_mean); float32, dual route, order <= 32): the bundle variant (epsilon > 0) at every
 * iteration, the transform variant (epsilon <= 0) while Wp = I, i.e. the first iteration from the prior weights.
 * Declined points (another Wp, spectrum too wide): MIA_FLAG_RETRY in flags, counted in *retry_count, untouched in W_out;
 * mia_lienks_update_retry_f32 redoes exactly those with the general kernel.  MIA_ERR_UNSUPPORTED: use mia_lienks_update_f32. */
int mia_lienks_update_matfun_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                                 const float* rec, int64_t P,
                                 const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                 float epsilon, float* W_out, int32_t* flags, int32_t* retry_count, void* stream);
int mia_lienks_update_retry_f32(const float* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                                const float* rec, int64_t P,
                                const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                                float tau, float epsilon, float* W_out, int32_t* flags, void* stream);
int mia_lienks_update_f64(const double* W_in, int64_t w_stride, int k, int64_t g0, int64_t g1,
                          const double* rec, int64_t P,
                          const int32_t* nbr_cnt, const int32_t* nbr_idx, const double* nbr_w, int p_cap, int p_max,
                          double tau, double epsilon, double* W_out, int32_t* flags_opt, void* stream);

/* ------------------------------------------------------------------------------------
 * Observation-space preparation, the step immediately upstream of the analysis
 * (AssimilationInterface._get_obs_space_variables, interface/base.py:359-379): from the ensemble in observation
 * space hx [k][P] (row stride ldh) and the observations y [P] of one observation subset
 *     mean_j = mean_i hx[i][j]              (state.split_mean_perts, base.py:367-369)
 *     d      = (y  - mean) R^-1/2           (base.py:370-371, Observation.mul_rcinv, observation.py:290-295)
 *     Yb[i]  = (hx[i] - mean) R^-1/2        (base.py:372)
 * uncorr: R = diag(var), value * (1 / sqrt(var))                     (observation.py:241-245, 276-278)
 * corr:   R = cov [P][P] (symmetric positive definite, lower triangle read), value @ inv(cholesky(R).T)
 *         (observation.py:247-275) computed as a blocked Cholesky sweep over [R; values] - no inverse formed;
 *         *info_opt (device) = 0, or j+1 if the leading minor of order j+1 is not positive definite.
 * Outputs, each optional: Yb [k][P] (row stride ldy), d [P], rec [P][kp] = the packed records of
 * mia_letkf_pack_obs_*.  Several subsets are stacked (base.py:374-377 _stack_obs) by pointing Yb/d/rec at the
 * subset's offset inside the stacked arrays (ldy = total number of observations).
 * ---------------------------------------------------------------------------------- */
int mia_obs_space_uncorr_f32(const float* hx, int64_t ldh, const float* y, const float* var, int k, int64_t P,
                             float* Yb_opt, int64_t ldy, float* d_opt, float* rec_opt, void* stream);
int mia_obs_space_uncorr_f64(const double* hx, int64_t ldh, const double* y, const double* var, int k, int64_t P,
                             double* Yb_opt, int64_t ldy, double* d_opt, double* rec_opt, void* stream);
int mia_obs_space_corr_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes);
int mia_obs_space_corr_f32(const float* hx, int64_t ldh, const float* y, const float* cov, int k, int64_t P,
                           float* Yb_opt, int64_t ldy, float* d_opt, float* rec_opt, int32_t* info_opt,
                           void* ws, size_t ws_bytes, void* stream);
int mia_obs_space_corr_f64(const double* hx, int64_t ldh, const double* y, const double* cov, int k, int64_t P,
                           double* Yb_opt, int64_t ldy, double* d_opt, double* rec_opt, int32_t* info_opt,
                           void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * One assimilation step of one rank's block of grid points as ONE call (native driver; the per-step
 * orchestration of the entries above costs more host time than the GPU work it enqueues).
 *
 * Distribution: the reference distributes the same per-grid-point work as dask chunks of grid points
 * (interface/letkf.py:118-131, DaskLocalization) on one host; here rank r of `world` owns the block
 * [r*n, min(G,(r+1)*n)), n = ceil(G/world), analyses it in n_chunks pieces and every piece is all-gathered
 * (RCCL, on `comm_stream`) and placed into the full (m, k, G) result while the next piece is analysed on
 * `stream`.  comm == NULL: single rank, Xa is written directly, comm_stream unused.  n_chunks <= 15.
 *
 * Communicator: mia_comm_load(path of the RCCL library the process already uses), rank 0 draws
 * mia_comm_unique_id (128 bytes) which the host distributes by any means (e.g. torch.distributed broadcast),
 * every rank calls mia_comm_create on its device.  mia_comm_create_custom substitutes caller callbacks for
 * the two collectives (used by the tests to emulate a second rank on a one-GPU box).
 *
 * method: 0 auto (= matfun, with the eigensolver redoing declined points), 1 eigensolver kernel, 2 matfun.   p_max_assumed: bound of the local
 * observation count the launch is sized for (lists capacity = round_up(., 8)).
 * counters [8] i32 (device): [0] longest list of the block, [1] lists longer than the capacity, [2] grid points
 * the matfun kernel declined, [3] error bits (bit 0: a segment waiter timed out) and TWO status bits: MIA_STEP_STATUS_SAMPLED (64):
 * the step ran on the fused kernel (csrc/letkf_tile2f.hip), whose [0] is exact only when it exceeds p_max_assumed -- otherwise
 * the maximum over one tile of sixteen points in 64 (a caller that carries the bound from step to step keeps it, or lowers it
 * with a margin); and, with it, MIA_STEP_STATUS_NONFINITE (128): some grid point's flags carry MIA_FLAG_NONFINITE (absent: none
 * does, the per-point flags need not be read to know); [4..7] the same, max- (or-)
 * reduced over all ranks (comm == NULL: left zero, [0..3] are the whole story).  The host reads them once
 * after the call (the only synchronisation of the step):
 *   [5] != 0 or [4] > p_max_assumed  -> the bound did not hold on some rank: repeat the step (phase 0) on ALL
 *                                       ranks with p_max_assumed >= [4];
 *   [6] != 0                          -> call again with phase = 1 on ALL ranks: the eigensolver kernel redoes
 *                                       the declined points in place (no-op where none) and the pieces are
 *                                       exchanged again.  The workspace must be untouched between the phases.
 * flags [block length] i32: MIA_FLAG_* per grid point of the block.
 * ---------------------------------------------------------------------------------- */
typedef struct mia_comm mia_comm_t;
typedef int (*mia_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes_per_rank, void* stream);
typedef int (*mia_allreduce_max_i32_fn)(void* ctx, int32_t* buf, int n, void* stream);
int mia_comm_load(const char* rccl_library_path /* NULL: "librccl.so" */);
int mia_comm_unique_id(void* id128);
int mia_comm_create(const void* id128, int rank, int world, mia_comm_t** comm);
/* a communicator that only carries the partition (rank, world): for MIA_STEP_NO_GATHER steps; needs no RCCL */
int mia_comm_create_partition(int rank, int world, mia_comm_t** out);
int mia_comm_create_custom(int rank, int world, mia_allgather_fn allgather, mia_allreduce_max_i32_fn allreduce_max,
                           void* ctx, mia_comm_t** comm);
/* Optional stream for the placement of gathered pieces (NULL: the exchange stream).  Used only by steps enqueued with
 * MIA_STEP_NO_JOIN: the result and counters[4..7] of such a step are complete once THIS stream has drained. */
/* Host-overhead helpers of a pipelined step loop (no reference counterpart): copy a step's eight counters to pinned host
 * memory on `on_stream` once `after_stream` has passed its current point and record *done_event there (created on first
 * use when *done_event is NULL; owned by the caller afterwards: mia_event_destroy); wait for such an event on the host or
 * make a stream wait for it.  One call each where a Python host would need five torch calls per step. */
/* Launch thread: the step call (+ its read-back) executed by one worker thread of the library, in submission order, so
 * that the ~65 us of HIP runtime calls a step needs overlap the caller's own per-step work.  mia_letkf_step_submit takes
 * the arguments of mia_letkf_sharded_step_streams_f32 followed by those of mia_letkf_step_readback (host8 NULL: no
 * read-back) and optional timing events (mia_letkf_step_timing_events), and returns a job; mia_letkf_step_join waits until
 * that job's launches are enqueued and returns the step call's status; mia_letkf_step_drain waits for the thread to run
 * dry (call it before a synchronous step call that must not overtake queued ones). */
int mia_letkf_step_submit(const float* X, int64_t G, int m, int k, const float* Yb, const float* d, int64_t P,
                          const double* grid_xyz, const double* obs_xyz, int n_coord, const int32_t* coord_group,
                          const double* gc_c, int n_r, double gc_eps, float inf_factor, float gamma, int method,
                          int p_max_assumed, mia_comm_t* comm, int n_chunks, int phase, float* Xa, int32_t* flags,
                          int32_t* counters, void* ws, size_t ws_bytes, void* stream, void* comm_stream, void* prep_stream,
                          int step_flags, int32_t* host8, void* after_stream, void* on_stream, void** done_event,
                          void* time_start_event, void* time_stop_event, void** job_out);
int mia_letkf_step_join(void* job);
/* The same submission through ONE argument block (a caller that keeps one block per pipeline slot rewrites only what changes from
 * step to step: for a Python caller the 37-argument call above is ~8 us of argument conversion per step, and at config 2 the caller's
 * ~37 us per step, not the GPU, bounded the pipeline).  in_event != NULL: prep_stream first waits for what caller_stream holds
 * (mia_stream_wait_stream(prep_stream, caller_stream, in_event)) -- the inputs are ready.  The fields are mia_letkf_step_submit's
 * arguments, in its order. */
typedef struct mia_step_args {
  const float* X; int64_t G; int32_t m, k; const float* Yb; const float* d; int64_t P;
  const double* grid_xyz; const double* obs_xyz; int32_t n_coord; int32_t coord_group[MIA_MAX_COORD];
  double gc_c[MIA_MAX_RADII]; int32_t n_r; double gc_eps; float inf_factor, gamma; int32_t method, p_max_assumed;
  mia_comm_t* comm; int32_t n_chunks, phase; float* Xa; int32_t* flags; int32_t* counters; void* ws; size_t ws_bytes;
  void* stream; void* comm_stream; void* prep_stream; int32_t step_flags; int32_t* host8; void* after_stream; void* on_stream;
  void** done_event; void* time_start_event; void* time_stop_event;
  void* caller_stream; void** in_event;
} mia_step_args_t;
int mia_letkf_step_submit_args(const mia_step_args_t* a, void** job_out);
/* One step taken at once on the caller's thread through the same block (MIA_STEP_NO_JOIN is ignored): drain of the launch threads,
 * the step call, the counters' read-back (after_stream / on_stream / done_event as in mia_letkf_step_readback) and, when out8 is
 * given, the host's wait for it and the eight counters -- the whole of a cycled filter's step in one call (a Python caller's
 * ~15 us of set-up code and argument conversion in front of the first launch were a fifth of the step's 80 us). */
int mia_letkf_step_run_args(const mia_step_args_t* a, int32_t* out8);
/* Collects a submitted step in one call: mia_letkf_step_join_info(job, batch_n), then waits on the host for *done_event (the event
 * the step's read-back recorded -- the pointer handed to the submission, read after the join), copies the eight counters from host8
 * to out8 and, if consumer_stream_valid, makes consumer_stream wait for that event (work enqueued there afterwards sees the step's
 * result).  Returns the step call's status. */
int mia_letkf_step_collect(void* job, void** done_event, const int32_t* host8, void* consumer_stream, int consumer_stream_valid,
                           int32_t* out8, int* batch_n);
/* Timing events for mia_letkf_step_submit's time_start_event / time_stop_event without a torch.cuda.Event per step: from a pool
 * that is never freed (an event handed back with mia_timing_event_release is reused, not destroyed: a launch thread that still
 * holds it touches a live event).  mia_timing_event_elapsed_ms: hipEventElapsedTime. */
int mia_timing_event_acquire(void** event);
int mia_timing_event_release(void* event);
int mia_timing_event_elapsed_ms(void* start_event, void* stop_event, float* ms);
int mia_letkf_step_drain(void);
int mia_letkf_step_readback(const int32_t* counters, int32_t* host8, void* after_stream, void* on_stream, void** done_event);
int mia_event_synchronize(void* event);
int mia_stream_wait_event(void* stream, void* event);
/* dst_stream waits for what src_stream holds so far; *event_io: NULL on the first call (the event is created and handed back; free
 * it with mia_event_destroy) */
int mia_stream_wait_stream(void* dst_stream, void* src_stream, void** event_io);
int mia_event_destroy(void* event);
/* Host time the library's two launch threads have spent enqueueing since start-up (microseconds; preparation stage and
 * analysis / exchange / read-back stage) and the number of steps handed to them. */
int mia_letkf_step_launch_stats(double* prep_us, double* rest_us, long long* steps);
int mia_comm_set_place_stream(mia_comm_t* comm, void* stream);
/* Direct exchange (csrc/sharded_step.hip, "Direct exchange"): library-owned, peer-mapped result buffers, so that every rank
 * writes its block of the analysis ensemble straight into all peers' (m, k, G) result over its xGMI links -- no ring, no
 * staging, no placement copy.  No reference counterpart (the reference's only distribution is dask on one host,
 * interface/letkf.py:118-131).
 *   mia_comm_peer_alloc   n_slots (<= 8, one per step in flight) result buffers of result_bytes + a fine-grained flag area;
 *                         ipc_handles_out: NULL or (n_slots + 1) x 64 bytes of hipIpcMemHandle_t to hand to the other ranks
 *   mia_comm_peer_open    all_handles = [world][n_slots + 1][64]: every rank's table (host-side all-gather); maps the peers
 *   mia_comm_peer_attach  the same for ranks that share an address space: the peer's buffer pointers and flag area
 *   mia_comm_peer_buffer / _sync_area: this rank's pointers
 *   mia_comm_peer_exchange  the exchange alone (block [rows][b0, b1) of buffer `slot`, counters int32[8] on the device)
 * mia_letkf_sharded_step_streams_f32 takes this route by itself when Xa is mia_comm_peer_buffer(comm, slot) of a
 * communicator whose peers are all mapped (n_chunks is then 1); every rank must pass the same slot.  Failures (IPC not
 * available, a waiter timing out: error bit 2 of counters[3] / [7]) leave RCCL as the route. */
int mia_comm_peer_alloc(mia_comm_t* comm, size_t result_bytes, int n_slots, void* ipc_handles_out);
int mia_comm_peer_open(mia_comm_t* comm, const void* all_handles);
int mia_comm_peer_attach(mia_comm_t* comm, int peer, void* const* result_bufs, void* sync_area);
/* bound of the direct exchange's device-side waits: 2^log2_polls polls of ~1-2 us each (10 .. 30; default 25, about a minute).
 * A waiter that gives up raises error bit 2 of counters[3] / [7] -- the grid always drains. */
int mia_comm_peer_wait_bound(mia_comm_t* comm, int log2_polls);
void* mia_comm_peer_buffer(mia_comm_t* comm, int slot);
void* mia_comm_peer_sync_area(mia_comm_t* comm);
int mia_comm_peer_exchange(mia_comm_t* comm, int slot, int rows, int64_t G, int64_t b0, int64_t b1, int32_t* counters,
                           void* stream);
/* A waiter of the last exchange on `slot` gave up (error bit 2 of counters[3] / [7]): wait again for the peers' flags of THAT
 * exchange and fold the counters once more (enqueued on `stream`; clears error bit 2, which is set again if this wait gives up
 * too).  A late peer costs its lateness, not the run; the caller bounds the number of re-waits. */
int mia_comm_peer_rewait(mia_comm_t* comm, int slot, int32_t* counters, void* stream);
int mia_comm_destroy(mia_comm_t* comm);
const char* mia_comm_last_error(void);
int mia_letkf_sharded_step_workspace_bytes(int64_t G, int m, int k, int64_t P, int n_coord, int world,
                                           int n_chunks, int p_max_assumed, size_t* bytes);
/* The library remembers, per workspace ADDRESS, what the tile lists in it were built for (geometry epochs, MIA_STEP_REUSE_LISTS)
 * and which of its two per-cell count arrays the next bucket build takes.  Call this before a step workspace is freed or handed to
 * another geometry's steps: the next step on that address then starts from scratch (full clear, lists rebuilt), whatever its
 * flags say.  A step in flight is not affected (its decisions travel with it).  Always MIA_OK. */
int mia_letkf_step_workspace_release(void* ws);
/* Launch coalescing (option "step_coalesce", default on): the launch thread puts up to four steps in flight whose preparation has
 * finished into ONE launch of the fused kernel.  mia_letkf_step_join_info is mia_letkf_step_join that also says how many steps shared
 * the step's analysis launch; mia_letkf_step_coalesce_stats counts such launches and the steps in them since the process began. */
int mia_letkf_step_join_info(void* job, int* batch_n);
int mia_letkf_step_coalesce_stats(long long* launches, long long* steps);
/* Diagnostics: host time stamps of the last steps handed to the launch threads (8 per step, ns of the steady clock: submitted,
 * preparation thread begins / has enqueued, analysis thread takes the step / has seen the preparation finished / has enqueued the
 * analysis / the read-back, 0), oldest first.  Returns the number of steps written. */
int mia_debug_step_trace(long long* out, int max_steps);
int mia_letkf_sharded_step_f32(const float* X /* [m][k][G] */, int64_t G, int m, int k,
                               const float* Yb /* [k][P] */, const float* d /* [P] */, int64_t P,
                               const double* grid_xyz /* [G][n_coord] */, const double* obs_xyz /* [P][n_coord] */,
                               int n_coord, const int32_t* coord_group /* host */, const double* gc_c /* host */,
                               int n_r, double gc_eps, float inf_factor, float gamma, int method,
                               int p_max_assumed, mia_comm_t* comm, int n_chunks, int phase,
                               float* Xa /* [m][k][G] */, int32_t* flags, int32_t* counters,
                               void* ws, size_t ws_bytes, void* stream, void* comm_stream);

/* The same step with its three phases on caller-chosen streams, for software-pipelining consecutive steps:
 *   prep_stream  (NULL: = stream)  records, observation index, neighbour lists -- a chain of small launches
 *   stream                         the analysis kernel(s), started once prep_stream has produced the lists
 *   comm_stream                    per-piece all-gather + placement, as above
 * With one analysis stream shared by all steps and one prep_stream per step in flight, the preparation of step i+1
 * runs beside the analysis of step i while the analyses themselves stay in order (two analysis kernels sharing the
 * CUs are slower than one after the other).  step_flags: MIA_STEP_NO_JOIN = do not make `stream` wait for the
 * exchange at the end (the next step's analysis need not wait for this step's all-gather): the result and
 * counters[4..7] are then complete once comm_stream has drained, which the caller orders itself. */
#define MIA_STEP_NO_JOIN 1
/* MIA_STEP_WS_CLEAN: the caller guarantees that `ws` was last used by a COMPLETED step of this entry with the same sizes
 * and has not been written since: the observation index then needs no fill launch (its kernels leave the header as they
 * found it).  Never set it for a fresh or recycled allocation. */
#define MIA_STEP_WS_CLEAN 4
#define MIA_STEP_TILE_EXTRA(n) (((n) & 7) << 4) /* tile route: n more row blocks of sixteen union slots per tile (after a step
                                                   reported unions that did not fit; beyond what the ensemble size allows the
                                                   step takes the per-point lists) */
#define MIA_STEP_FRESH_BOX 0x400  /* tile route: recompute the observations' bounding box this step (a step reported error bit 8:
                                    an observation outside the box its workspace held, or other radii) */
#define MIA_STEP_SCAN_INDEX 0x800 /* tile route: scan-based observation index instead of fixed-capacity buckets (a step reported
                                    error bit 16: a cell with more observations than a bucket holds) */
#define MIA_STEP_REUSE_LISTS 0x1000 /* tile route, geometry epoch: use the tile lists `ws` already holds -- built by an earlier
                                    * COMPLETED step on this workspace with the same grid / observation coordinates, radii, eps,
                                    * block and tile format -- and rebuild only the split records.  The reference recomputes the
                                    * localisation on every call (gaspari_cohn.py:97-136); results are identical when the
                                    * geometry is.  counters[0], [1] stay 0 */
#define MIA_STEP_KEEP_LISTS 0x4000 /* tile route: build this step's tile lists IN MEMORY (a later step of the same geometry epoch will
                                    * ask for them with MIA_STEP_REUSE_LISTS) rather than inside the analysis wavefronts (option
                                    * "tile_fused"); the results are the same bit for bit */
#define MIA_STEP_NO_GATHER 0x2000 /* world > 1: the analysis STAYS block-sharded (the reference's dask chunks along `grid`,
                                   * interface/letkf.py:118-131): Xa is this rank's block, (m k, block length), block = grid
                                   * points [rank n, min(G, (rank + 1) n)) with n = ceil(G / world); nothing is exchanged,
                                   * counters[4..7] stay zero.  `comm` may be a partition-only communicator
                                   * (mia_comm_create_partition) */
#define MIA_STEP_NO_TILE_LISTS 8 /* per-point lists even where the tile route would apply (after a step reported tiles whose
                                   union did not fit: counters[1] != 0 with counters[0] <= p_max_assumed) */
int mia_letkf_sharded_step_streams_f32(const float* X, int64_t G, int m, int k,
                                       const float* Yb, const float* d, int64_t P,
                                       const double* grid_xyz, const double* obs_xyz,
                                       int n_coord, const int32_t* coord_group /* host */, const double* gc_c /* host */,
                                       int n_r, double gc_eps, float inf_factor, float gamma, int method,
                                       int p_max_assumed, mia_comm_t* comm, int n_chunks, int phase,
                                       float* Xa, int32_t* flags, int32_t* counters,
                                       void* ws, size_t ws_bytes, void* stream, void* comm_stream,
                                       void* prep_stream, int step_flags);

/* Profiling hook (one shot, per calling thread): the NEXT step enqueued by mia_letkf_sharded_step*_f32 records
 * `start_event` on the analysis stream immediately before its analysis kernel(s) -- after the stream has waited for
 * the preparation -- and `stop_event` immediately after them.  Both are hipEvent_t created by the caller with timing
 * enabled; NULL, NULL disarms.  This is how bench.py measures the dominant kernel INSIDE its timed loop. */
int mia_letkf_step_timing_events(void* start_event, void* stop_event);

#ifdef __cplusplus
}
#endif
#endif /* MIA_LETKF_H */

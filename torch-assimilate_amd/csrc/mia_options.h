// Route options shared by the translation units (set through mia_set_option, include/mia_letkf.h).
#pragma once

enum {
  MIA_OPT_CHEB_DMAX = 0,     // largest Chebyshev degree the matfun kernels accept before declining a point (3 .. 62)
  MIA_OPT_CHEB_TABLE,        // coefficients from the per-device table (1) or computed in the kernel (0)
  MIA_OPT_CHEB_ROWBATCH,     // many state rows: 16-row MFMA batches (1) or the row-by-row path (0)
  MIA_OPT_CHEB_BIG,          // 64 < k <= 128 with > 64 local observations: two-rows-per-lane matfun kernel (1) or eigensolver (0)
  MIA_OPT_TILE,              // sixteen grid points per wavefront (letkf_tile.hip) where the shape allows (1) or one (0)
  MIA_OPT_TILE_SPLIT,        // the tile kernel's products as split half-precision MFMAs (1) or f32 MFMAs (0)
  MIA_OPT_LOCALIZE_QUAD,     // neighbour lists of short lists: four lanes per grid point (1) or one (0)
  MIA_OPT_STEP_HOSTWAIT,     // steps in flight: the launch thread waits for a step's preparation on the host (1) or the analysis stream does (0)
  MIA_OPT_STEP_LAZY_SORT,    // step driver: observation index without its per-cell sort when the tile kernel takes the analysis (1) / always sorted (0)
  MIA_OPT_SEGMENT_SIGNAL,    // step driver with several pieces: one segmented launch (1) or one launch + event per piece (0)
  MIA_OPT_TILE_LISTS,        // step driver: tile-shaped lists + split records + letkf_tile2_kernel where the shape allows (1) or the per-point lists (0)
  MIA_OPT_BUCKET_INDEX,      // step driver, tile route: observations binned into fixed-capacity cell buckets by ONE kernel over the cell grid the
                             // workspace holds (1) or bounding box + count + scan + scatter every step (0)
  MIA_OPT_TILE_PAIR,         // tile route, unions of more than 32 slots: two wavefronts per tile (letkf_tile2p.hip) (1) or one (0)
  MIA_OPT_TILE_FUSED,        // step driver, tile route, unions of at most 32 slots: the analysis wavefronts localise their
                             // tiles themselves (letkf_tile2f.hip: no list kernel, no lists in memory) (1) or lists first (0)
  MIA_OPT_STEP_COALESCE,     // steps in flight on the fused kernel: 0 (default): every step is launched on its own the moment it is ready;
                             // 1 .. 4: the launch thread keeps at most this many launches of its own running and puts the steps that
                             // become ready meanwhile -- up to four -- into ONE launch (measured slower per step: sharded_step.hip)
  MIA_OPT_COUNT_
};

namespace mia {
int option(int id);
// snapshot of all options (MIA_OPT_COUNT_ ints) / run this thread under a snapshot (nullptr: the process-wide values again)
void option_snapshot(int* out);
void option_override(const int* snapshot);
}

// Timing / accuracy experiments of tools/ (phase skipping, tolerances, alternative launch shapes) read the environment --
// in builds with -DMIA_EXPERIMENTS only (MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS python tools/...).  The default build compiles
// these to constants: no environment variable changes what the shipped library computes.
#ifdef MIA_EXPERIMENTS
#include <cstdlib>
#define MIA_EXP_SET(var, name, conv) do { if (const char* e_ = std::getenv(name)) (var) = conv(e_); } while (0)
#define MIA_EXP_FLAG(name) (std::getenv(name) != nullptr)
#else
#define MIA_EXP_SET(var, name, conv) do { } while (0)
#define MIA_EXP_FLAG(name) false
#endif

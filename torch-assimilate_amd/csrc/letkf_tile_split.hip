// The split-precision instantiations of the sixteen-points-per-wavefront kernel (letkf_tile.hip, template parameter SPL):
// a translation unit of their own so that they compile beside the f32 instantiations.
#define MIA_TILE_TU_SPLIT 1
#include "letkf_tile.hip"

// screened lane kernels for Bingham targets at d = 11 .. 13 (see gsss_fast_bingham_lane.h)
#include "gsss_fast_bingham_lane.h"
namespace gsss {
template int lane_bingham_wide<11>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
template int lane_bingham_wide<12>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
template int lane_bingham_wide<13>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
}

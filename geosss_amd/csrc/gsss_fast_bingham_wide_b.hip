// screened lane kernels for Bingham targets at d = 14 .. 16 (see gsss_fast_bingham_lane.h)
#include "gsss_fast_bingham_lane.h"
namespace gsss {
template int lane_bingham_wide<14>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
template int lane_bingham_wide<15>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
template int lane_bingham_wide<16>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
}

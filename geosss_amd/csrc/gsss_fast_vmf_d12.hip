// screened lane kernels for vMF mixtures at d = 12, one chain per lane (see gsss_fast_vmf_lane.h: lane_vmf_wide)
#include "gsss_fast_vmf_lane.h"
namespace gsss {
template int lane_vmf_wide<12>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
}

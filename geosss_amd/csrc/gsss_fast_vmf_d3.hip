// lane-per-chain vMF-mixture kernels at d = 3 (see gsss_fast_vmf_lane.h)
#include "gsss_fast_vmf_lane.h"
namespace gsss {
template int lane_vmf<3>(const TargetBlock &, const RunBlock &, bool, FastProbe *, hipStream_t);
}

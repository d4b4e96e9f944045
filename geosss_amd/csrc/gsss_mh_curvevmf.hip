// RWMH / spherical HMC kernels for the CurveVmf target (gsss_mh.h), every vector layout and draw source
#include "gsss_launch.h"
#include "gsss_mh.h"

namespace gsss {
#define GSSS_MH_CASE_CurveVmf(ID, V, NAME) \
    case ID:                        \
        return mh_dispatch<V, CurveVmf>(draws, sampler, tb, rb, mb, st);
GSSS_DEFINE_MH_LAUNCHER(CurveVmf)
}  // namespace gsss

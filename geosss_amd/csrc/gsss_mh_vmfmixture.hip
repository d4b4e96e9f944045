// RWMH / spherical HMC kernels for the VmfMixture target (gsss_mh.h), every vector layout and draw source
#include "gsss_launch.h"
#include "gsss_mh.h"

namespace gsss {
#define GSSS_MH_CASE_VmfMixture(ID, V, NAME) \
    case ID:                        \
        return mh_dispatch<V, VmfMixture>(draws, sampler, tb, rb, mb, st);
GSSS_DEFINE_MH_LAUNCHER(VmfMixture)
}  // namespace gsss

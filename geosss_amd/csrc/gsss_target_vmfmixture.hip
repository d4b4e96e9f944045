// Kernel instantiations for the VmfMixture target: every vector layout x {Philox, replay, numpy} draws.
#include "gsss_launch.h"

namespace gsss {
#define GSSS_RUN_CASE_VmfMixture(ID, V, NAME) \
    case ID:                         \
        return draws == kDrawsReplay ? do_run<V, VmfMixture, ReplayDraws>(tb, rb, st) \
               : draws == kDrawsNumpy ? do_run<V, VmfMixture, NumpyDraws>(tb, rb, st) \
                                      : do_run<V, VmfMixture, PhiloxDraws>(tb, rb, st);
#define GSSS_LOGPROB_CASE_VmfMixture(ID, V, NAME) \
    case ID:                             \
        return do_logprob<V, VmfMixture>(tb, x, n, out, grad, st);
GSSS_DEFINE_TARGET_LAUNCHERS(VmfMixture)
}  // namespace gsss

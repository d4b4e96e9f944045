// Kernel instantiations for the CurveVmf target: every vector layout x {Philox, replay, numpy} draws.
#include "gsss_launch.h"

namespace gsss {
#define GSSS_RUN_CASE_CurveVmf(ID, V, NAME) \
    case ID:                         \
        return draws == kDrawsReplay ? do_run<V, CurveVmf, ReplayDraws>(tb, rb, st) \
               : draws == kDrawsNumpy ? do_run<V, CurveVmf, NumpyDraws>(tb, rb, st) \
                                      : do_run<V, CurveVmf, PhiloxDraws>(tb, rb, st);
#define GSSS_LOGPROB_CASE_CurveVmf(ID, V, NAME) \
    case ID:                             \
        return do_logprob<V, CurveVmf>(tb, x, n, out, grad, st);
GSSS_DEFINE_TARGET_LAUNCHERS(CurveVmf)
}  // namespace gsss

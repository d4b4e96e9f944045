// Kernel instantiations for the Bingham target: every vector layout x {Philox, replay, numpy} draws.
#include "gsss_launch.h"

namespace gsss {
#define GSSS_RUN_CASE_Bingham(ID, V, NAME) \
    case ID:                         \
        return draws == kDrawsReplay ? do_run<V, Bingham, ReplayDraws>(tb, rb, st) \
               : draws == kDrawsNumpy ? do_run<V, Bingham, NumpyDraws>(tb, rb, st) \
                                      : do_run<V, Bingham, PhiloxDraws>(tb, rb, st);
#define GSSS_LOGPROB_CASE_Bingham(ID, V, NAME) \
    case ID:                             \
        return do_logprob<V, Bingham>(tb, x, n, out, grad, st);
GSSS_DEFINE_TARGET_LAUNCHERS(Bingham)
}  // namespace gsss

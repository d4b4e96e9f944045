// GSSS_MODE_FAST instantiations of the group-speculative curve-vMF kernel (gsss_curvespec.h).
#include "gsss_curvespec.h"

namespace gsss {

// L lanes per chain, L speculative single-precision tries per batch (gsss_curvespec.h): packed ensembles on the Philox
// or replay stream, screening on.  Statistics builds, the numpy stream and one-wavefront-per-chain placement stay with
// the lane / cooperative kernels below.
int launch_curvespec(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, bool lane, hipStream_t st)
{
#define GSSS_SPEC(LL, QQ)                                                                                   \
    do {                                                                                                    \
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<%d, %d, %d>", LL, QQ, tb.k <= 10 ? 10 : 17);          \
        return tb.k <= 10 ? do_curvespec<LL, QQ, 10>(tb, rb, replay, st) : do_curvespec<LL, QQ, 17>(tb, rb, replay, st); \
    } while (0)
    if (tb.d <= 16) GSSS_SPEC(4, 1);
    if (tb.d <= 64) GSSS_SPEC(16, 1);
    if (tb.d <= 128) GSSS_SPEC(16, 2);
    if (tb.d <= 192) GSSS_SPEC(16, 3);
    GSSS_SPEC(16, 4);
#undef GSSS_SPEC
}

}  // namespace gsss

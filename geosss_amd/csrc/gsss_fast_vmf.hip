// GSSS_MODE_FAST instantiations for von Mises-Fisher mixtures: (d, K) pairs of the benchmark
// configurations (BASELINE.json) and the golden fixtures.
#include "gsss_fast.h"

namespace gsss {

#define GSSS_FAST_VMF_SHAPES(X) X(3, 1) X(3, 2) X(3, 3) X(3, 4) X(3, 5) X(3, 10) X(4, 4) X(10, 5)

int launch_fast_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, bool probe, hipStream_t st)
{
#define GSSS_CASE(D, K)                                            \
    if (tb.d == D && tb.k == K) {                                  \
        if (probe) return GSSS_OK;                                 \
        return do_fast<D, FastVmf<D, K>>(tb, rb, replay, st);      \
    }
    GSSS_FAST_VMF_SHAPES(GSSS_CASE)
#undef GSSS_CASE
    if (!probe) set_error("fast mode is not built for a vMF mixture with d=%d, K=%d", tb.d, tb.k);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

// Screened lane kernels for Bingham / BinghamFisher targets at d = 11 .. 16, one chain per lane (round 4): the compact diagonal
// target of the paper's eigenbasis experiments and the general one (dense A, linear term).  Served: packed ensembles on the
// library stream with the screen on (or verified); everything else stays with the cooperative kernels.
#pragma once
#include "gsss_screen.h"

namespace gsss {

template <int D>
int lane_bingham_wide(const TargetBlock &tb, const RunBlock &rb, FastProbe *probe, hipStream_t st)
{
    const bool compact = tb.k == 1;  // a diagonal A, no linear term
    if (probe) {
        if (compact) GSSS_PROBE(false, "screened_kernel<%d, ScreenBinghamDiag<%d>>", D, D);
        GSSS_PROBE(false, "screened_kernel<%d, ScreenBingham<%d>>", D, D);
    }
    if (compact) return do_screened_run<D, ScreenBinghamDiag<D>, false>(tb, rb, st);
    return do_screened_run<D, ScreenBingham<D>, false>(tb, rb, st);
}
#define GSSS_BINGHAM_WIDE_DIMS(X) X(11) X(12) X(13) X(14) X(15) X(16)
#define GSSS_DECLARE_WIDE(D) extern template int lane_bingham_wide<D>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
GSSS_BINGHAM_WIDE_DIMS(GSSS_DECLARE_WIDE)
#undef GSSS_DECLARE_WIDE

}  // namespace gsss

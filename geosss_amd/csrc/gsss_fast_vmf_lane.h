// Lane-per-chain fast kernels for von Mises-Fisher mixtures, any K <= 16 components at d = 3 .. 10.
//
// Kernels are built per dimension and per component-count BUCKET KC: a mixture of K components runs the
// kernel of the smallest bucket >= K (FastVmf::stage pads the surplus components with mu = 0, logc = log 0:
// exact zeros in every sum, so the bucket does not change a single bit of the chain).  Screened kernels
// (gsss_screen.h): buckets 3, 4, 6, 10, 16; the all-double fallback and the one-wavefront-per-chain kernels
// (gsss_fast.h): buckets 4 and 16.  One translation unit per dimension (compile time).
#pragma once
#include "gsss_screen.h"

namespace gsss {

constexpr double kScreenMaxKappa = 4000.0;  // margin ~ 2e-6 kappa: beyond this a few per cent of the tries stay undecided

// the screened kernel unless the caller forces all-double arithmetic, the ensemble is small (one wavefront per
// chain), numpy's stream is asked for, or the concentration is so large that the margin would leave tries undecided
template <int D, int KS, int KF>
static int run_lane_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    const bool screen = rb.screen && !rb.spread && rb.rng_state == nullptr && tb.scale <= kScreenMaxKappa;
    // numpy's stream for a packed ensemble: the screened kernel too (round 4; K <= 10 -- the widest bucket stays all-double)
    if constexpr (KS <= 10) {
        if (rb.screen && !rb.spread && rb.rng_state != nullptr && !replay && tb.scale <= kScreenMaxKappa)
            return do_screened_numpy<D, ScreenVmf<D, KS>>(tb, rb, st);
    }
    if (!screen) return do_fast<D, FastVmf<D, KF>>(tb, rb, replay, st);
    return replay ? do_screened_run<D, ScreenVmf<D, KS>, true>(tb, rb, st) : do_screened_run<D, ScreenVmf<D, KS>, false>(tb, rb, st);
}

template <int D>
int lane_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st)
{
    const int ks = tb.k <= 3 ? 3 : (tb.k <= 4 ? 4 : (tb.k <= 6 ? 6 : (tb.k <= 10 ? 10 : 16)));
    const int kf = tb.k <= 4 ? 4 : 16;
    if (probe) {
        if (rb.screen && tb.scale <= kScreenMaxKappa) GSSS_PROBE(true, "screened_kernel<%d, ScreenVmf<%d, %d>>", D, D, ks);
        GSSS_PROBE(true, "fast_kernel<%d, FastVmf<%d, %d>>", D, D, kf);
    }
    switch (ks) {
    case 3: return run_lane_vmf<D, 3, 4>(tb, rb, replay, st);
    case 4: return run_lane_vmf<D, 4, 4>(tb, rb, replay, st);
    case 6: return run_lane_vmf<D, 6, 16>(tb, rb, replay, st);
    case 10: return run_lane_vmf<D, 10, 16>(tb, rb, replay, st);
    default: return run_lane_vmf<D, 16, 16>(tb, rb, replay, st);
    }
}

// d = 11 .. 16 (round 4): the screened lane kernel alone, one chain per lane (screen_parks is false there), mixtures of up to
// ten components in the buckets 3, 6 and 10.  What it does not serve -- replayed and numpy streams, one-wavefront placement, the
// all-double variant, wider mixtures, kappa beyond the screen's reach -- stays with the cooperative kernels (the caller checks).
template <int D>
int lane_vmf_wide(const TargetBlock &tb, const RunBlock &rb, FastProbe *probe, hipStream_t st)
{
    const int ks = tb.k <= 3 ? 3 : (tb.k <= 6 ? 6 : 10);
    if (probe) GSSS_PROBE(false, "screened_kernel<%d, ScreenVmf<%d, %d>>", D, D, ks);
    if (ks == 3) return do_screened_run<D, ScreenVmf<D, 3>, false>(tb, rb, st);
    if (ks == 6) return do_screened_run<D, ScreenVmf<D, 6>, false>(tb, rb, st);
    return do_screened_run<D, ScreenVmf<D, 10>, false>(tb, rb, st);
}
inline bool lane_wide_serves(const RunBlock &rb, bool replay)
{
    return rb.screen != 0 && !rb.spread && rb.rng_state == nullptr && !replay;
}
#define GSSS_VMF_WIDE_DIMS(X) X(11) X(12) X(13) X(14) X(15) X(16)
#define GSSS_DECLARE_WIDE(D) extern template int lane_vmf_wide<D>(const TargetBlock &, const RunBlock &, FastProbe *, hipStream_t);
GSSS_VMF_WIDE_DIMS(GSSS_DECLARE_WIDE)
#undef GSSS_DECLARE_WIDE

#define GSSS_VMF_LANE_DIMS(X) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#define GSSS_DECLARE(D) extern template int lane_vmf<D>(const TargetBlock &, const RunBlock &, bool, FastProbe *, hipStream_t);
GSSS_VMF_LANE_DIMS(GSSS_DECLARE)
#undef GSSS_DECLARE

}  // namespace gsss

// GSSS_MODE_FAST instantiations for von Mises-Fisher mixtures: (d, K) pairs of the benchmark
// configurations (BASELINE.json) and the golden fixtures.
#include "gsss_screen.h"

namespace gsss {

constexpr double kScreenMaxKappa = 4000.0;  // margin ~ 2e-6 kappa: beyond this a few per cent of the tries stay undecided

#define GSSS_FAST_VMF_SHAPES(X) \
    X(3, 1) X(3, 2) X(3, 3) X(3, 4) X(3, 5) X(3, 6) X(3, 8) X(3, 10) X(4, 4) X(5, 5) X(10, 3) X(10, 5) X(10, 10)

// lane kernels: the screened kernel unless the caller forces all-double arithmetic, the ensemble is small (one wavefront per
// chain) or the concentration is so large that the screen's margin would leave most tries undecided
template <int D, int K>
static int run_lane_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    const bool screen = rb.screen && !rb.spread && rb.rng_state == nullptr && tb.scale <= kScreenMaxKappa;
    if (!screen) return do_fast<D, FastVmf<D, K>>(tb, rb, replay, st);
    return replay ? do_screened_run<D, ScreenVmf<D, K>, true>(tb, rb, st) : do_screened_run<D, ScreenVmf<D, K>, false>(tb, rb, st);
}

int launch_fast_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st)
{
#define GSSS_CASE(D, K)                                            \
    if (tb.d == D && tb.k == K) {                                  \
        if (probe) {                                               \
            if (rb.screen && tb.scale <= kScreenMaxKappa) GSSS_PROBE(true, "screened_kernel<%d, ScreenVmf<%d, %d>>", D, D, K); \
            GSSS_PROBE(true, "fast_kernel<%d, FastVmf<%d, %d>>", D, D, K); \
        }                                                          \
        return run_lane_vmf<D, K>(tb, rb, replay, st);             \
    }
    GSSS_FAST_VMF_SHAPES(GSSS_CASE)
#undef GSSS_CASE
    // larger d: lanes cooperate on one chain (K = 3, 5 or 10 components)
#define GSSS_COOP(K)                                                                                          \
    if (tb.k == K && tb.d > 10 && tb.d <= 256) {                                                              \
        if (probe)                                                                                            \
            GSSS_PROBE(false, "coopfast_kernel<CoopVec<%d, 4>, CoopVmf<%d>>", tb.d <= 16 ? 4 : (tb.d <= 64 ? 16 : 64), K); \
        if (tb.d <= 16) return do_coopfast<CoopVec<4, 4>, CoopVmf<CoopVec<4, 4>, K>>(tb, rb, replay, st);     \
        if (tb.d <= 64) return do_coopfast<CoopVec<16, 4>, CoopVmf<CoopVec<16, 4>, K>>(tb, rb, replay, st);   \
        return do_coopfast<CoopVec<64, 4>, CoopVmf<CoopVec<64, 4>, K>>(tb, rb, replay, st);                   \
    }
    GSSS_COOP(3) GSSS_COOP(5) GSSS_COOP(10)
#undef GSSS_COOP
    if (!probe) set_error("fast mode is not built for a vMF mixture with d=%d, K=%d", tb.d, tb.k);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

// GSSS_MODE_FAST instantiations for curve-vMF targets (10 knots, the reference's brownian_curve default).
#include "gsss_screen.h"
#include "gsss_spec64.h"

namespace gsss {

// d = 3, 6, ..., 24 is the reference's own sweep (sh/submit_job_curve_varying_ndim.sh:11); d = 10 its default
#define GSSS_FAST_CURVE_DIMS(X) X(3) X(6) X(9) X(10) X(12) X(15) X(18) X(21) X(24)

int launch_fast_curve(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st)
{
    // (knot counts: what the all-double / one-wavefront kernels below cover too, so that a shape is either served in every
    // placement and variant or in none)
    // d >= 4: on S^2 the Philox stream draws the tangent as one angle, which only the lane kernels do.  (Round 3: from d = 4,
    // not 9 -- 10^5 chains, 10^9 chain-steps/s: d = 6 2.97 (screened lane kernel) -> 4.74, d = 4, 5, 7, 8 1.05 (sixteen-lane
    // cooperative kernel) -> 4.6 .. 4.8.)
    const bool spec = rb.screen && !rb.spread && rb.rng_state == nullptr && tb.k >= 2 &&
                      tb.k <= (tb.d > 64 ? 17 : 16) && tb.d >= 4 && tb.d <= 256;
    // lane-per-chain kernels: the listed dimensions, any curve of 2 .. 10 knots (built for 10; FastCurve pads)
#define GSSS_CASE(D)                                                \
    if (tb.d == D && tb.k >= 2 && tb.k <= 10) {                     \
        if (spec) return launch_curvespec(tb, rb, replay, probe, true, st); \
        const bool screen = rb.screen && !rb.spread && rb.rng_state == nullptr; \
        if (probe) {                                                \
            if (rb.screen) GSSS_PROBE(true, "screened_kernel<%d, ScreenCurve<%d, 10>>", D, D); \
            GSSS_PROBE(true, "fast_kernel<%d, FastCurve<%d, 10>>", D, D); \
        }                                                           \
        if (!screen) return do_fast<D, FastCurve<D, 10>>(tb, rb, replay, st); \
        return replay ? do_screened_run<D, ScreenCurve<D, 10>, true>(tb, rb, st) : do_screened_run<D, ScreenCurve<D, 10>, false>(tb, rb, st); \
    }
    GSSS_FAST_CURVE_DIMS(GSSS_CASE)
#undef GSSS_CASE
    if (spec) return launch_curvespec(tb, rb, replay, probe, false, st);
    // 64 < d <= 256, up to 17 knots: one chain per wavefront, four speculative tries per iteration
    if (tb.d > 64 && tb.d <= 256 && tb.k >= 2 && tb.k <= 17) {
        if (probe) GSSS_PROBE(false, "curve64_kernel<%d>", tb.k <= 11 ? 12 : 20);
        return tb.k <= 11 ? do_curve64<12>(tb, rb, replay, st) : do_curve64<20>(tb, rb, replay, st);
    }
    // every other d <= 64 (and 11 .. 16 knots at the lane dimensions): 16 lanes cooperate on one chain
    if (tb.d >= 3 && tb.d <= 64 && tb.k >= 2 && tb.k <= 16) {
        if (probe) GSSS_PROBE(false, "coopfast_kernel<CoopVec<16, 4>, CoopCurve<%d>>", tb.k <= 10 ? 10 : 16);
        if (tb.k <= 10) return do_coopfast<CoopVec<16, 4>, CoopCurve<CoopVec<16, 4>, 10>>(tb, rb, replay, st);
        return do_coopfast<CoopVec<16, 4>, CoopCurve<CoopVec<16, 4>, 16>>(tb, rb, replay, st);
    }
    if (tb.k >= 2 && tb.k <= 10 && tb.d > 256 && tb.d <= 512) {
        if (probe) GSSS_PROBE(false, "coopfast_kernel<CoopVec<64, 8>, CoopCurve<10>>");
        return do_coopfast<CoopVec<64, 8>, CoopCurve<CoopVec<64, 8>, 10>>(tb, rb, replay, st);
    }
    if (!probe) set_error("fast mode is not built for a curve-vMF target with d=%d, %d knots", tb.d, tb.k);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

// GSSS_MODE_FAST instantiations of the group-speculative curve-vMF kernel (gsss_curvespec.h).
#include "gsss_curvespec.h"

namespace gsss {

// L lanes per chain, L speculative single-precision tries per batch (gsss_curvespec.h): packed ensembles on the Philox
// or replay stream, screening on.  Statistics builds, the numpy stream and one-wavefront-per-chain placement stay with
// the lane / cooperative kernels below.
int launch_curvespec(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, bool lane, hipStream_t st)
{
    // L lanes per chain, Q component quads per lane (d <= 4 Q L), kernels built for 10 and for 17 knots.  Measured at 10^5
    // chains (tools/bench_curve_sweep.py, 10^9 chain-steps/s): d = 17 .. 32 <4,2> 2.6 against <16,1> 1.8; d = 33 .. 48 <4,3>
    // 2.2 / 1.8; d = 49 .. 64 <4,4> 2.1 / 1.8 -- the per-step serial work is repeated in 4 instead of 16 lanes, and sixteen
    // groups share a wavefront.  (<4,4> with 17 knots spills registers: those shapes stay with <16,1>.)
#define GSSS_SPEC(LL, QQ)                                                                                   \
    do {                                                                                                    \
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<%d, %d, %d>", LL, QQ, tb.k <= 10 ? 10 : 17);          \
        return tb.k <= 10 ? do_curvespec<LL, QQ, 10>(tb, rb, replay, st) : do_curvespec<LL, QQ, 17>(tb, rb, replay, st); \
    } while (0)
    // d <= 16, <= 10 knots: TWO lanes per chain with eight components each were measured in round 3 (32 chains share a
    // wavefront's per-step serial work instead of 16) and LOST: 25.6 against 20.7 ms per 10^8 chain-steps at d = 10 -- batches of
    // two speculative tries need 3.9 instead of 2.3 rounds of the single-precision curve evaluation per step, and eight
    // components per lane only fit three wavefronts per SIMD with u parked in LDS and 36 B of scratch.  GSSS_CURVE_L2=1 runs
    // them (parity-tested: the kernel is generic in L), the default stays with four-lane groups.
    if (tb.d <= 16 && tb.k <= 10) {
        static const bool two = [] {
            const char *e = getenv("GSSS_CURVE_L2");
            return e && e[0] == '1';
        }();
        if (two) {
            if (probe) GSSS_PROBE(lane, "curvespec_kernel<2, 2, 10>");
            return do_curvespec<2, 2, 10>(tb, rb, replay, st);
        }
    }
    // Round 5: dimensions that miss a whole number of quads per lane by at most one component per lane run the UNEVEN layout --
    // Q = 3 quads and one tail slot per lane (gsss_curvespec.h, R = 1) -- in the three-wavefront register class of the three-quad
    // builds instead of four quads at two wavefronts.  Measured, ms per 10^8 chain-steps (profiles/r05_ab_curve_tail.log):
    //   d = 49 .. 52   <4, 3, 10, +1>   d = 50 (BASELINE cfg4): 32.50 -> 27.23 (+19 %)
    //   d = 97 .. 104  <8, 3, 10, +1>   d = 100: 52.15 -> 44.01 (+18 %)
    //   d = 193 .. 208 <16, 3, 10, +1>  d = 200 (cfg4): 94.03 -> 80.45 (+17 %) -- with the tail slot of the tangent kept in a register:
    //                                   parked in LDS like the quads' the workgroup needs 55.8 KB, two workgroups per CU under a
    //                                   kernel built for three wavefronts per SIMD: 106.7 ms
    // GSSS_CURVE_TAIL=0 turns the layout off (A/B).  Plain and replayed launches of curves of <= 10 knots; statistics builds keep
    // the even layouts.
    if (tb.k <= 10 && (probe || rb.stats == nullptr)) {
        const char *env_tail = getenv("GSSS_CURVE_TAIL");  // (read per launch: tests switch it)
        const int tail = env_tail ? atoi(env_tail) : 1;
#define GSSS_SPEC_TAIL(LL)                                                                     \
    do {                                                                                       \
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<%d, 3, 10, +1>", LL);                    \
        return do_curvespec<LL, 3, 10, 1>(tb, rb, replay, st);                                 \
    } while (0)
        if (tail >= 1 && tb.d > 48 && tb.d <= 52) GSSS_SPEC_TAIL(4);
        if (tail >= 1 && tb.d > 96 && tb.d <= 104) GSSS_SPEC_TAIL(8);
        if (tail >= 1 && tb.d > 192 && tb.d <= 208) GSSS_SPEC_TAIL(16);
#undef GSSS_SPEC_TAIL
    }
    if (tb.d <= 16) GSSS_SPEC(4, 1);
    if (tb.d <= 32) GSSS_SPEC(4, 2);
    if (tb.d <= 48) GSSS_SPEC(4, 3);
    if (tb.d <= 64 && tb.k <= 10) {
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<4, 4, 10>");
        return do_curvespec<4, 4, 10>(tb, rb, replay, st);
    }
    if (tb.d <= 64) GSSS_SPEC(16, 1);
    // d = 65 .. 128, <= 10 knots: eight-lane groups (1.6 / 1.4 against 1.2 for <16,2>)
    if (tb.d <= 96 && tb.k <= 10) {
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<8, 3, 10>");
        return do_curvespec<8, 3, 10>(tb, rb, replay, st);
    }
    if (tb.d <= 128 && tb.k <= 10) {
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<8, 4, 10>");
        return do_curvespec<8, 4, 10>(tb, rb, replay, st);
    }
    if (tb.d <= 128) GSSS_SPEC(16, 2);
    if (tb.d <= 192) GSSS_SPEC(16, 3);
    GSSS_SPEC(16, 4);
#undef GSSS_SPEC
}

}  // namespace gsss

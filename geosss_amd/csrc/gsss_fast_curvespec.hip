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
    // Round 5: dimensions that miss a whole number of quads per lane by at most two components per lane run an UNEVEN layout -- Q
    // quads and R = 1 or 2 tail slots per lane (gsss_curvespec.h) -- in the register class of the Q-quad build instead of the
    // (Q + 1)-quad one (four quads: two wavefronts per SIMD instead of three).  Measured, ms per 10^8 chain-steps against the even
    // layout (profiles/r05_ab_curve_tail.log):
    //   <4, 1, 10, +1>  d = 17 .. 20    d = 18: 20.84 -> 19.53 (+6.7 %)      <4, 1, 10, +2>  d = 21 .. 24    d = 24 (bench): 20.74 -> 20.05 (+3.4 %)
    //   <4, 2, 10, +1>  d = 33 .. 36    d = 34: 23.80 -> 22.20 (+7.2 %)      <4, 2, 10, +2>  d = 37 .. 40    d = 38: 24.05 -> 23.12 (+4.0 %)
    //   <4, 3, 10, +1>  d = 49 .. 52    d = 50 (BASELINE cfg4): 31.40 -> 25.29 (+24 %)
    //   <4, 3, 10, +2>  d = 53 .. 56    d = 54: 31.47 -> 26.47 (+19 %) -- two tail slots stay in registers: parked in LDS the workgroup is
    //                                   54.1 KB and the CU holds two of them (32.2 ms)
    //   <8, 3, 10, +1>  d = 97 .. 104   d = 100: 52.07 -> 43.64 (+19 %)      <8, 3, 10, +2>  d = 105 .. 112  d = 108: 52.43 -> 45.06 (+16 %)
    //   <16, 3, 10, +1> d = 193 .. 208  d = 200 (cfg4): 93.78 -> 74.45 (+26 %) -- its ONE tail slot in a register for the same reason
    //   (<16, 3, 10, +2>, d = 209 .. 224: 55 KB of LDS whatever is parked, two workgroups per CU: 93.9 -> 98.8 ms; not built)
    // GSSS_CURVE_TAIL=0 turns the layouts off (A/B).  Plain and replayed launches of curves of <= 10 knots; statistics builds keep
    // the even layouts.
    if (tb.k <= 10 && (probe || rb.stats == nullptr)) {
        const char *env_tail = getenv("GSSS_CURVE_TAIL");  // (read per launch: tests switch it)
        const int tail = env_tail ? atoi(env_tail) : 1;
#define GSSS_SPEC_TAIL(LL, QQ, RR)                                                             \
    do {                                                                                       \
        if (probe) GSSS_PROBE(lane, "curvespec_kernel<%d, %d, 10, +%d>", LL, QQ, RR);          \
        return do_curvespec<LL, QQ, 10, RR>(tb, rb, replay, st);                               \
    } while (0)
        // <L, Q, +R> holds 4 Q L + R L components: four-lane groups at d <= 64, eight at 65 .. 128, sixteen beyond
        if (tail >= 1 && tb.d > 16 && tb.d <= 20) GSSS_SPEC_TAIL(4, 1, 1);
        if (tail >= 1 && tb.d > 20 && tb.d <= 24) GSSS_SPEC_TAIL(4, 1, 2);
        if (tail >= 1 && tb.d > 32 && tb.d <= 36) GSSS_SPEC_TAIL(4, 2, 1);
        if (tail >= 1 && tb.d > 36 && tb.d <= 40) GSSS_SPEC_TAIL(4, 2, 2);
        if (tail >= 1 && tb.d > 48 && tb.d <= 52) GSSS_SPEC_TAIL(4, 3, 1);
        if (tail >= 1 && tb.d > 52 && tb.d <= 56) GSSS_SPEC_TAIL(4, 3, 2);
        if (tail >= 1 && tb.d > 96 && tb.d <= 104) GSSS_SPEC_TAIL(8, 3, 1);
        if (tail >= 1 && tb.d > 104 && tb.d <= 112) GSSS_SPEC_TAIL(8, 3, 2);
        if (tail >= 1 && tb.d > 192 && tb.d <= 208) GSSS_SPEC_TAIL(16, 3, 1);
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

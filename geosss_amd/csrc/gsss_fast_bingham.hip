// GSSS_MODE_FAST instantiations for Bingham targets.
#include "gsss_fast_bingham_lane.h"

namespace gsss {

#define GSSS_FAST_BINGHAM_DIMS(X) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)

int launch_fast_bingham(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st)
{
    // tb.k: bit 0 a diagonal A, bit 1 a linear term.  The paper's eigenbasis targets (diagonal, no linear term) run the compact
    // screen target (three wavefronts per SIMD at d = 9, 10: ScreenBinghamDiag, gsss_screen.h)
#define GSSS_CASE(D)                                               \
    if (tb.d == D) {                                               \
        const bool screen = rb.screen && !rb.spread && rb.rng_state == nullptr; \
        const bool compact = tb.k == 1;                            \
        if (probe) {                                               \
            if (rb.screen && compact) GSSS_PROBE(true, "screened_kernel<%d, ScreenBinghamDiag<%d>>", D, D); \
            if (rb.screen) GSSS_PROBE(true, "screened_kernel<%d, ScreenBingham<%d>>", D, D); \
            GSSS_PROBE(true, "fast_kernel<%d, FastBingham<%d>>", D, D); \
        }                                                          \
        if (rb.screen && !rb.spread && rb.rng_state != nullptr && !replay)   /* numpy's stream, packed: the screened kernel too */ \
            return compact ? do_screened_numpy<D, ScreenBinghamDiag<D>>(tb, rb, st) : do_screened_numpy<D, ScreenBingham<D>>(tb, rb, st); \
        if (!screen) return do_fast<D, FastBingham<D>>(tb, rb, replay, st); \
        if (compact) return replay ? do_screened_run<D, ScreenBinghamDiag<D>, true>(tb, rb, st) : do_screened_run<D, ScreenBinghamDiag<D>, false>(tb, rb, st); \
        return replay ? do_screened_run<D, ScreenBingham<D>, true>(tb, rb, st) : do_screened_run<D, ScreenBingham<D>, false>(tb, rb, st); \
    }
    GSSS_FAST_BINGHAM_DIMS(GSSS_CASE)
#undef GSSS_CASE
    // d = 11 .. 16, packed ensembles on the library stream: still one lane per chain (round 4)
    if (tb.d >= 11 && tb.d <= 16 && rb.screen != 0 && !rb.spread && rb.rng_state == nullptr && !replay) {
        switch (tb.d) {
#define GSSS_CASE_WIDE(D) \
    case D: return lane_bingham_wide<D>(tb, rb, probe, st);
            GSSS_BINGHAM_WIDE_DIMS(GSSS_CASE_WIDE)
#undef GSSS_CASE_WIDE
        default: break;
        }
    }
    // larger d: lanes cooperate on one chain.  A must fit the LDS beside the groups' scratch rows: d <= 126 ((d + 1) x 128 doubles
    // of rows + 16 groups x 258 of scratch in 160 KB) -- beyond, fast mode is not offered and mode "auto" runs the exact kernels
    // (which read a dense A of d > 128 from global memory)
    using Wide = CoopBingham<CoopVec<16, 8>>;
    const bool fits = tb.d <= 64 || (coop_param_doubles<Wide>(tb.d) + (size_t)Wide::kScratchPerGroup * (kBlock / 16)) * sizeof(double) <= 160 * 1024;
    if (tb.d > 10 && tb.d <= 128 && fits) {
        // Lanes per chain x slots per lane, measured at 10^5 chains (10^9 chain-steps/s, eigenbasis / dense A): d <= 32 four
        // lanes with eight slots 5.3 / 3.5 against 2.4 / 1.8 for sixteen lanes with four (the per-step serial work -- Philox
        // and Box-Muller rounds, reductions, the try loop -- is repeated in every lane of a group, and sixteen groups share a
        // wavefront); d <= 64 eight lanes with eight slots for a diagonal A (3.0 against 2.6), sixteen with four for a dense
        // one (its d x d products want the lanes: 1.36 against 1.26).
        const bool diag = (tb.k & 1) != 0;
        const int ll = tb.d <= 32 ? 4 : (tb.d <= 64 && diag ? 8 : 16), ss = tb.d <= 16 ? 4 : (tb.d <= 32 || tb.d > 64 || diag ? 8 : 4);
        if (probe) GSSS_PROBE(false, "coopfast_kernel<CoopVec<%d, %d>, CoopBingham>", ll, ss);
        if (tb.d <= 16) return do_coopfast<CoopVec<4, 4>, CoopBingham<CoopVec<4, 4>>>(tb, rb, replay, st);
        if (tb.d <= 32) return do_coopfast<CoopVec<4, 8>, CoopBingham<CoopVec<4, 8>>>(tb, rb, replay, st);
        if (tb.d <= 64 && diag) return do_coopfast<CoopVec<8, 8>, CoopBingham<CoopVec<8, 8>>>(tb, rb, replay, st);
        if (tb.d <= 64) return do_coopfast<CoopVec<16, 4>, CoopBingham<CoopVec<16, 4>>>(tb, rb, replay, st);
        return do_coopfast<CoopVec<16, 8>, CoopBingham<CoopVec<16, 8>>>(tb, rb, replay, st);
    }
    if (!probe) set_error("fast mode is not built for a Bingham target with d=%d", tb.d);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

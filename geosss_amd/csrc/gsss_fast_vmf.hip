// GSSS_MODE_FAST dispatch for von Mises-Fisher mixtures: lane-per-chain kernels for d <= 10 (any K <= 16,
// gsss_fast_vmf_lane.h, one translation unit per d), cooperative kernels beyond.
#include "gsss_fast_vmf_lane.h"

namespace gsss {

int launch_fast_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st)
{
    if (tb.k >= 1 && tb.k <= 16) {
        switch (tb.d) {
#define GSSS_CASE(D) \
    case D: return lane_vmf<D>(tb, rb, replay, probe, st);
            GSSS_VMF_LANE_DIMS(GSSS_CASE)
#undef GSSS_CASE
        default: break;
        }
    }
    // d = 11 .. 16, K <= 10, packed ensembles on the library stream: still one lane per chain (round 4; rounds 1-3 dropped to the
    // four-lane cooperative kernel at d = 11: 1.3 - 2.6e10 -> ~5e9 chain-steps/s)
    if (tb.k >= 1 && tb.k <= 10 && tb.d >= 11 && tb.d <= 16 && tb.scale <= kScreenMaxKappa && lane_wide_serves(rb, replay)) {
        switch (tb.d) {
#define GSSS_CASE(D) \
    case D: return lane_vmf_wide<D>(tb, rb, probe, st);
            GSSS_VMF_WIDE_DIMS(GSSS_CASE)
#undef GSSS_CASE
        default: break;
        }
    }
    // larger d: lanes cooperate on one chain; component buckets 3, 5, 10, 16 (surplus components padded, as above)
    if (tb.k >= 1 && tb.k <= 16 && tb.d > 10 && tb.d <= 256) {
        const int kc = tb.k <= 3 ? 3 : (tb.k <= 5 ? 5 : (tb.k <= 10 ? 10 : 16));
        // Lanes per chain x slots per lane (d <= lanes x slots).  The per-step serial work -- Philox and Box-Muller rounds, the
        // reductions, the try loop -- is repeated in every lane of a group, so few lanes with many slots win as long as the
        // registers hold them (16 slots: two wavefronts per SIMD).  Measured at 10^5 chains, K = 5 (10^9 chain-steps/s):
        // d = 32 <4,8> 2.8 against <16,4> 1.1; d = 50 <4,16> 2.2 / <8,8> 1.7 / <16,4> 1.1; d = 100 <8,16> 1.33 / <16,8> 1.05
        // / <64,4> 0.37; d = 200 (K = 3) <16,16> 0.92 against <64,4> 0.50.
        const int ll = tb.d <= 64 ? 4 : (tb.d <= 128 ? 8 : 16), ss = tb.d <= 16 ? 4 : (tb.d <= 32 ? 8 : 16);
        if (probe) GSSS_PROBE(false, "coopfast_kernel<CoopVec<%d, %d>, CoopVmf<%d>>", ll, ss, kc);
#define GSSS_COOP(K)                                                                                          \
    if (kc == K) {                                                                                            \
        if (tb.d <= 16) return do_coopfast<CoopVec<4, 4>, CoopVmf<CoopVec<4, 4>, K>>(tb, rb, replay, st);     \
        if (tb.d <= 32) return do_coopfast<CoopVec<4, 8>, CoopVmf<CoopVec<4, 8>, K>>(tb, rb, replay, st);     \
        if (tb.d <= 64) return do_coopfast<CoopVec<4, 16>, CoopVmf<CoopVec<4, 16>, K>>(tb, rb, replay, st);   \
        if (tb.d <= 128) return do_coopfast<CoopVec<8, 16>, CoopVmf<CoopVec<8, 16>, K>>(tb, rb, replay, st);  \
        return do_coopfast<CoopVec<16, 16>, CoopVmf<CoopVec<16, 16>, K>>(tb, rb, replay, st);                 \
    }
        GSSS_COOP(3) GSSS_COOP(5) GSSS_COOP(10) GSSS_COOP(16)
#undef GSSS_COOP
    }
    if (!probe) set_error("fast mode is not built for a vMF mixture with d=%d, K=%d", tb.d, tb.k);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

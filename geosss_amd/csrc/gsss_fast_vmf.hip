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
    // larger d: lanes cooperate on one chain; component buckets 3, 5, 10, 16 (surplus components padded, as above)
    if (tb.k >= 1 && tb.k <= 16 && tb.d > 10 && tb.d <= 256) {
        const int kc = tb.k <= 3 ? 3 : (tb.k <= 5 ? 5 : (tb.k <= 10 ? 10 : 16));
        if (probe)
            GSSS_PROBE(false, "coopfast_kernel<CoopVec<%d, 4>, CoopVmf<%d>>", tb.d <= 16 ? 4 : (tb.d <= 64 ? 16 : 64), kc);
#define GSSS_COOP(K)                                                                                          \
    if (kc == K) {                                                                                            \
        if (tb.d <= 16) return do_coopfast<CoopVec<4, 4>, CoopVmf<CoopVec<4, 4>, K>>(tb, rb, replay, st);     \
        if (tb.d <= 64) return do_coopfast<CoopVec<16, 4>, CoopVmf<CoopVec<16, 4>, K>>(tb, rb, replay, st);   \
        return do_coopfast<CoopVec<64, 4>, CoopVmf<CoopVec<64, 4>, K>>(tb, rb, replay, st);                   \
    }
        GSSS_COOP(3) GSSS_COOP(5) GSSS_COOP(10) GSSS_COOP(16)
#undef GSSS_COOP
    }
    if (!probe) set_error("fast mode is not built for a vMF mixture with d=%d, K=%d", tb.d, tb.k);
    return GSSS_E_UNSUPPORTED;
}

}  // namespace gsss

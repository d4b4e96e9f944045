// gsss_launch.h -- host-side dispatch from (target kind, vector layout, draw source) to a kernel.
#pragma once
#include "gsss_device.h"

namespace gsss {

// Vector layouts compiled in.  id -> type; ids are what gsss_run_args.variant selects.
using VL2 = LaneVec<2>;
using VL3 = LaneVec<3>;
using VL4 = LaneVec<4>;
using VL5 = LaneVec<5>;
using VL6 = LaneVec<6>;
using VL8 = LaneVec<8>;
using VL10 = LaneVec<10>;
using VC4x4 = CoopVec<4, 4>;
using VC4x8 = CoopVec<4, 8>;
using VC16x4 = CoopVec<16, 4>;
using VC16x8 = CoopVec<16, 8>;
using VC64x4 = CoopVec<64, 4>;
using VC64x8 = CoopVec<64, 8>;
using VC64x16 = CoopVec<64, 16>;
using VC64x32 = CoopVec<64, 32>;

#define GSSS_VEC_LIST(X) \
    X(1, VL2, "lane2")       \
    X(2, VL3, "lane3")       \
    X(3, VL4, "lane4")       \
    X(4, VL5, "lane5")       \
    X(5, VL6, "lane6")       \
    X(6, VL8, "lane8")       \
    X(7, VL10, "lane10")     \
    X(8, VC4x4, "coop4x4")   \
    X(9, VC4x8, "coop4x8")   \
    X(10, VC16x4, "coop16x4") \
    X(11, VC16x8, "coop16x8") \
    X(12, VC64x4, "coop64x4") \
    X(13, VC64x8, "coop64x8") \
    X(14, VC64x16, "coop64x16") \
    X(15, VC64x32, "coop64x32")

constexpr size_t kMaxLdsBytes = 160 * 1024;

struct VecInfo {
    int id, L, N, dpad;
    bool exact_dim;
    const char *name;
};
const VecInfo *vec_table(int *count);
// id of the layout used for dimension d (variant == 0) or validation of a forced one; <0 on error
int select_vec(int d, int variant);

// draw source of a launch
enum : int { kDrawsPhilox = 0, kDrawsReplay = 1, kDrawsNumpy = 2 };

template <template <class> class TT>
int launch_run(int vec_id, int draws, const TargetBlock &tb, const RunBlock &rb, hipStream_t st);
template <template <class> class TT>
int launch_logprob(int vec_id, const TargetBlock &tb, const double *x, int64_t n, double *out, bool grad, hipStream_t st);

void set_error(const char *fmt, ...);

#define GSSS_HIP_TRY(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            ::gsss::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return GSSS_E_HIP;                                                          \
        }                                                                               \
    } while (0)

template <class V, template <class> class TT, template <class> class DR>
int do_run(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    using T = TT<V>;
    const size_t lds = (T::lds_doubles(tb.k, tb.d) + scratch_doubles<V, T>() + DR<V>::kLdsDoubles) * sizeof(double);
    if (lds > kMaxLdsBytes) {
        set_error("target parameters need %zu B of LDS (> %zu)", lds, kMaxLdsBytes);
        return GSSS_E_UNSUPPORTED;
    }
    auto kern = run_kernel<V, TT, DR, false>;
    if (rb.stats != nullptr) kern = run_kernel<V, TT, DR, true>;  // running statistics: a second build, so that the plain kernels carry none of it
    if (lds > 48 * 1024)
        GSSS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t per_block = (V::L == 1 && rb.spread) ? kBlock / 64 : kBlock / V::L;
    const int64_t grid = (rb.n_chains + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb);
    GSSS_HIP_TRY(hipGetLastError());
    return GSSS_OK;
}

template <class V, template <class> class TT>
int do_logprob(const TargetBlock &tb, const double *x, int64_t n, double *out, bool grad, hipStream_t st)
{
    using T = TT<V>;
    const size_t lds = (T::lds_doubles(tb.k, tb.d) + scratch_doubles<V, T>()) * sizeof(double);
    if (lds > kMaxLdsBytes) {
        set_error("target parameters need %zu B of LDS (> %zu)", lds, kMaxLdsBytes);
        return GSSS_E_UNSUPPORTED;
    }
    auto kern = grad ? logprob_kernel<V, TT, true> : logprob_kernel<V, TT, false>;
    if (lds > 48 * 1024)
        GSSS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t per_block = kBlock / V::L;
    const int64_t grid = (n + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, x, n, out);
    GSSS_HIP_TRY(hipGetLastError());
    return GSSS_OK;
}

// Each target's translation unit expands this once.
#define GSSS_DEFINE_TARGET_LAUNCHERS(TT)                                                                      \
    template <>                                                                                               \
    int launch_run<TT>(int vec_id, int draws, const TargetBlock &tb, const RunBlock &rb, hipStream_t st)       \
    {                                                                                                         \
        switch (vec_id) {                                                                                     \
            GSSS_VEC_LIST(GSSS_RUN_CASE_##TT)                                                                  \
        }                                                                                                     \
        set_error("unknown vector layout %d", vec_id);                                                        \
        return GSSS_E_INVALID;                                                                                \
    }                                                                                                         \
    template <>                                                                                               \
    int launch_logprob<TT>(int vec_id, const TargetBlock &tb, const double *x, int64_t n, double *out,         \
                           bool grad, hipStream_t st)                                                         \
    {                                                                                                         \
        switch (vec_id) {                                                                                     \
            GSSS_VEC_LIST(GSSS_LOGPROB_CASE_##TT)                                                              \
        }                                                                                                     \
        set_error("unknown vector layout %d", vec_id);                                                        \
        return GSSS_E_INVALID;                                                                                \
    }

}  // namespace gsss

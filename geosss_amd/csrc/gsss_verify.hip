// gsss_verify.hip -- verification entry points of the single-precision screen (include/gsss.h: gsss_screen_constants,
// gsss_f32_error_sweep).  Nothing here is on the sampling path.
//
// The screened kernels decide four tries out of five on v_sin_f32 / v_cos_f32 / v_exp_f32 / v_log_f32 / v_sqrt_f32 with an error
// margin built from the constants of gsss_screen_consts.h; the decision they protect is geosss/mcmc.py:397 `if p(y) > threshold`.
// The sweep evaluates EVERY float of each instruction's argument range on the device against double precision and returns the
// largest errors, so that a test can hold the constants to the hardware it actually runs on.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gsss.h"
#include "gsss_screen_consts.h"

namespace gsss {
void set_error(const char *fmt, ...);  // gsss_capi.hip

namespace {

constexpr int kSweepBlock = 256;
constexpr int kSweepOut = 4;

__device__ void atomic_max_double(double *p, double v)
{
    unsigned long long *u = reinterpret_cast<unsigned long long *>(p);
    unsigned long long old = *u;
    while (__longlong_as_double((long long)old) < v) {  // (all values >= 0: ordered like their bit patterns, but compare as doubles)
        const unsigned long long seen = atomicCAS(u, old, (unsigned long long)__double_as_longlong(v));
        if (seen == old) break;
        old = seen;
    }
}

// half the spacing of the floats around x (the larger of the two sides at a power of two): a double that rounds to x lies within it
__device__ __forceinline__ double half_ulp(float x)
{
    int e;
    (void)frexpf(fabsf(x), &e);                   // |x| = m 2^e, m in [0.5, 1): spacing 2^(e - 24)
    if (e < -125) e = -125;                       // denormals: spacing 2^-149
    return ldexp(1.0, e - 25);
}

// which: 0 sin / cos of x revolutions, 1 2^x, 2 log2 x, 3 sqrt x.  Bit patterns lo .. lo + count - 1 of one sign.
__global__ void __launch_bounds__(kSweepBlock) f32_sweep_kernel(uint32_t lo_bits, uint64_t count, int which, double *acc)
{
    double e[kSweepOut] = {0.0, 0.0, 0.0, 0.0};
    const double two_pi = 6.283185307179586476925286766559, inv_ln2 = 1.4426950408889634;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        if (which == 0) {
            // sincos_rev32 (gsss_screen.h): t = fl32(theta / 2 pi), |t| <= 1.  [0], [1]: the hardware at the float; [2], [3]: against
            // ANY theta whose revolutions round to this float (|d/dt sin 2 pi t| <= 2 pi)
            // (|t| = 1 ends the range: only revolutions BELOW it round to it, from within 2^-25)
            const double a = two_pi * (double)x, slack = two_pi * (fabsf(x) == 1.0f ? 2.9802322387695312e-8 : half_ulp(x));
            const double es = fabs((double)__builtin_amdgcn_sinf(x) - sin(a)), ec = fabs((double)__builtin_amdgcn_cosf(x) - cos(a));
            e[0] = fmax(e[0], es);
            e[1] = fmax(e[1], ec);
            e[2] = fmax(e[2], es + slack);
            e[3] = fmax(e[3], ec + slack);
        } else if (which == 1) {
            // [0] relative error where the true result is a normal float, [1] absolute error where it is below the normals (the
            // hardware flushes), [2] arguments whose true result overflows and the hardware did not return +inf (must be 0)
            const double want = exp2((double)x), got = (double)__builtin_amdgcn_exp2f(x);
            if (want >= 3.4028235677973366e38) e[2] += (got == (double)INFINITY) ? 0.0 : 1.0;
            else if (want >= 1.1754943508222875e-38) e[0] = fmax(e[0], fabs(got - want) / want);
            else e[1] = fmax(e[1], fabs(got - want));
        } else if (which == 2) {
            // log2_32 (gsss_screen.h): v_log_f32 of the mantissa m in [0.5, 1) of a double, rounded to single.  [0]: the hardware at
            // the float; [1]: against ANY m that rounds to this float (d/dm log2 m = 1 / (m ln 2), m >= x - half an ulp)
            const double err = fabs((double)__builtin_amdgcn_logf(x) - log2((double)x));
            e[0] = fmax(e[0], err);
            if (x <= 1.0f) e[1] = fmax(e[1], err + half_ulp(x) * inv_ln2 / ((double)x - half_ulp(x)));
        } else {
            const double want = sqrt((double)x);
            e[0] = fmax(e[0], fabs((double)__builtin_amdgcn_sqrtf(x) - want) / want);
        }
    }
    for (int k = 0; k < kSweepOut; ++k) {
        if (which == 1 && k == 2) {
            if (e[k] > 0.0) atomicAdd(&acc[k], e[k]);
        } else if (e[k] > 0.0) {
            atomic_max_double(&acc[k], e[k]);
        }
    }
}

struct Range {
    float a, b;  // one sign, |a| <= |b|: consecutive bit patterns
};

}  // namespace
}  // namespace gsss

extern "C" {

int gsss_screen_constants(double *out, int32_t n)
{
    using namespace gsss;
    if (!out || n < 5) {
        set_error("gsss_screen_constants needs room for 5 doubles");
        return GSSS_E_INVALID;
    }
    out[0] = (double)kSinCosErr32;
    out[1] = (double)kExp2Err32;
    out[2] = (double)kLog2Err32;
    out[3] = (double)kSqrtRelErr32;
    out[4] = (double)kUnit32;
    return GSSS_OK;
}

int gsss_f32_error_sweep(int32_t which, double *out_host, uint64_t *n_swept_out, int device, void *stream)
{
    using namespace gsss;
    if (which < 0 || which > 3 || !out_host) {
        set_error("bad argument to gsss_f32_error_sweep");
        return GSSS_E_INVALID;
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) {
        set_error("gsss_f32_error_sweep: no such device");
        return GSSS_E_HIP;
    }
    // every float of the ranges the kernels can reach: |t| <= 1 revolution; every finite argument of 2^x; the mantissas [0.5, 1]
    // of frexp (and (1, 2], the form other callers might use); every positive normal float under the square root
    static const Range ranges[4][2] = {{{0.0f, 1.0f}, {-0.0f, -1.0f}},
                                       {{0.0f, 3.4028234664e38f}, {-0.0f, -3.4028234664e38f}},
                                       {{0.5f, 2.0f}, {1.0f, 1.0f}},
                                       {{1.17549435e-38f, 3.4028234664e38f}, {1.0f, 1.0f}}};
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *acc = nullptr;
    int rc = GSSS_OK;
    uint64_t swept = 0;
    double zero[kSweepOut] = {0.0, 0.0, 0.0, 0.0};
    if (hipMalloc(&acc, sizeof(zero)) != hipSuccess || hipMemcpyAsync(acc, zero, sizeof(zero), hipMemcpyHostToDevice, st) != hipSuccess) {
        rc = GSSS_E_HIP;
    } else {
        for (int r = 0; r < 2 && rc == GSSS_OK; ++r) {
            if (r == 1 && which >= 2) break;  // one range
            uint32_t ua, ub;
            memcpy(&ua, &ranges[which][r].a, 4);
            memcpy(&ub, &ranges[which][r].b, 4);
            const uint64_t count = (uint64_t)(ub - ua) + 1;
            hipLaunchKernelGGL(f32_sweep_kernel, dim3(16384), dim3(kSweepBlock), 0, st, ua, count, (int)which, acc);
            if (hipGetLastError() != hipSuccess) rc = GSSS_E_HIP;
            swept += count;
        }
        if (rc == GSSS_OK && (hipMemcpyAsync(out_host, acc, sizeof(zero), hipMemcpyDeviceToHost, st) != hipSuccess ||
                              hipStreamSynchronize(st) != hipSuccess))
            rc = GSSS_E_HIP;
    }
    if (acc) (void)hipFree(acc);
    (void)hipSetDevice(prev);
    if (rc != GSSS_OK) {
        (void)hipGetLastError();
        set_error("gsss_f32_error_sweep: a HIP call failed");
        return rc;
    }
    if (n_swept_out) *n_swept_out = swept;
    return GSSS_OK;
}

}  // extern "C"

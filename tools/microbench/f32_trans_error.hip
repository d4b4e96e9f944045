// Exhaustive error sweep of the hardware single-precision transcendentals the screened fast kernels use to
// bound a try's level (DESIGN.md "f32 screening"): v_sin_f32 / v_cos_f32 (argument in revolutions) over
// EVERY float in [-1, 1], v_exp_f32 (2^x) over every float in [-160, 130], v_log_f32 over every float in [0.5, 2).
// Prints the largest errors; the margins in gsss_screen.h are these values with a safety factor.
//   hipcc -O2 --offload-arch=gfx950 f32_trans_error.hip -o f32_trans_error && ./f32_trans_error
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

struct Acc { double sin_abs, cos_abs, exp_rel, log_abs; };

__device__ void amax(double *p, double v)
{
    unsigned long long *u = reinterpret_cast<unsigned long long *>(p);
    unsigned long long old = *u;
    while (__longlong_as_double((long long)old) < v) {
        const unsigned long long seen = atomicCAS(u, old, (unsigned long long)__double_as_longlong(v));
        if (seen == old) break;
        old = seen;
    }
}

__global__ void sweep(uint32_t lo_bits, uint32_t count, int which, Acc *acc)
{
    double e0 = 0.0, e1 = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        if (which == 0) {        // sin / cos of x revolutions, |x| <= 1
            const double a = 6.283185307179586476925286766559 * (double)x;
            e0 = fmax(e0, fabs((double)__builtin_amdgcn_sinf(x) - sin(a)));
            e1 = fmax(e1, fabs((double)__builtin_amdgcn_cosf(x) - cos(a)));
        } else if (which == 1) { // 2^x
            const double want = exp2((double)x);
            const double got = (double)__builtin_amdgcn_exp2f(x);
            if (want >= 1.17549435e-38 && want < 3.0e38) e0 = fmax(e0, fabs(got - want) / want);
            else if (want < 1.17549435e-38) e1 = fmax(e1, fabs(got - want));  // denormal range: absolute error
        } else {                 // log2 x
            e0 = fmax(e0, fabs((double)__builtin_amdgcn_logf(x) - log2((double)x)));
        }
    }
    if (which == 0) { amax(&acc->sin_abs, e0); amax(&acc->cos_abs, e1); }
    if (which == 1) { amax(&acc->exp_rel, e0); amax(&acc->log_abs, 0.0); amax(&acc->sin_abs, 0.0); if (e1 > 0) amax(&acc->cos_abs, e1); }
    if (which == 2) amax(&acc->log_abs, e0);
}

static void run(const char *name, float a, float b, int which, Acc *d)
{
    // floats of one sign between |a| and |b| are consecutive bit patterns
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    if (ua > ub) { uint32_t t = ua; ua = ub; ub = t; }
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, ua, ub - ua + 1, which, d);
    hipDeviceSynchronize();
    printf("swept %s: %u values\n", name, ub - ua + 1);
}

int main()
{
    Acc *d, h = {0, 0, 0, 0};
    hipMalloc(&d, sizeof(Acc));
    hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    run("sincos [0, 1]", 0.0f, 1.0f, 0, d);
    run("sincos [-1, -0]", -0.0f, -1.0f, 0, d);
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_sin_f32 max abs err %.3e   v_cos_f32 max abs err %.3e  (argument in revolutions, |x| <= 1)\n", h.sin_abs, h.cos_abs);
    Acc z = {0, 0, 0, 0};
    hipMemcpy(d, &z, sizeof(z), hipMemcpyHostToDevice);
    run("exp2 [0, 130]", 0.0f, 130.0f, 1, d);
    run("exp2 [-160, -0]", -0.0f, -160.0f, 1, d);
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_exp_f32 max rel err %.3e (normal results), max abs err %.3e (denormal results)\n", h.exp_rel, h.cos_abs);
    hipMemcpy(d, &z, sizeof(z), hipMemcpyHostToDevice);
    run("log2 [0.5, 2]", 0.5f, 2.0f, 2, d);
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_log_f32 max abs err %.3e on [0.5, 2]\n", h.log_abs);
    return 0;
}

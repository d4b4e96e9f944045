// Micro-benchmark: sustained VALU issue rates on gfx950 for the instruction classes the sampler
// uses (f64 fma/add/mul, 32-bit logic, v_mad_u64_u32, v_cndmask, ldexp, rcp/sqrt f64), at a
// chosen number of waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x * 3u + 1u, u1 = u0 + 7, u2 = u0 + 11, u3 = u0 + 13, u4 = u0 + 17, u5 = u0 + 19, u6 = u0 + 23, u7 = u0 + 29;
    unsigned long long w0 = u0, w1 = u1, w2 = u2, w3 = u3;
    const double m = 0.9999999, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
            if (OP == 1) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
            if (OP == 2) { u0 ^= u1; u1 ^= u2; u2 ^= u3; u3 ^= u4; u4 ^= u5; u5 ^= u6; u6 ^= u7; u7 ^= u0 + 1; }
            if (OP == 3) { w0 = (unsigned long long)(unsigned)w0 * 0xD2511F53u + w1; w1 = (unsigned long long)(unsigned)w1 * 0xCD9E8D57u + w2; w2 = (unsigned long long)(unsigned)w2 * 0xD2511F53u + w3; w3 = (unsigned long long)(unsigned)w3 * 0xCD9E8D57u + w0; }
            if (OP == 4) { a0 = a0 > 0.5 ? a1 : a2; a1 = a1 > 0.5 ? a2 : a3; a2 = a2 > 0.5 ? a3 : a4; a3 = a3 > 0.5 ? a4 : a0; }
            if (OP == 5) { a0 = ldexp(a0, 1); a1 = ldexp(a1, -1); a2 = ldexp(a2, 1); a3 = ldexp(a3, -1); a4 = ldexp(a4, 1); a5 = ldexp(a5, -1); a6 = ldexp(a6, 1); a7 = ldexp(a7, -1); }
            if (OP == 6) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); }
            if (OP == 7) { a0 = __builtin_amdgcn_sqrt(a0); a1 = __builtin_amdgcn_sqrt(a1); a2 = __builtin_amdgcn_sqrt(a2); a3 = __builtin_amdgcn_sqrt(a3); }
            if (OP == 8) { a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m; }
            if (OP == 9) { u0 += u1; u1 += u2; u2 += u3; u3 += u4; u4 += u5; u5 += u6; u6 += u7; u7 += u0; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + (double)(w0 + w1 + w2 + w3);
}

template <int OP>
void run(const char *name, int per_iter, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
    int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_wave = (double)iters * 8 * per_iter;
    double cyc = ms * 1e-3 * 2.4e9;  // nominal clock
    printf("%-14s waves/SIMD=%d  %.2f cycles per wave-instr per SIMD (@2.4GHz nominal), %.3f ms\n", name, waves_per_simd,
           cyc / (instr_per_wave * waves_per_simd), ms);
    CHECK(hipFree(out));
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", 8, w);
        run<1>("v_add_f64", 8, w);
        run<8>("v_mul_f64", 8, w);
        run<2>("v_xor_b32", 8, w);
        run<9>("v_add_u32", 8, w);
        run<3>("v_mad_u64_u32", 4, w);
        run<4>("cmp+cndmask64", 4, w);
        run<5>("v_ldexp_f64", 8, w);
        run<6>("v_rcp_f64", 4, w);
        run<7>("v_sqrt_f64", 4, w);
    }
    return 0;
}

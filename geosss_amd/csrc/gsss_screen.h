// gsss_screen.h -- the lane-per-chain throughput kernel with single-precision SCREENING of the tries.
//
// A try of the shrinkage loop only needs a yes/no: is the level of y(theta) above the threshold
// (mcmc.py:397)?  Four out of five tries are rejections, most of them by a wide margin.  So a try is
// first evaluated in single precision on the hardware transcendentals (v_sin_f32 / v_cos_f32 / v_exp_f32:
// 8 cycles each, against ~150 cycles for a double-precision sincos and ~75 per exp) together with a
// RIGOROUS bound on the error of that evaluation:
//
//     level32 < threshold (1 - margin)   ->  rejected, exactly as the double-precision test would
//     level32 > threshold (1 + margin)   ->  accepted, ditto
//     otherwise                          ->  undecided: the double-precision level decides
//
// The margin is derived per step from the magnitudes of the step's coefficients and from the measured
// worst-case errors of the hardware functions (tools/microbench/f32_trans_error.hip sweeps EVERY float of the
// argument ranges used here: |sin|,|cos| error <= 1.254e-7, 2^x relative error <= 8.5e-8, log2 <= 6e-8).
// Every decision is therefore the decision the all-double kernel (fast_kernel, gsss_fast.h) takes, and
// everything that enters the chain -- theta, sin/cos of the accepted theta, the new state, the level
// carried to the next step -- is computed in double precision exactly as there: the two kernels produce
// the same chains bit for bit (tests/test_hip_parity.py::test_screened_equals_double).
//
// The double-precision work of an accepted (or undecided) try is DEFERRED to the next batched set-up
// phase, where most lanes take part, instead of being executed for the few lanes that accepted in this
// very iteration.  Scheduling (two chains per lane, one parked in LDS, set-up when enough lanes wait) is
// that of fast_kernel.
#pragma once
#include <type_traits>

#include "gsss_fast.h"
#include "gsss_screen_consts.h"

#ifndef GSSS_VMF_ONE_BUILD
#define GSSS_VMF_ONE_BUILD 1  // (0: without the one-chain-per-lane build of the K = 7 .. 10 mixtures, A/B)
#endif
#ifndef GSSS_VMF_ONE_ALL
#define GSSS_VMF_ONE_ALL 0  // (measurement: build it for every bucket; GSSS_ONE_PER_LANE=2 then runs it)
#endif
#ifndef GSSS_SCREEN_REGEN_THR
#define GSSS_SCREEN_REGEN_THR 1  // (A/B: 0 parks the threshold uniform of the S^2 mixtures as round 2 did)
#endif

namespace gsss {

// A try that stopped: accepted for sure / double precision decides.  Its theta rests in the end of the bracket it would
// become were it rejected (mcmc.py:400: lo for theta < 0, else hi) -- the other end is dead if it is accepted and unchanged
// if it is not -- and the status says which end: + kFinalInHi.  (No word of its own in a chain's parked state.)
enum : int32_t { kFinalAccept = 3, kFinalDecide = 4, kFinalInHi = 2 };
__device__ __forceinline__ bool is_final(int32_t st) { return st >= kFinalAccept; }
__device__ __forceinline__ bool is_decide(int32_t st) { return st == kFinalDecide || st == kFinalDecide + kFinalInHi; }

__device__ __forceinline__ void sincos_rev32(double theta, float &s, float &c)
{
    const float t = (float)(theta * 0.15915494309189535);  // revolutions
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

// log2 of a positive finite double to single precision
__device__ __forceinline__ float log2_32(double v)
{
    int e;
    const double m = frexp(v, &e);
    return (float)e + __builtin_amdgcn_logf((float)m);
}

// ------------------------------------------------------------------------------------------
// screening side of the restricted targets.  A target provides, next to gsss_fast.h's interface,
//   kCoef32Floats              size of the single-precision pack q[]
//   coeffs(cf, x, u)           the double-precision coefficients of the circle (bit-identical to make()'s)
//   make32(cf, U, q)           single-precision pack and error margin for the step, from the coefficients and the
//                              threshold uniform U; false if the level of x is not finite
//   screen(q, c, s)            -1 reject / +1 accept (both certain) / 0 undecided
//   threshold(cf, x, u, U)     the double-precision threshold exactly as fast_kernel forms it (undecided tries only)
//   level_exact(cf, c, s)      the double-precision level fast_kernel compares with that threshold
// and the two fused forms the kernel calls (a target may build them without ever holding a whole Coef in registers):
//   setup32(x, u, U, q)        = coeffs + make32
//   kParkSkip, refill(x, u, q) the first kParkSkip floats of q are NOT parked with a chain's second state: refill() forms
//                              them again from x and u when the chain is taken up (same operations, same bits).  Wide
//                              mixtures: 2 K floats = 40 % of the parked state, one more workgroup per CU without them
//   kMinWaves                  wavefronts per SIMD the kernel is built for (__launch_bounds__)
//   kTradeMin                  a lane swaps its stopped chain for its parked one when at least this many lanes of the
//                              wavefront want to (or none can try): the swap runs for the whole wavefront whoever takes part
//   decide(x, u, U, c, s)      = level_exact(c, s) > threshold, the all-double decision of an undecided try
// ------------------------------------------------------------------------------------------
template <int D, int KC>
struct ScreenVmf : FastVmf<D, KC> {
    using Base = FastVmf<D, KC>;
    using Coef = typename Base::Coef;
    static constexpr int kCoef32Floats = 3 * KC + 1;
    // K >= 6: mu_k.x and mu_k.u in single precision (2 K floats) are formed again at take-up instead of being parked:
    // 18 instead of 28 words of parked state at K = 10, three workgroups per CU instead of two (and the register budget
    // of three wavefronts per SIMD asked of the compiler: 170 -> 168)
    static constexpr int kParkSkip = KC >= 6 ? 2 * KC : 0;  // (K = 3: 36.5 against 34.8 ms with it -- four workgroups per CU fit anyway)
    static constexpr int kMinWaves = D > 10 ? ((D <= 12 && KC <= 3) ? 3 : 2) : (KC >= 6 ? 3 : ((GSSS_SCREEN_REGEN_THR && D == 3 && KC <= 3) ? 5 : 1));
    // K >= 6 forms the 2 K coefficients again at take-up (kParkSkip), which makes a swap as dear as a pair of tries: swapping only
    // when 24 lanes want to is worth 9 % (K = 10, kappa = 500: 56.6 -> 51.7 ms per 10^9 chain-steps; 12: 52.5, 32: 59.2, 44: 71.7);
    // with the cheap swaps of K <= 5 and of the Bingham target waiting costs more than it saves (27.7 -> 28.3 / 28.6 ms at 12 / 24).
    static constexpr int kTradeMin = KC >= 6 ? 24 : 1;
    static constexpr bool kCompact = false;
    // The one-chain-per-lane BUILD (screened_kernel<.., STAGE>: no code for a parked chain; do_screened_run) at any ensemble size
    // where it was measured ahead at 10^6 chains (tools/bench_vmf_pure_one.py, profiles/r04_vmf_pure_one.log): bucket 10 at every
    // d (+6 .. 11 %; cfg5, S^2 K = 10 kappa = 500: 48.9 -> 46.8 ms), bucket 6 up to d = 8 (+4 .. 10 %), bucket 16 up to d = 5 (+7 %),
    // bucket 4 on S^2 (+7 %), bucket 3 at d = 4 (+4.5 %).  The README kernel (S^2, K = 3) stays with two chains per lane (24.1
    // against 24.7 ms); everything else is within 1 % either way and keeps the old rule.
    static constexpr bool kOneAhead = KC == 10 || (KC == 6 && D <= 8) || (KC == 16 && D <= 5) || (KC == 4 && D == 3) || (KC == 3 && D == 4);
    static constexpr bool kStageRows = GSSS_VMF_ONE_BUILD && (kOneAhead || GSSS_VMF_ONE_ALL);
    static constexpr bool kHoldRows = false;  // ... without the rows held back in LDS (registers)
    static constexpr bool kPreferOne = GSSS_VMF_ONE_BUILD && kOneAhead;
    static constexpr int kNumpyWaves = (KC >= 10 || (D >= 9 && KC >= 6)) ? 2 : ((D <= 4 && KC <= 3) ? 4 : 3);  // wavefronts per SIMD of the numpy-stream build: without scratch, but for the smallest
                                                   // shapes, where a fourth wavefront is worth 12-20 spilled bytes (README target 40.9 -> 38.7 ms)
    // S^2, K <= 3 (the README target, BASELINE cfg2): the threshold uniform is not parked -- an undecided try draws it again
    // from the counter-based stream -- which makes the parked state 15 words: five workgroups per CU instead of four (the
    // kernel needs 95 registers: five wavefronts per SIMD fit)
    static constexpr bool kRegenThr = GSSS_SCREEN_REGEN_THR && D == 3 && KC <= 3;
    __device__ __forceinline__ void retail(float (&)[kCoef32Floats]) const {}
    __device__ __forceinline__ void refill(const double (&x)[D], const double (&u)[D], float (&q)[kCoef32Floats]) const
    {
        constexpr double L = 1.4426950408889634074;
#pragma unroll
        for (int k = 0; k < KC; ++k) {  // the operations of coeffs() and make32(), hence their bits
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double mkj = this->mu[k * D + j];
                ax = fma(mkj, x[j], ax);
                au = fma(mkj, u[j], au);
            }
            q[k] = (float)(ax * L);
            q[KC + k] = (float)(au * L);
        }
    }

    __device__ __forceinline__ void coeffs(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double mkj = this->mu[k * D + j];
                ax = fma(mkj, x[j], ax);
                au = fma(mkj, u[j], au);
            }
            cf.ax[k] = ax;
            cf.au[k] = au;
        }
        double m = -INFINITY;  // the offset make() chose for this step
#pragma unroll
        for (int k = 0; k < KC; ++k) m = fmax(m, cf.ax[k] + this->logc[k]);
        cf.m = m;
    }
    // Exponents to base 2, relative to log2 of the threshold thr = level(x) U (mcmc.py:389):
    //   q = [ax L | au L | (logc - m) L - log2 thr | margin].
    // log2 thr itself is only formed in single precision here (its error is part of the margin); the double-
    // precision threshold is formed when a try is left undecided (threshold()).  Returns false when the level of
    // x is not a positive finite number.
    __device__ __forceinline__ bool make32(const Coef &cf, double u_thr, float (&q)[kCoef32Floats]) const
    {
        constexpr double L = 1.4426950408889634074;
        float b = 0.0f, s0 = 0.0f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            q[k] = (float)(cf.ax[k] * L);
            q[KC + k] = (float)(cf.au[k] * L);
            q[2 * KC + k] = (float)((this->logc[k] - cf.m) * L);
            s0 += __builtin_amdgcn_exp2f(q[k] + q[2 * KC + k]);  // theta = 0: cos = 1, sin = 0 exactly
        }
        // log2(level(x) U): s0 lies in [1, K] (the largest exponent is 0 up to rounding)
        int e;
        const float m0 = frexpf(s0, &e);
        const float t2 = ((float)e + __builtin_amdgcn_logf(m0)) + log2_32(u_thr);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            q[2 * KC + k] -= t2;
            // (padding components beyond K have exponent -1e5: they add +0 to every sum and carry no error)
            if (k < this->K) b = fmaxf(b, fabsf(q[k]) + fabsf(q[KC + k]) + fabsf(q[2 * KC + k]) + fabsf(t2));
        }
        // error of one exponent of a try: (|ax| + |au|) (eps_sincos + 2^-24) + |lc| 2^-24 + two fma roundings, plus the
        // error of log2 thr: three roundings per exponent of s0, v_exp_f32, the additions, v_log_f32 twice, the subtraction
        const float e_a = b * (kSinCosErr32 + 7.0f * kUnit32) + (1.4427f * (kExp2Err32 + (float)KC * kUnit32) + 6.0e-8f + kLog2Err32);
        // relative error of the sum of 2^exponent: ln 2 * e_a (1 + e_a) + v_exp_f32 + the additions; 25 % on top
        float margin = 1.25f * (0.69315f * e_a * (1.0f + e_a) + kExp2Err32 + (float)KC * kUnit32) + 1.0e-7f;
        if (!(u_thr > 1e-290) || !(margin < 0.25f)) margin = INFINITY;  // (also U = 0, NaN): double precision decides every try
        q[3 * KC] = margin;
        return s0 > 0.5f && s0 < 3.0e38f;
    }
    // the double-precision threshold of the step, exactly as fast_kernel forms it
    __device__ __forceinline__ double threshold(Coef &cf, const double (&x)[D], const double (&u)[D], double u_thr) const
    {
        return this->make(cf, x, u, 0.0, true) * u_thr;
    }
    __device__ __forceinline__ int screen(const float (&q)[kCoef32Floats], float c, float s) const
    {
        // K >= 6: the exponents two at a time (v_pk_fma_f32: the same fused operations, the same bits, half the
        // instructions; K = 10: 58.4 -> 56.4 ms.  At K = 3 the moves that pair the operands cost more: 28.0 -> 29.6 ms)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 c2 = {c, c}, s2 = {s, s};
        float sum = 0.0f;
        constexpr int kPacked = KC >= 6 ? KC - KC % 2 : 0;
#pragma unroll
        for (int k = 0; k + 1 < kPacked + 1 && k < kPacked; k += 2) {
            const f2 ax = {q[k], q[k + 1]}, au = {q[KC + k], q[KC + k + 1]}, lc = {q[2 * KC + k], q[2 * KC + k + 1]};
            const f2 e = __builtin_elementwise_fma(c2, ax, __builtin_elementwise_fma(s2, au, lc));
            sum += __builtin_amdgcn_exp2f(e.x);
            sum += __builtin_amdgcn_exp2f(e.y);
        }
#pragma unroll
        for (int k = kPacked; k < KC; ++k) sum += __builtin_amdgcn_exp2f(fmaf(c, q[k], fmaf(s, q[KC + k], q[2 * KC + k])));
        const float margin = q[3 * KC];
        return sum < 1.0f - margin ? -1 : (sum > 1.0f + margin ? 1 : 0);
    }
    __device__ __forceinline__ double level_exact(const Coef &cf, double c, double s) const
    {
        double a[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) a[k] = fma(c, cf.ax[k], fma(s, cf.au[k], this->logc[k])) - cf.m;
        return this->level_full(a);
    }
    __device__ __forceinline__ bool setup32(const double (&x)[D], const double (&u)[D], double u_thr, float (&q)[kCoef32Floats]) const
    {
        Coef cf;
        coeffs(cf, x, u);
        return make32(cf, u_thr, q);
    }
    __device__ __forceinline__ bool decide(const double (&x)[D], const double (&u)[D], double u_thr, double c, double s) const
    {
        Coef cf;
        const double thr = threshold(cf, x, u, u_thr);
        return level_exact(cf, c, s) > thr;
    }
};

// Bingham / BinghamFisher: log-density q(theta) = c^2 qxx + c s qxu + s^2 quu + c bx + s bu (distributions.py:86, :113-114),
// accepted iff q(theta) > thr = q(0) + log U.  With c^2 + s^2 = 1 the threshold is folded into the quadratic terms, so the
// screen evaluates g = c^2 (qxx - thr) + c s qxu + s^2 (quu - thr) + c bx + s bu against 0 +- margin.
template <int D>
struct ScreenBingham : FastBingham<D> {
    using Base = FastBingham<D>;
    using Coef = typename Base::Coef;
    static constexpr int kCoef32Floats = 6;
    static constexpr int kParkSkip = 0, kMinWaves = D > 10 ? 2 : 1, kTradeMin = 1;
    static constexpr bool kCompact = false, kRegenThr = false, kStageRows = true;
    static constexpr bool kPreferOne = D != 6;  // the one-chain-per-lane build at any ensemble size (do_screened_run)
    static constexpr bool kHoldRows = true;
    static constexpr int kNumpyWaves = D >= 7 ? 2 : 3;
    __device__ __forceinline__ void retail(float (&)[kCoef32Floats]) const {}
    __device__ __forceinline__ void refill(const double (&)[D], const double (&)[D], float (&)[kCoef32Floats]) const {}
    __device__ __forceinline__ void coeffs(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
        (void)this->make(cf, x, u, 0.0, true);
    }
    __device__ __forceinline__ bool make32(const Coef &cf, double u_thr, float (&q)[kCoef32Floats]) const
    {
        // thr = level(x) + log U (mcmc.py:389), log U in single precision: the screen needs the threshold to ~1e-6 of the
        // coefficients' scale, and the error of the logarithm (log2_32: <= kLog2Err32 + 2^-24 |log2 U|, times ln 2, one
        // more rounding each for the product and the two differences) goes into the margin; the double-precision
        // threshold is formed when a try stays undecided (threshold())
        const double lvl0 = cf.qxx + cf.bx;
        const float l2 = log2_32(u_thr);
        const float lu = l2 * 0.69314718f;
        q[0] = (float)(cf.qxx - lvl0) - lu;
        q[1] = (float)cf.qxu;
        q[2] = (float)(cf.quu - lvl0) - lu;
        q[3] = (float)cf.bx;
        q[4] = (float)cf.bu;
        const float sum = fabsf(q[0]) + fabsf(q[1]) + fabsf(q[2]) + fabsf(q[3]) + fabsf(q[4]);
        const float e_lu = (kLog2Err32 + 2.0f * kUnit32 * fabsf(l2)) * 0.6932f + kUnit32 * (fabsf(lu) + fabsf(q[0]) + fabsf(q[2]));
        // every term carries at most two trigonometric factors (2 eps), its coefficient's rounding and the fma roundings;
        // the threshold's error enters through c^2 q0 + s^2 q2 with weight c^2 + s^2 = 1
        float margin = 1.25f * (sum * (2.0f * kSinCosErr32 + 6.0f * kUnit32) + e_lu) + 1.0e-30f;
        if (!(u_thr > 1e-290) || !(margin < 1.0e30f)) margin = INFINITY;
        q[5] = margin;
        return lvl0 > -INFINITY && lvl0 < INFINITY;
    }
    __device__ __forceinline__ int screen(const float (&q)[kCoef32Floats], float c, float s) const
    {
        const float g = fmaf(c, fmaf(c, q[0], fmaf(s, q[1], q[3])), s * fmaf(s, q[2], q[4]));
        return g < -q[5] ? -1 : (g > q[5] ? 1 : 0);
    }
    __device__ __forceinline__ double threshold(Coef &cf, const double (&x)[D], const double (&u)[D], double u_thr) const
    {
        return this->make(cf, x, u, 0.0, true) + fm::log_fast(u_thr);
    }
    __device__ __forceinline__ double level_exact(const Coef &cf, double c, double s) const { return this->level(cf, c, s); }
    __device__ __forceinline__ bool setup32(const double (&x)[D], const double (&u)[D], double u_thr, float (&q)[kCoef32Floats]) const
    {
        Coef cf;
        coeffs(cf, x, u);
        return make32(cf, u_thr, q);
    }
    __device__ __forceinline__ bool decide(const double (&x)[D], const double (&u)[D], double u_thr, double c, double s) const
    {
        Coef cf;
        const double thr = threshold(cf, x, u, u_thr);
        return level_exact(cf, c, s) > thr;
    }
};

// Bingham with a DIAGONAL A and no linear term -- the eigenbasis targets of the paper (random_bingham(eigensystem=True),
// scripts/bingham.py:131; BASELINE cfg3) -- with a COMPACT parked state (round 3).  The lane kernel keeps a second chain per lane
// parked in LDS; at d = 10 that state was 28 words (x, u, bracket, threshold uniform, six single-precision floats, counters):
// 57 KB per workgroup, two workgroups per CU, two wavefronts per SIMD -- the vector pipes issued 77 % of the time.  Here:
//   * only the d diagonal entries are staged (80 B instead of 880 B of parameters);
//   * three coefficients (no b.x, b.u), so the pack is q0, q1, q2 + margin, and the margin is not parked but formed again from
//     q0 .. q2 when a chain is taken up (retail(): the same expression as at set-up, a dozen instructions);
//   * the threshold uniform is not parked: an undecided try (rare) draws it again from the counter-based stream;
//   * the row of the next retained sample is not parked: it follows from the step count (one exact integer division at the
//     end of a step, only when samples or statistics are kept).
// 25 words = 51 200 B per workgroup: three workgroups per CU (3 x 53 888 B <= 160 KB), three wavefronts per SIMD.
// Arithmetic: FastBingham's diagonal branch operation for operation (b = 0 adds exact zeros there), so the chains are those of
// fast_kernel<D, FastBingham<D>> bit for bit (test_screened_equals_double).
template <int D>
struct ScreenBinghamDiag {
    static constexpr bool kLinear = false;
    static constexpr int kCoef32Floats = 4;  // q0 = -log U, q1 = qxu, q2 = (quu - qxx) - log U | margin
    static constexpr int kParkSkip = 0, kMinWaves = D >= 14 ? 2 : (D >= 9 ? 3 : 1), kTradeMin = 1;  // (d >= 14 spills at three)
    static constexpr bool kCompact = true, kRegenThr = true, kStageRows = true;
    static constexpr bool kPreferOne = D >= 5;  // the one-chain-per-lane build at any ensemble size (do_screened_run)
    static constexpr bool kHoldRows = true;
    static constexpr int kNumpyWaves = 3;
    const double *a;  // LDS [D]: the diagonal of A
    struct Coef {
        double qxx, qxu, quu;
    };
    __host__ __device__ static size_t lds_doubles() { return (size_t)D; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        for (int i = threadIdx.x; i < D; i += kBlock) lds[i] = tb.blob[(size_t)i * D + i];
        a = lds;
    }
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
        double qxx = 0.0, qxu = 0.0, quu = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {  // FastBingham::make, diagonal branch
            const double ajj = a[j];
            const double xa = x[j] * ajj, ua = u[j] * ajj;
            qxx = fma(xa, x[j], qxx);
            qxu = fma(xa, u[j], fma(ua, x[j], qxu));
            quu = fma(ua, u[j], quu);
        }
        cf.qxx = qxx;
        cf.qxu = qxu;
        cf.quu = quu;
        return qxx + 0.0;  // (+ b.x = 0, as there)
    }
    __device__ __forceinline__ double level_exact(const Coef &cf, double c, double s) const
    {
        return fma(c * c, cf.qxx, fma(c * s, cf.qxu, (s * s) * cf.quu)) + fma(c, 0.0, s * 0.0);  // FastBingham::level with bx = bu = 0
    }
    // margin from the parked coefficients alone: |log2 U| <= 1.4428 |q0| bounds the logarithm's error terms of ScreenBingham::make32
    __device__ __forceinline__ void retail(float (&q)[kCoef32Floats]) const
    {
        const float sum = fabsf(q[0]) + fabsf(q[1]) + fabsf(q[2]);
        const float l2 = 1.4428f * fabsf(q[0]);
        const float e_lu = (kLog2Err32 + 2.0f * kUnit32 * l2) * 0.6932f + kUnit32 * (2.0f * fabsf(q[0]) + fabsf(q[2]));
        float margin = 1.25f * (sum * (2.0f * kSinCosErr32 + 6.0f * kUnit32) + e_lu) + 1.0e-30f;
        if (!(margin < 1.0e30f)) margin = INFINITY;  // (U = 0 or NaN: q0 is not finite -- double precision decides every try)
        q[3] = margin;
    }
    __device__ __forceinline__ void refill(const double (&)[D], const double (&)[D], float (&)[kCoef32Floats]) const {}
    __device__ __forceinline__ bool setup32(const double (&x)[D], const double (&u)[D], double u_thr, float (&q)[kCoef32Floats]) const
    {
        Coef cf;
        const double lvl0 = make(cf, x, u);
        const float lu = log2_32(u_thr) * 0.69314718f;  // log U in single precision: its error is in the margin (ScreenBingham::make32)
        q[0] = 0.0f - lu;                                // (float)(qxx - lvl0) = 0 exactly
        q[1] = (float)cf.qxu;
        q[2] = (float)(cf.quu - lvl0) - lu;
        if (!(u_thr > 1e-290)) q[0] = INFINITY;
        retail(q);
        return lvl0 > -INFINITY && lvl0 < INFINITY;
    }
    __device__ __forceinline__ int screen(const float (&q)[kCoef32Floats], float c, float s) const
    {
        const float g = fmaf(c, fmaf(c, q[0], s * q[1]), s * (s * q[2]));
        return g < -q[3] ? -1 : (g > q[3] ? 1 : 0);
    }
    __device__ __forceinline__ bool decide(const double (&x)[D], const double (&u)[D], double u_thr, double c, double s) const
    {
        Coef cf;
        const double thr = make(cf, x, u) + fm::log_fast(u_thr);  // mcmc.py:389, as fast_kernel forms it
        return level_exact(cf, c, s) > thr;
    }
};

// The single-precision side of the curve-vMF screen, independent of how the chain's components are laid out
// (lane kernels: ScreenCurve below; lane groups: gsss_curvespec.h).  q = [a_i.x | a_i.u | thr / kappa | margin].
template <int NK, bool PACKED = false>
struct Curve32 {
    const float4 *seg32;  // LDS [NK-1]: cos, sin, 1 / (sin + 1e-10) per segment in single precision (read as one broadcast b128)
    float inv_sin_min;
    float ln2_over_kappa;
    int nseg;
    double kappa;
    static constexpr int kFloats = 2 * NK + 2;
    // max over the segments of the clipped y . nearest, single precision.
    // Per segment (a, b) the reference takes t* = atan2(B, A), clips it to [0, theta_g] and evaluates y . p(t) there
    // (spherical_curve.py:10-32): a.y f for t* <= 0, b.y f for t* >= theta_g, |P y| / (sin theta_g + 1e-10) in between, with
    // f = sin / (sin + 1e-10).  Round 3: the MAXIMUM over the segments needs less than that per segment.  Every knot that
    // starts a segment counts as it is: the segment it starts gives at least a.y f in each of its three branches (interior:
    // the amplitude of y . p(t); t* > theta_g: b.y > a.y), so max_g >= a_i.y f for every i < nseg -- and a knot is the b of
    // the segment before it, so the "b" branch of every segment but the last adds nothing new.  What is left per segment: the
    // interior value where 0 < t* < theta_g (B > 0 and A >= cos theta_g |(A, B)|), and for the LAST segment its b end where
    // the reference's rule picks it.  Same value as the reference's maximum up to f (1e-10 / sin theta_g relative: in the
    // margin, eval_error); continuous across the branch boundaries (t* = 0: |P y| / sin = a.y; t* = theta_g: = b.y), so the
    // Lipschitz bound of the margin holds whichever branch the single-precision evaluation takes.  16 instead of 34 vector
    // instructions a segment (no three-way select, v_sqrt instead of v_rsq and two products).
    // Round 4: TWO segments per instruction.  Segment g needs a_g.y and a_{g+1}.y; with H = NK / 2 and Z_i = (a_i.y, a_{i+H}.y)
    // held as one register pair, segments g and g + H are the two halves of the same packed operations on (Z_g, Z_{g+1}):
    // v_pk_mul_f32 / v_pk_fma_f32 form A, B, A^2 + B^2, cos(theta_g) h and h / sin for both at once (the IEEE operations of
    // the scalar loop, hence its bits; square roots, comparisons, selects and maxima stay one per segment): 8 instead of 15
    // vector instructions a segment, and the 2 NK products of a.y = c a.x + s a.u two at a time.  The constants rest in LDS
    // pair-major: seg32[g] = (cos_g, cos_{g+H}, sin_g, sin_{g+H}), then H pairs (1 / sin_g, 1 / sin_{g+H}).
    // Measured (profiles/r04_ab_packed_segments.log, ms per 10^8 chain-steps): d = 10 20.70 -> 20.27, d = 24 28.24 -> 27.68, d = 50
    // 34.00 -> 33.92; with the copy of the loop for full curves below (and no preloaded constants): d = 10 19.2, d = 24 26.6, d = 50
    // 32.6, d = 200 96.2 -> 94.9.  PACKED in every group kernel; the d = 3 lane kernel (ScreenCurve, unmeasured) keeps the scalar loop.
    static constexpr int kH = NK / 2;  // segments per half (NK = 10: 0..4 | 5..9, the tenth a padding one; NK = 17: 0..7 | 8..15)
    static_assert(6 * kH <= 4 * (NK - 1), "the pair-major constants fit where the per-segment ones were");
    typedef float f2 __attribute__((ext_vector_type(2)));
    template <bool PRELOAD = false>
    __device__ __forceinline__ float best32(const float (&q)[kFloats], float c, float s) const
    {
      if constexpr (PACKED) {
        const f2 c2 = {c, c}, s2 = {s, s};
        auto qv = [&](int i) -> float { return i < NK ? q[i] : 0.0f; };
        auto qu = [&](int i) -> float { return i < NK ? q[NK + i] : 0.0f; };
        auto zpair = [&](int i) -> f2 {  // a.y = fma(c, a.x, s * a.u), as the scalar loop forms it
            const f2 ax = {qv(i), qv(i + kH)}, au = {qu(i), qu(i + kH)};
            return __builtin_elementwise_fma(c2, ax, s2 * au);
        };
        f2 ay = zpair(0);
        const float2 *rd = reinterpret_cast<const float2 *>(seg32 + kH);
        float4 pre[PRELOAD ? kH : 1];
        float2 prer[PRELOAD ? kH : 1];
        if (PRELOAD) {
#pragma unroll
            for (int g = 0; g < kH; ++g) {
                pre[g] = seg32[g];
                prer[g] = rd[g];
            }
        }
        // (a curve of exactly NK knots -- the usual case, 10 knots in the 10-knot build -- needs neither the per-segment "is
        // this segment real" select nor a run-time test for the last segment: a copy of the loop for it, chosen per wavefront)
        auto scan = [&](auto full_c) -> float {
        constexpr bool kFull = decltype(full_c)::value;
        float best = -INFINITY;
#pragma unroll
        for (int g = 0; g < kH; ++g) {
            const float4 sg = PRELOAD ? pre[PRELOAD ? g : 0] : seg32[g];
            const float2 rr = PRELOAD ? prer[PRELOAD ? g : 0] : rd[g];
            const f2 ct = {sg.x, sg.y}, st = {sg.z, sg.w}, rden = {rr.x, rr.y};
            const f2 by = zpair(g + 1);
            const f2 A = ay * st;
            const f2 B = __builtin_elementwise_fma(-ay, ct, by);
            const f2 h2 = __builtin_elementwise_fma(A, A, B * B);
            const f2 h = {__builtin_amdgcn_sqrtf(h2.x), __builtin_amdgcn_sqrtf(h2.y)};
            const f2 cth = ct * h, hr = h * rden;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int seg = g + half * kH;
                if (seg + 1 < NK) {  // (NK = 10: segment 9 does not exist)
                    const float Ah = half ? A.y : A.x, Bh = half ? B.y : B.x, ayh = half ? ay.y : ay.x, byh = half ? by.y : by.x;
                    const bool before_b = Ah >= (half ? cth.y : cth.x);  // t* <= theta_g
                    const bool inside = (Bh > 0.0f) & before_b;
                    float v = inside ? (half ? hr.y : hr.x) : ayh;
                    if (kFull ? seg == NK - 2 : seg == nseg - 1) {       // (uniform) the last segment: b where the reference clips to it
                        const bool at_a = (Bh < 0.0f) | ((Bh == 0.0f) & (Ah >= 0.0f));
                        v = fmaxf(v, (!before_b & !at_a) ? byh : -INFINITY);
                    }
                    best = fmaxf(best, (kFull || seg < nseg) ? v : -INFINITY);
                }
            }
            ay = by;
            __builtin_amdgcn_sched_barrier(0);  // two segments at a time, as before (all of them side by side cost registers the callers do not have)
        }
        return fminf(best, 1.0f);
        };
        if constexpr (!PRELOAD) {  // (the two-wavefront builds, which preload, spill with a second copy of the loop)
            if (nseg == NK - 1) return scan(std::true_type{});
        }
        return scan(std::false_type{});
      } else {

        float best = -INFINITY;
        float ay = fmaf(c, q[0], s * q[NK]);
        // PRELOAD (kernels with registers to spare: two wavefronts per SIMD): all segments' constants are read in one go -- one
        // LDS round trip per evaluation instead of one per segment in front of its first use
        float4 pre[PRELOAD ? NK - 1 : 1];
        if (PRELOAD) {
#pragma unroll
            for (int g = 0; g + 1 < NK; ++g) pre[g] = seg32[g];
        }
#pragma unroll
        for (int g = 0; g + 1 < NK; ++g) {
            const float by = fmaf(c, q[g + 1], s * q[NK + g + 1]);
            const float4 sg = PRELOAD ? pre[PRELOAD ? g : 0] : seg32[g];
            const float ct = sg.x, st = sg.y, rden = sg.z;
            const float A = ay * st;
            const float B = fmaf(-ay, ct, by);
            const float h = __builtin_amdgcn_sqrtf(fmaf(A, A, B * B));
            const bool before_b = A >= ct * h;                   // t* <= theta_g
            const bool inside = (B > 0.0f) & before_b;
            float v = inside ? h * rden : ay;
            if (g == nseg - 1) {                                 // (uniform) the last segment: b where the reference clips to it
                const bool at_a = (B < 0.0f) | ((B == 0.0f) & (A >= 0.0f));
                v = fmaxf(v, (!before_b & !at_a) ? by : -INFINITY);
            }
            best = fmaxf(best, g < nseg ? v : -INFINITY);
            ay = by;
            // two segments at a time (they pair up in v_pk_* instructions); all NK - 1 side by side cost 8 more registers
            // where the callers have none to spare (measured, d = 10 / 50: 30.8 / 64.2 -> 29.8 / 60.9 ms)
            if (g % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (PRELOAD) {  // (the unused fourth word counts as read: ds_read_b128, 4 LDS cycles a wavefront, instead of ds_read_b96, 8)
#pragma unroll
            for (int g = 0; g + 1 < NK; ++g) asm volatile("" ::"v"(pre[g].w));
        }
        return fminf(best, 1.0f);  // (the reference clips every y . nearest to [-1, 1]; the maximum of unit vectors' dots is >= -1)
      }
    }
    // error bound of one best32 evaluation with the coefficients q[0 .. 2 NK)
    __device__ __forceinline__ float eval_error(const float (&q)[kFloats]) const
    {
        float b = 0.0f;
#pragma unroll
        for (int i = 0; i < NK; ++i) b = fmaxf(b, fabsf(q[i]) + fabsf(q[NK + i]));
        const float delta = b * (kSinCosErr32 + 3.0f * kUnit32);                // error of a.y
        return (3.0f * delta + 40.0f * kUnit32 + 2.0e-10f) * inv_sin_min * 1.05f;  // see finish32; 2e-10: the factor f of best32
    }
    // finish32 with the single-precision level of x supplied by the caller (lvl0: within e_lvl0 of the double-precision
    // level of x over kappa -- e.g. the value best32 gave for the accepted try of the previous step) and log U taken in
    // single precision: tau = lvl0 + log2(U) (ln 2 / kappa).  Errors: log2_32 <= kLog2Err32 + 2^-24 |log2 U| (the sum
    // of exponent and mantissa logarithm is rounded once more), ln2_over_kappa and the fma one rounding each.
    // Returns e_eval of these coefficients through `e_eval`.
    __device__ __forceinline__ bool finish32_carried(double u_thr, float (&q)[kFloats], float lvl0, float e_lvl0, float &e_eval) const
    {
        e_eval = eval_error(q);
        const float l2 = log2_32(u_thr);
        const float lk = l2 * ln2_over_kappa;
        const float tau = lvl0 + lk;
        q[2 * NK] = tau;
        const float e_tau = e_lvl0 + (kLog2Err32 + kUnit32 * fabsf(l2)) * ln2_over_kappa + 3.0f * kUnit32 * (fabsf(lk) + fabsf(tau));
        float margin = 1.25f * (e_eval + e_tau + kUnit32 * (fabsf(tau) + 1.0f));
        if (!(u_thr > 1e-290) || !(margin < 0.25f)) margin = INFINITY;
        q[2 * NK + 1] = margin;
        return lvl0 >= -1.0f && lvl0 <= 1.0f;  // (NaN fails)
    }
    // margin and threshold from the rounded coefficients q[0 .. 2 NK); false if the level of x is not finite
    __device__ __forceinline__ bool finish32(double u_thr, float (&q)[kFloats]) const
    {
        float b = 0.0f;
#pragma unroll
        for (int i = 0; i < NK; ++i) b = fmaxf(b, fabsf(q[i]) + fabsf(q[NK + i]));
        // error of a.y: (|a.x| + |a.u|) (eps + 2^-24) + the fma roundings
        const float delta = b * (kSinCosErr32 + 3.0f * kUnit32);
        // one evaluation: 3 delta Lipschitz + ~16 roundings of values <= 2, all over sin(theta_g); v_rsq_f32 relative 2^-22
        const float e_eval = (3.0f * delta + 40.0f * kUnit32 + 2.0e-10f) * inv_sin_min * 1.05f;
        // thr / kappa = level32(x) / kappa + log(U) / kappa: the level of x is evaluated by the same routine at theta = 0
        const float lvl0 = best32(q, 1.0f, 0.0f);
        const double tau = (double)lvl0 + fm::log_fast(u_thr) / kappa;
        q[2 * NK] = (float)tau;
        float margin = 1.25f * (2.0f * e_eval + kUnit32 * (fabsf(q[2 * NK]) + 1.0f));
        if (!(u_thr > 1e-290) || !(margin < 0.25f)) margin = INFINITY;
        q[2 * NK + 1] = margin;
        return lvl0 >= -1.0f && lvl0 <= 1.0f;  // (NaN fails)
    }
    __device__ __forceinline__ int screen(const float (&q)[kFloats], float c, float s) const
    {
        const float g = best32(q, c, s) - q[2 * NK];
        return g < -q[2 * NK + 1] ? -1 : (g > q[2 * NK + 1] ? 1 : 0);
    }
    // stages the single-precision segment constants from the double-precision ones seg[NK-1][4] (cos, sin, 1 / (sin + 1e-10));
    // the caller synchronises before (seg complete) and after
    __device__ void stage(float4 *s32, const double *seg, int nseg_, double kappa_)
    {
        if constexpr (PACKED) {
            auto cs = [&](int g, int w) -> float { return g < NK - 1 ? (float)seg[4 * g + w] : (w == 0 ? 1.0f : 0.0f); };  // (padding: theta = 0)
            for (int g = threadIdx.x; g < kH; g += kBlock) {
                s32[g] = make_float4(cs(g, 0), cs(g + kH, 0), cs(g, 1), cs(g + kH, 1));
                reinterpret_cast<float2 *>(s32 + kH)[g] = make_float2(cs(g, 2), cs(g + kH, 2));
            }
        } else {
            for (int g = threadIdx.x; g < NK - 1; g += kBlock)
                s32[g] = make_float4((float)seg[4 * g], (float)seg[4 * g + 1], (float)seg[4 * g + 2], 0.0f);
        }
        seg32 = s32;
        nseg = nseg_;
        kappa = kappa_;
        ln2_over_kappa = (float)(0.6931471805599453 / kappa_);
        float m = 0.0f;
#pragma unroll
        for (int g = 0; g + 1 < NK; ++g)
            if (g < nseg_) m = fmaxf(m, fabsf((float)seg[4 * g + 2]));
        inv_sin_min = m;
    }
};

// curve-vMF: level = kappa * max_g clip(y . nearest point of segment g) (FastCurve).  In units of the dot product the
// screen compares  max_g xy_g - thr / kappa  with 0 +- margin.  y . nearest is a Lipschitz function of (a_g.y, a_{g+1}.y)
// with constant <= 3 / sin(theta_g) (interior branch: |P y| / sin; end points: a.y itself), whichever branch the
// single-precision evaluation takes, so margin = (3 delta + arithmetic) / min_g sin(theta_g), delta = error of a.y.
template <int D, int NK>
struct ScreenCurve : FastCurve<D, NK> {
    using Base = FastCurve<D, NK>;
    using Coef = typename Base::Coef;
    static constexpr int kCoef32Floats = 2 * NK + 2;  // ax | au | thr / kappa | margin
    static constexpr int kParkSkip = 0, kMinWaves = 1, kTradeMin = 1;
    static constexpr bool kCompact = false, kRegenThr = false, kStageRows = false, kHoldRows = false;
    static constexpr int kNumpyWaves = 2;
    __device__ __forceinline__ void retail(float (&)[kCoef32Floats]) const {}
    __device__ __forceinline__ void refill(const double (&)[D], const double (&)[D], float (&)[kCoef32Floats]) const {}
    Curve32<NK> c32;
    __host__ __device__ static size_t lds_doubles() { return Base::lds_doubles() + 2 * (size_t)(NK - 1); }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        Base::stage(lds, tb);
        __syncthreads();
        c32.stage(reinterpret_cast<float4 *>(lds + Base::lds_doubles()), this->seg, this->nseg, this->kappa);
    }
    __device__ __forceinline__ void coeffs(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double kij = this->knots[i * D + j];
                ax = fma(kij, x[j], ax);
                au = fma(kij, u[j], au);
            }
            cf.ax[i] = ax;
            cf.au[i] = au;
            // one knot at a time: left alone the scheduler hoists all NK * D loads of the knots (2 NK D registers)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ bool make32(const Coef &cf, double u_thr, float (&q)[kCoef32Floats]) const
    {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            q[i] = (float)cf.ax[i];
            q[NK + i] = (float)cf.au[i];
        }
        return finish32(u_thr, q);
    }
    __device__ __forceinline__ bool finish32(double u_thr, float (&q)[kCoef32Floats]) const { return c32.finish32(u_thr, q); }
    __device__ __forceinline__ int screen(const float (&q)[kCoef32Floats], float c, float s) const { return c32.screen(q, c, s); }
    __device__ __forceinline__ double threshold(Coef &cf, const double (&x)[D], const double (&u)[D], double u_thr) const
    {
        return this->make(cf, x, u, 0.0, true) + fm::log_fast(u_thr);
    }
    __device__ __forceinline__ double level_exact(const Coef &cf, double c, double s) const { return this->level(cf, c, s); }
    // The fused forms never hold the 2 NK double-precision coefficients at once (80 registers at NK = 10): a knot's two
    // dots are rounded into the single-precision pack, or fed to the segment algebra, as soon as they are formed.
    __device__ __forceinline__ bool setup32(const double (&x)[D], const double (&u)[D], double u_thr, float (&q)[kCoef32Floats]) const
    {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double kij = this->knots[i * D + j];
                ax = fma(kij, x[j], ax);
                au = fma(kij, u[j], au);
            }
            q[i] = (float)ax;
            q[NK + i] = (float)au;
            __builtin_amdgcn_sched_barrier(0);  // one knot at a time (see coeffs)
        }
        return finish32(u_thr, q);
    }
    // FastCurve::make followed by level(cf, 1, 0) and level(cf, c, s), operation for operation, knot by knot
    __device__ __forceinline__ bool decide(const double (&x)[D], const double (&u)[D], double u_thr, double c, double s) const
    {
        double best0 = -INFINITY, dot0 = 0.0, best1 = -INFINITY, dot1 = 0.0;
        double ay0 = 0.0, ay1 = 0.0;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double kij = this->knots[i * D + j];
                ax = fma(kij, x[j], ax);
                au = fma(kij, u[j], au);
            }
            const double by0 = fma(1.0, ax, 0.0 * au);
            const double by1 = fma(c, ax, s * au);
            if (i > 0) {
                this->segment(i - 1, ay0, by0, best0, dot0);
                this->segment(i - 1, ay1, by1, best1, dot1);
            }
            ay0 = by0;
            ay1 = by1;
            __builtin_amdgcn_sched_barrier(0);
        }
        return this->kappa * dot1 > this->kappa * dot0 + fm::log_fast(u_thr);
    }
};

template <int D, class TP>
struct ScreenChain {
    static constexpr int kQ = TP::kCoef32Floats + (TP::kCoef32Floats & 1);  // padded to whole 64-bit words
    static constexpr int kSkip = TP::kParkSkip;                              // leading floats of q that are not parked
    static_assert(kSkip % 2 == 0, "whole words");
    double x[D], u[D];
    double lo, hi;
    double thr;   // the step's uniform U of the threshold (mcmc.py:389); the threshold itself is formed on demand
    float q[kQ];
    uint32_t n_try;
    int32_t steps_done, row, t, status, err, cursor;
    // Compact targets (TP::kCompact: ScreenBinghamDiag): the last float of q (the margin) is formed again at take-up
    // (TP::retail), the threshold uniform is drawn again for an undecided try (not from a replayed stream: there it is parked)
    // and the retained row follows from the step count -- x, u, lo, hi | q0 q1 | q2 steps_done | n_try flags.
    static constexpr bool kCompact = TP::kCompact;
    static constexpr bool kRegenThr = TP::kRegenThr;  // the threshold uniform is parked only with a replayed stream
    static_assert(!kCompact || (TP::kCoef32Floats == 4 && kSkip == 0 && kRegenThr), "compact layout: three parked floats");
    static constexpr int kWordsNoReplay = kCompact ? 2 * D + 2 + 3 : 2 * D + 2 + (kRegenThr ? 0 : 1) + (kQ - kSkip) / 2 + 2;
    static constexpr int kWords = kWordsNoReplay + 1 + (kRegenThr ? 1 : 0);  // replay: + the cursor (+ the threshold uniform)
};

// d > 10 (round 4: the lane kernels of d = 11 .. 16): one chain per lane, nothing parked -- a second chain's 2 d + ... words
// would hold the kernel to two workgroups per CU, and the registers (x and u alone are 4 d) to two or three wavefronts per
// SIMD anyway: the hardware's switching between wavefronts does what the lanes' own between two chains does below d = 11.
template <int D, class TP>
__host__ __device__ constexpr bool screen_parks()
{
    return D <= 10 && (size_t)ScreenChain<D, TP>::kWords * kBlock * sizeof(double) <= 78 * 1024;  // two workgroups per CU (160 KB)
}
template <int D, class TP, bool REPLAY>
__host__ __device__ constexpr size_t screen_lds_doubles()
{
    return TP::lds_doubles() + kTabLds + (screen_parks<D, TP>() ? (size_t)(REPLAY ? ScreenChain<D, TP>::kWords
                                                                        : ScreenChain<D, TP>::kWordsNoReplay) * kBlock
                                                      : 0);
}

__device__ __forceinline__ void lds_trade(float &a, int32_t &b, unsigned long long *slot)
{
    const unsigned long long mine = (unsigned long long)__float_as_uint(a) | ((unsigned long long)(uint32_t)b << 32);
    const unsigned long long o = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    a = __uint_as_float((uint32_t)o);
    b = (int32_t)(uint32_t)(o >> 32);
}

// n / d and whether d divides n, exactly, for 0 <= n < 2^31 and a wavefront-uniform 0 < d < 2^31 (rcp = 1.0 / d): the
// double-precision quotient is off by at most one either way
__device__ __forceinline__ bool divides(int32_t n, int32_t d, double rcp, int32_t &quot)
{
    int32_t q = (int32_t)((double)n * rcp);
    int32_t r = n - q * d;
    if (r < 0) {
        --q;
        r += d;
    }
    if (r >= d) {
        ++q;
        r -= d;
    }
    quot = q;
    return r == 0;
}

__device__ __forceinline__ void lds_trade(float &a, float &b, unsigned long long *slot)
{
    const unsigned long long mine = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    const unsigned long long o = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    a = __uint_as_float((uint32_t)o);
    b = __uint_as_float((uint32_t)(o >> 32));
}

// (measured: asking for two wavefronts per SIMD at d = 10 makes the curve kernel spill 73 registers: 45 -> 64 ms)
// NUMPY (round 4): the draws come from numpy's own PCG64 / ziggurat stream (RunBlock::rng_state, NumpyDraws) at the replay
// path's consumption points -- the reference's order: d normals, the threshold uniform, theta_0, then a uniform per try that
// is made -- a generator per chain, one chain per lane, the ziggurat tables where second chains would be parked.  The tries are
// screened like the library stream's: the same accept decisions as fast_kernel<.., NUMPY>, hence the reference's chain from its
// seed (tests/test_hip_parity.py::test_reference_chain_from_seed[packed], test_numpy_stream_lane_kernel).
template <int D, class TP, bool REPLAY, bool STATS = false, bool STAGE = false, bool NUMPY = false>
__global__ void __launch_bounds__(kBlock, (STATS || (REPLAY && !NUMPY)) ? 1 : (NUMPY ? TP::kNumpyWaves : TP::kMinWaves))
    screened_kernel(TargetBlock tb, RunBlock a)  // (NUMPY: the generator's state and the ziggurat's temporaries on top of the plain kernel's registers)
{
    static_assert(!NUMPY || (REPLAY && !STAGE), "numpy's stream is a sequential source: it is read where the replay buffer is");
    using V = LaneVec<D>;
    using Chain = ScreenChain<D, TP>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TP tp;
    tp.stage(lds, tb);
    const fm::Tables tab = stage_tables(lds + TP::lds_doubles());
    unsigned long long *park = reinterpret_cast<unsigned long long *>(lds + TP::lds_doubles() + kTabLds) + threadIdx.x;
    NumpyDraws<V> nd;
    if constexpr (NUMPY) nd.stage(lds + TP::lds_doubles() + kTabLds);
    // Retained rows in the reference's (chains, draws, dims) order are 8 D bytes at an 8 D byte stride: for D not a multiple of
    // four a row ends inside a 32-byte sector, and rows that leave one at a time (a chain keeps a row every `thin` steps,
    // milliseconds apart) are written as partial sectors -- 1.39 x the bytes at D = 10.  With one chain per lane the LDS that
    // would park second chains is free: a lane holds a row back (at most kStageP - 1 of them) until the run of rows ends on
    // a sector boundary and stores the run at once (a.stage_rows, set by do_screened_run; same bytes in the same places).
    // A build of its own (STAGE; do_screened_run picks it for Bingham targets at d <= 10 when rows are kept that way): inside the
    // plain kernels the extra state cost the two-chains-per-lane launches of the bench 0.6 % for nothing (measured).
    constexpr int kStageP = (!STAGE || !TP::kHoldRows || D % 4 == 0) ? 1 : ((D % 2 == 0) ? 2 : 4);
    double *stage = reinterpret_cast<double *>(park);  // [kStageP - 1][D][kBlock] doubles (nothing is parked in that mode)
    int32_t n_staged = 0;
    __syncthreads();

    const int32_t n = (int32_t)a.n_chains;
    // This workgroup's work: its chunk of chains for the whole launch -- or, in the sliced partial round of a launch
    // (plan_partial_round, gsss_device.h), the (chunk, step slice) of the ticket it draws.  A chain's step count runs over the
    // launch's steps [s_begin, n_steps): counters of the stream and retained rows need nothing else.
    constexpr int kChunk = (screen_parks<D, TP>() && !NUMPY && !STAGE) ? 2 * kBlock : kBlock;  // chains per workgroup
    __shared__ uint32_t sched_word[4];
    const bool sliced = a.sched != nullptr && (int32_t)blockIdx.x >= a.sched_first;
    uint32_t chunk = blockIdx.x;
    int32_t s_begin = 0, n_steps = (int32_t)a.n_steps;
    bool timed_out = false;
    if (sliced) {
        int32_t len;
        if (!SliceSched::take(a, a.one_per_lane ? kBlock : kChunk, sched_word, chunk, s_begin, len, timed_out)) return;  // (one ticket per workgroup: never)
        n_steps = s_begin + len;
    }
    const bool shrink = a.sampler == GSSS_SHRINK;
    const int32_t thin = (int32_t)a.thin;
    const double rcp_thin = 1.0 / (double)thin;  // (compact chains: the retained row follows from the step count)
    const int32_t max_tries = a.max_tries < (1 << 25) ? a.max_tries : (1 << 25) - 1;  // t shares a word with the flags
    constexpr uint32_t kTryBase = 1u + (uint32_t)((D + 3) / 4);
    constexpr bool kPark = screen_parks<D, TP>() && !NUMPY && !STAGE;  // (NUMPY: a generator per chain; STAGE: built for one chain per lane)
    constexpr int kPerBlock = kPark ? 2 * kBlock : kBlock;
    // (one_per_lane: kBlock chains per workgroup, the lane's second slot stays empty -- a chain id past the ensemble)
    const int32_t id0 = (int32_t)chunk * (a.one_per_lane ? kBlock : kPerBlock) + (int32_t)threadIdx.x;
    const int32_t id1 = a.one_per_lane ? n : id0 + kBlock;

    Chain cur;
    int32_t slot = 0;
    int32_t parked_status = kDone;
    // STATS, one chain per lane: the chain's running statistics on chip for the launch (StatsLane, gsss_device.h) -- the ring of
    // its last projections where a second chain would be parked
    StatsLane<D> sl;
    const bool stats_onchip = STATS && a.stats_onchip > 0;

    auto chain_id = [&]() { return slot ? id1 : id0; };
    auto philox = [&]() {
        PhiloxDraws<V, true> dr;
        dr.tab = tab;
        dr.init(a, chain_id(), D);
        dr.begin_step(a.step_offset + (uint64_t)cur.steps_done);
        return dr;
    };
    auto replay_take = [&]() -> double {
        if constexpr (NUMPY) return nd.next_double();
        if (cur.cursor >= (int32_t)a.replay_stride) {
            cur.err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            return 0.5;
        }
        return a.replay[(size_t)chain_id() * a.replay_stride + cur.cursor++];
    };
    auto count_tries = [&]() {
        const uint32_t sum = cur.n_try + (uint32_t)cur.t;
        if (sum < cur.n_try) cur.err |= GSSS_CHAIN_COUNTER_SATURATED;
        cur.n_try = sum < cur.n_try ? 0xFFFFFFFFu : sum;
    };
    auto needs_service = [](int32_t st) { return st == kPending || is_final(st); };

    auto init = [&]() {
        const int32_t c = chain_id();
        const bool valid = c < n;
        const int32_t cc = valid ? c : 0;
#pragma unroll
        for (int j = 0; j < D; ++j) cur.x[j] = a.state[(size_t)j * n + cc];
#pragma unroll
        for (int j = 0; j < D; ++j) cur.u[j] = 0.0;
#pragma unroll
        for (int i = 0; i < Chain::kQ; ++i) cur.q[i] = 0.0f;
        cur.lo = cur.hi = cur.thr = 0.0;
        cur.n_try = 0u;
        cur.steps_done = s_begin;
        cur.row = s_begin / thin;
        cur.err = 0;
        cur.cursor = 0;
        cur.t = 0;
        cur.status = (valid && n_steps > s_begin) ? kPending : kDone;
        if constexpr (NUMPY) nd.init(a, cc, D);
        if constexpr (STATS) {
            if (stats_onchip && valid) sl.load(a, cc, reinterpret_cast<double *>(park));
        }
        if (sliced && valid) {
            if (SliceSched::dead(a, a.one_per_lane ? kBlock : kChunk)[cc] != 0) cur.status = kDone;  // stopped with an error flag in an earlier slice
            if (timed_out && cur.status != kDone) {                             // cannot happen (SliceSched::take)
                cur.err |= GSSS_CHAIN_MAX_TRIES | GSSS_CHAIN_COUNTER_SATURATED;
                cur.status = kDone;
            }
        }
    };

    // everything a step needs before its first try (mcmc.py:387-392); arithmetic identical to fast_kernel's
    auto setup = [&]() {
        double u_thr, u_th0;
        uint32_t w_phi = 0u;  // S^2, Philox stream: the angle word of the tangent direction
        if (REPLAY) {
            if constexpr (NUMPY) {
#pragma unroll
                for (int j = 0; j < D; ++j) cur.u[j] = nd.standard_normal();
            } else if (cur.cursor + D <= (int32_t)a.replay_stride) {
#pragma unroll
                for (int j = 0; j < D; ++j)
                    cur.u[j] = a.replay[(size_t)chain_id() * a.replay_stride + cur.cursor + j];
                cur.cursor += D;
            } else {
#pragma unroll
                for (int j = 0; j < D; ++j) cur.u[j] = 0.5;
                cur.cursor = (int32_t)a.replay_stride;
                cur.err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            }
            u_thr = replay_take();
            u_th0 = shrink ? replay_take() : 0.0;
        } else {
            const PhiloxDraws<V, true> dr = philox();
            if constexpr (D == 3) {
                dr.step_s2(u_thr, u_th0, w_phi);
            } else {
                dr.normals(cur.u, 0);
                dr.block(0u, u_thr, u_th0);
            }
        }
        bool x_ok;
        if constexpr (D == 3 && !REPLAY) {  // Philox stream on S^2: the unit tangent is drawn directly (tangent3, gsss_device.h)
            const double xx = vdot<V>(cur.x, cur.x);
            x_ok = xx < INFINITY;
            const double rnx = inv_norm(xx);
            const double nrm[3] = {cur.x[0] * rnx, cur.x[1] * rnx, cur.x[2] * rnx};
            philox().tangent(nrm, cur.u, 0, w_phi);
        } else
        {  // u = spherical_projection(z, x), sphere.py:29-33
            const double xx = vdot<V>(cur.x, cur.x);
            x_ok = xx < INFINITY;  // a NaN / Inf state: the curve's clipped level swallows NaN (v_max), so the state itself is looked at
            const double rnx = inv_norm(xx);
            double cz = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) cz = fma(cur.u[j], cur.x[j] * rnx, cz);
#pragma unroll
            for (int j = 0; j < D; ++j) cur.u[j] = fma(-cz, cur.x[j] * rnx, cur.u[j]);
            const double rnw = inv_norm(vdot<V>(cur.u, cur.u));
#pragma unroll
            for (int j = 0; j < D; ++j) cur.u[j] *= rnw;
        }
        cur.thr = u_thr;  // the uniform; the double-precision threshold is formed only if a try stays undecided
        const bool finite = tp.setup32(cur.x, cur.u, u_thr, reinterpret_cast<float (&)[TP::kCoef32Floats]>(cur.q));
        // verification (GSSS_VARIANT_FAST_VERIFY): an infinite margin leaves EVERY try undecided -- each is then decided in double
        // precision by decide(), and the run must reproduce the screened one bit for bit
        // (built into the kernels of d = 11 .. 16, which have no all-double lane sibling to be compared with; d <= 10 is held to
        // fast_kernel bit for bit instead, and a launch-wide flag kept live through the main loop cost those kernels 0.5 %)
        if constexpr (D > 10) {
            if (a.screen == 2) cur.q[TP::kCoef32Floats - 1] = INFINITY;
        }
        if (shrink) {
            cur.hi = kTwoPi * u_th0;
            cur.lo = cur.hi - kTwoPi;
        } else {
            cur.lo = 0.0;
            cur.hi = kTwoPi;
        }
        cur.t = 0;
        cur.status = kReady;
        if (!finite || !x_ok) {
            cur.err |= GSSS_CHAIN_NONFINITE;
            cur.status = kDone;
        }
    };

    // up to FOUR proposals, screened in single precision: a try's uniform is one 32-bit word of the stream (philox-v3, round 5), block
    // j feeds tries 4j .. 4j + 3.  (Rounds 2-4: two 53-bit uniforms a block, a pair of tries an attempt -- the Philox block was half of
    // an attempt's cycles: README mixture 24.3 -> 20.7 ms per 10^9 chain-steps, K = 10 kappa = 500 46.8 -> 42.4, profiles/r05_ab_try32.log.)
    // A replayed stream hands over doubles: two tries an attempt, as before.
    constexpr bool kTry32 = !REPLAY;
    constexpr int kPerAttempt = kTry32 ? 4 : 2;
    auto attempt = [&]() {
        if (cur.t >= max_tries) {
            count_tries();
            cur.err |= GSSS_CHAIN_MAX_TRIES;
            cur.status = kDone;
            return;
        }
        double u_pair[2];
        uint32_t w_try[4];
        if constexpr (kTry32) {
            philox().words(kTryBase + (uint32_t)(cur.t >> 2), w_try);
        } else if (!REPLAY) {
            philox().block(kTryBase + (uint32_t)(cur.t >> 1), u_pair[0], u_pair[1]);
        }
        // (odd / mid-block: the tries before this one were made already -- the last of them decided in double precision)
        const int first = REPLAY ? 0 : (cur.t & (kPerAttempt - 1));
        bool stopped = false;
#pragma unroll
        for (int h = 0; h < kPerAttempt; ++h) {
            if (!stopped && h >= first && (h == first || cur.t < max_tries) &&
                !(REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED))) {
                double uu;
                if constexpr (kTry32)
                    uu = try_uniform(w_try[h]);
                else
                    uu = REPLAY ? replay_take() : u_pair[h & 1];
                if (REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED)) {
                    stopped = true;  // nothing left to propose with
                } else {
                    const double theta = fma(cur.hi - cur.lo, uu, cur.lo);  // mcmc.py:395
                    ++cur.t;
                    float s32, c32;
                    sincos_rev32(theta, s32, c32);
                    const int verdict = tp.screen(reinterpret_cast<const float (&)[TP::kCoef32Floats]>(cur.q), c32, s32);
                    // (selects, not branches: four exec-mask regions per try otherwise)
                    // a rejected try shrinks the bracket (mcmc.py:400); a stopped one leaves theta in the end it would become
                    const bool rej = verdict < 0, neg = theta < 0.0, moves = !rej | shrink;
                    cur.lo = (moves & neg) ? theta : cur.lo;
                    cur.hi = (moves & !neg) ? theta : cur.hi;
                    cur.status = rej ? cur.status : ((verdict > 0 ? kFinalAccept : kFinalDecide) + (neg ? 0 : kFinalInHi));
                    stopped = !rej;
                }
            }
        }
        if (REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED) && cur.status == kReady) {
            count_tries();
            cur.status = kDone;
        }
    };

    // the rows held back in LDS go out in front of the row that starts at `p_next` (rows of one chain are contiguous)
    auto unstage = [&](double *p_next) {
#pragma unroll
        for (int b = 0; b < kStageP - 1; ++b) {
            if (b < n_staged) {
                double *q = p_next - (size_t)(n_staged - b) * D;
#pragma unroll
                for (int j = 0; j < D; ++j) q[j] = stage[(size_t)(b * D + j) * kBlock];
            }
        }
        n_staged = 0;
    };
    // the double-precision part of a stopped try: decide it if the screen could not, then move (mcmc.py:396-399)
    auto finalise = [&]() {
        const double theta = cur.status >= kFinalAccept + kFinalInHi ? cur.hi : cur.lo;
        double sn, cs;
        fm::sincos_tab(theta, tab, sn, cs);
        bool accepted = true;
        if (is_decide(cur.status)) {  // rare: the double-precision test itself (mcmc.py:389, 397)
            double u_thr = cur.thr;
            if constexpr (Chain::kRegenThr && !REPLAY) {  // not parked: the step's threshold uniform, drawn again
                double u_th0;
                if constexpr (D == 3) {
                    uint32_t w_phi;
                    philox().step_s2(u_thr, u_th0, w_phi);
                } else {
                    philox().block(0u, u_thr, u_th0);
                }
            }
            accepted = tp.decide(cur.x, cur.u, u_thr, cs, sn);
        }
        if (accepted) {
#pragma unroll
            for (int j = 0; j < D; ++j) cur.x[j] = fma(sn, cur.u[j], cs * cur.x[j]);  // mcmc.py:396
            count_tries();
            ++cur.steps_done;
            bool keep = false;
            if (a.samples != nullptr || STATS) {
                if constexpr (Chain::kCompact) {
                    int32_t rows;
                    keep = divides(cur.steps_done, thin, rcp_thin, rows);
                    cur.row = rows - 1;
                } else {
                    keep = cur.steps_done == (cur.row + 1) * thin;
                }
            }
            if (keep) {
                if (a.samples != nullptr) {
                    if (kStageP > 1 && a.stage_rows) {
                        double *p = &a.samples[sample_index(a, cur.row, 0, D, chain_id())];  // chain-major: the row is contiguous
                        const bool ends_on_sector = (reinterpret_cast<uintptr_t>(p + D) & 31u) == 0u;
                        if (!ends_on_sector && n_staged < kStageP - 1) {
#pragma unroll
                            for (int j = 0; j < D; ++j) stage[(size_t)(n_staged * D + j) * kBlock] = cur.x[j];
                            ++n_staged;
                        } else {
                            unstage(p);
#pragma unroll
                            for (int j = 0; j < D; ++j) p[j] = cur.x[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < D; ++j) a.samples[sample_index(a, cur.row, j, D, chain_id())] = cur.x[j];
                    }
                }
                if constexpr (STATS) {
                    if (stats_onchip)
                        sl.draw(a, chain_id(), cur.x);
                    else
                        stats_update<D>(a, chain_id(), cur.x);
                }
                ++cur.row;
            }
            const bool exhausted = REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED);
            cur.status = (cur.steps_done < n_steps && !exhausted) ? kPending : kDone;
        } else {
            // mcmc.py:400: the bracket already ends at theta; the rejection sampler's fixed bracket (0, 2 pi) is restored
            if (!shrink) cur.hi = kTwoPi;
            cur.status = kReady;
        }
    };

    auto pack_flags = [&]() { return cur.t | (cur.status << 25) | (int32_t)((uint32_t)cur.err << 28); };
    auto trade = [&]() {
        unsigned long long *p = park;
        auto word = [&](double &v) {
            lds_trade(v, p);
            p += kBlock;
        };
#pragma unroll
        for (int j = 0; j < D; ++j) word(cur.x[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) word(cur.u[j]);
        word(cur.lo);
        word(cur.hi);
        if constexpr (Chain::kCompact) {
            if (REPLAY) word(cur.thr);
            lds_trade(cur.q[0], cur.q[1], p);
            p += kBlock;
            lds_trade(cur.q[2], cur.steps_done, p);
            p += kBlock;
        } else {
            if (!Chain::kRegenThr || REPLAY) word(cur.thr);
#pragma unroll
            for (int i = Chain::kSkip; i < Chain::kQ; i += 2) {
                lds_trade(cur.q[i], cur.q[i + 1], p);
                p += kBlock;
            }
            lds_trade(cur.steps_done, cur.row, p);
            p += kBlock;
        }
        if (REPLAY) {
            int32_t zero = 0;
            lds_trade(cur.cursor, zero, p);
            p += kBlock;
        }
        const int32_t st = cur.status;
        int32_t nt = (int32_t)cur.n_try, packed = pack_flags();
        lds_trade(nt, packed, p);
        cur.n_try = (uint32_t)nt;
        cur.t = packed & 0x1FFFFFF;
        cur.status = (packed >> 25) & 7;
        cur.err = (int32_t)((uint32_t)packed >> 28);
        parked_status = st;
        slot ^= 1;
        // the chain taken up tries next -- or may, if its undecided try is decided a rejection: its unparked coefficients
        // are formed again (a chain that waits for set-up or for its move gets new ones there / needs none)
        if (Chain::kSkip > 0 && (cur.status == kReady || is_decide(cur.status))) tp.refill(cur.x, cur.u, reinterpret_cast<float (&)[TP::kCoef32Floats]>(cur.q));
        if constexpr (Chain::kCompact) tp.retail(reinterpret_cast<float (&)[TP::kCoef32Floats]>(cur.q));  // the margin
    };

    // A slice's state, counters and flags go to the chunk's next slice: written through (SliceSched::hand_over), so that
    // publishing them needs no write-back of the whole L2 (the statistics build hands its accumulators over with plain stores
    // and keeps the release).
#ifndef GSSS_HAND_OVER_PLAIN  // (A/B builds: 1 = plain stores and a release, as the group kernels do)
#define GSSS_HAND_OVER_PLAIN 0
#endif
    constexpr bool kThrough = !STATS && !GSSS_HAND_OVER_PLAIN;
    auto put_out = [&](auto *p, auto v) {
        if (kThrough && sliced)
            SliceSched::hand_over(p, v);
        else
            *p = v;
    };
    auto flush = [&]() {
        const int32_t c = chain_id();
        if (c >= n) return;
        // (the chain's next row starts behind them: row steps_done / thin -- compact chains derive cur.row from the step count at
        // every move, where it names the LAST kept row until the next one is due)
        if (kStageP > 1 && n_staged > 0) unstage(&a.samples[sample_index(a, cur.steps_done / thin, 0, D, c)]);
        if constexpr (NUMPY) nd.finish(a, c, true);
        if constexpr (STATS) {
            if (stats_onchip) sl.store(a, c);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) put_out(&a.state[(size_t)j * n + c], cur.x[j]);
        if (a.n_reject) put_out(&a.n_reject[c], a.n_reject[c] + ((int64_t)cur.n_try - (cur.steps_done - s_begin)));
        if (a.n_tries) put_out(&a.n_tries[c], a.n_tries[c] + (int64_t)cur.n_try);
        if (a.err && cur.err) put_out(&a.err[c], a.err[c] | cur.err);
        if (sliced && cur.steps_done < n_steps) put_out(&SliceSched::dead(a, a.one_per_lane ? kBlock : kChunk)[c], (int32_t)1);  // stopped early: stays stopped
    };

    if (kPark && !a.one_per_lane) {  // the chain of slot 1 is initialised, set up and parked; then the chain of slot 0
        // (one_per_lane: the launch carries no LDS for parked chains)
        slot = 1;
        init();
        if (cur.status == kPending) setup();
        unsigned long long *p = park;
        auto put = [&](double v) {
            *p = (unsigned long long)__double_as_longlong(v);
            p += kBlock;
        };
        auto put2 = [&](uint32_t lo32, uint32_t hi32) {
            *p = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
            p += kBlock;
        };
#pragma unroll
        for (int j = 0; j < D; ++j) put(cur.x[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) put(cur.u[j]);
        put(cur.lo);
        put(cur.hi);
        if constexpr (Chain::kCompact) {
            if (REPLAY) put(cur.thr);
            put2(__float_as_uint(cur.q[0]), __float_as_uint(cur.q[1]));
            put2(__float_as_uint(cur.q[2]), (uint32_t)cur.steps_done);
        } else {
            if (!Chain::kRegenThr || REPLAY) put(cur.thr);
#pragma unroll
            for (int i = Chain::kSkip; i < Chain::kQ; i += 2) put2(__float_as_uint(cur.q[i]), __float_as_uint(cur.q[i + 1]));
            put2((uint32_t)cur.steps_done, (uint32_t)cur.row);
        }
        if (REPLAY) put2((uint32_t)cur.cursor, 0u);
        put2(cur.n_try, (uint32_t)pack_flags());
        parked_status = cur.status;
    }
    slot = 0;
    init();
    if (cur.status == kPending) setup();

    bool stuck = false;
    for (;;) {
        // a lane whose current chain cannot try (stopped, waiting or finished) takes its other chain when that one can
        if (TP::kTradeMin > 1) {
            const bool want = kPark && cur.status != kReady && parked_status == kReady;
            const unsigned long long wants = __ballot(want), can = __ballot(cur.status == kReady);
            if (want && (__popcll(wants) >= TP::kTradeMin || can == 0ull)) trade();
        } else if (kPark && cur.status != kReady && parked_status == kReady)
            trade();
        const unsigned long long trying = __ballot(cur.status == kReady);
        if (cur.status == kReady) attempt();
        const unsigned long long live = __ballot(cur.status != kDone || parked_status != kDone);
        if (live == 0ull) break;
        const unsigned long long waiting = __ballot(needs_service(cur.status));  // nothing to try until serviced
        const unsigned long long pend = __ballot(needs_service(cur.status) || needs_service(parked_status));
        const int n_live = __popcll(live);
        // the service phase (double precision) costs several try iterations (single precision): it runs when three
        // quarters of the live lanes wait for it (or seven eighths have something for it)
#ifndef GSSS_SVC_WAIT_NUM
#define GSSS_SVC_WAIT_NUM 3
#define GSSS_SVC_WAIT_DEN 4
#endif
#ifndef GSSS_SVC_PEND_NUM
#define GSSS_SVC_PEND_NUM 7
#define GSSS_SVC_PEND_DEN 8
#endif
        const bool service = pend != 0ull && (GSSS_SVC_WAIT_DEN * __popcll(waiting) >= GSSS_SVC_WAIT_NUM * n_live ||
                                              GSSS_SVC_PEND_DEN * __popcll(pend) >= GSSS_SVC_PEND_NUM * n_live);
        if (service) {
            if (kPark && !needs_service(cur.status) && needs_service(parked_status)) trade();  // bring the waiting chain in
            if (is_final(cur.status)) finalise();
            if (cur.status == kPending) setup();
        }
        if (trying == 0ull && !service && __ballot(kPark && cur.status != kReady && parked_status == kReady) == 0ull) {
            stuck = true;  // no lane tried, nothing was serviced, nothing to trade in: cannot happen; never spin on the GPU
            break;
        }
    }
    if (stuck && cur.status != kDone) cur.err |= GSSS_CHAIN_MAX_TRIES | GSSS_CHAIN_COUNTER_SATURATED;
    // every lane stores its chain of slot 0, then its chain of slot 1: a wavefront's stores are then whole 512-byte runs (lanes that
    // ended on different slots wrote every line in two halves -- twice the bytes once the hand-over is written through: K = 10
    // mixture 76 bytes per chain and slice where 40 are due)
    if (kPark && slot == 1) trade();
    flush();
    if (kPark && !a.one_per_lane) {
        trade();
        flush();
    }
    if (sliced) SliceSched::publish<kThrough>(a, sched_word);
}

// numpy's stream through the screened kernel: one lane per chain, unsliced (a generator's state lives in its lane for the launch)
template <int D, class TP>
int do_screened_numpy(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    const size_t lds = (TP::lds_doubles() + kTabLds + NumpyDraws<LaneVec<D>>::kLdsDoubles) * sizeof(double);
    auto kern = rb.stats != nullptr ? screened_kernel<D, TP, true, true, false, true> : screened_kernel<D, TP, true, false, false, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    RunBlock rbl = rb;
    rbl.one_per_lane = 1;
    rbl.stage_rows = 0;
    rbl.sched = nullptr;
    const int64_t grid = (rb.n_chains + kBlock - 1) / kBlock;
    last_launch() = LaunchInfo{grid, 0, 0.0};
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rbl);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("screened kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

template <int D, class TP, bool REPLAY>
int do_screened_run(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    size_t lds = screen_lds_doubles<D, TP, REPLAY>() * sizeof(double);
    auto kern = screened_kernel<D, TP, REPLAY, false>;
    if constexpr (!REPLAY) {  // running statistics: a build of its own (the plain kernel carries none of it)
        if (rb.stats != nullptr) kern = screened_kernel<D, TP, false, true>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    int per_block = screen_parks<D, TP>() ? 2 * kBlock : kBlock;
    // Small and mid-size ensembles run ONE chain per lane (256-chain workgroups, the lane's second slot empty): a second
    // wavefront somewhere else is worth more than a lane's second chain, and rounds of workgroups half as long fill the chip
    // more evenly.  Measured over the ensemble size (tools/bench_packing.py, 2000 steps per launch; r2 = rounds of workgroups
    // with two chains per lane): one per lane wins or ties up to r2 ~ 1.35 for every kernel (README mixture, 1280 resident
    // workgroups: 131 072 chains 1.8 -> 2.7e10 chain-steps/s, 327 680 2.9 -> 3.6e10, 458 752 3.3 -> 3.9e10, 786 432 3.8 -> 4.1e10);
    // beyond, two per lane is 2-10 % ahead for the README kernel (917 504 .. 1 048 576 chains: 4.2-4.4 against 4.0-4.1e10) and
    // within a few per cent either way for the others: two per lane from r2 = 1.35 on.
    bool one_per_lane = false;
    SlicePlan plan;
    int32_t first = 0;
    const char *env_one = getenv("GSSS_ONE_PER_LANE");  // "0": always two chains per lane, "2": always one (tests, measurements)
    if (!(env_one && env_one[0] == '0') && screen_parks<D, TP>() && !REPLAY) {
        int per_cu = 0;
        int64_t resident = 0;
        if (env_one && env_one[0] == '2')
            one_per_lane = true;
        else if ((resident = resident_workgroups(reinterpret_cast<const void *>(kern), lds, &per_cu)) >= 1) {
            const int64_t b2 = (rb.n_chains + 2 * kBlock - 1) / (2 * kBlock);
            one_per_lane = 20 * b2 < 27 * resident;
            // ... and at ANY size where the LDS of the parked chains is what limits the workgroups per CU: without it more
            // wavefronts are resident, and the hardware's switching between them beats the lanes' own between two chains
            // (tools/bench_packing_shapes.py, 10^6 chains: vMF mixtures d = 10 K = 3 / 5 / 10 1.84 -> 2.35 / 1.62 -> 1.91 / 1.33 ->
            // 1.60e10 chain-steps/s, d = 8 K = 3 2.09 -> 2.58e10; equal occupancy: two per lane ahead by up to 10 %)
            int per_cu_one = 0;
            if (!one_per_lane && resident_workgroups(reinterpret_cast<const void *>(kern), (TP::lds_doubles() + kTabLds) * sizeof(double), &per_cu_one) >= 1)
                one_per_lane = per_cu_one > per_cu;
        }
    }
    // The one-chain-per-lane BUILD (below) is ahead of two chains per lane at 10^6 chains too for most shapes that have it
    // (TP::kStageRows): TP::kPreferOne says where.  Bingham targets (tools/bench_pure_one.py, profiles/r04_bingham_pure_one.log):
    // eigenbasis d = 5 .. 7, 9, 10 +3 .. 5 %, dense d = 3 .. 5, 7 +4 %, d = 8 +22 %, d = 10 +8 %; behind at eigenbasis d = 4 (-2 %) and
    // dense d = 6 (-10 %: 170 registers, two wavefronts).  Mixtures (tools/bench_vmf_pure_one.py, profiles/r04_vmf_pure_one.log):
    // ScreenVmf::kOneAhead.
    if constexpr (!REPLAY && TP::kStageRows && D <= 10) {
        if (TP::kPreferOne && !(env_one && env_one[0] == '0') && rb.stats == nullptr) one_per_lane = true;
    }
    // Running statistics (round 5): one chain per lane with the launch's working set on chip -- the accumulators and the lag sums
    // in registers, the ring of the last L projections in the LDS a parked chain would take (StatsLane, gsss_device.h: the same
    // bits as the per-draw read-modify-write) -- for L <= kStatsMaxLags lags, wherever the ring fits the LDS.
    // GSSS_STATS_ONCHIP=0: the per-draw path (A/B, tests).
    int32_t stats_onchip = 0;
    size_t stats_ring = 0;  // doubles of LDS
    if constexpr (!REPLAY) {
        const char *env_st = getenv("GSSS_STATS_ONCHIP");
        if (rb.stats != nullptr && !(env_st && env_st[0] == '0') && rb.stats_lags <= kStatsMaxLags) {
            const size_t ring = (size_t)rb.stats_lags * kBlock;
            if ((TP::lds_doubles() + kTabLds + ring) * sizeof(double) <= (size_t)160 * 1024) {  // (a workgroup's LDS on gfx950)
                stats_onchip = 1;
                stats_ring = ring;
                one_per_lane = true;
            }
        }
    }
    bool stage_rows = false;
    if (one_per_lane) {
        per_block = kBlock;
        lds = (TP::lds_doubles() + kTabLds + stats_ring) * sizeof(double);  // nothing is parked: no LDS for it (statistics: the ring)
        // A build of the kernel for ONE chain per lane (screened_kernel<.., STAGE>: no code for a parked chain -- Bingham d = 10: 154
        // instead of 143 registers, but nothing of the trade logic in the loop: 10^6 chains, 39.3 -> 36.85 ms, ahead of two chains
        // per lane at 37.45), which for the Bingham targets (TP::kHoldRows) also holds chain-major retained rows that are not whole
        // sectors back in LDS until their run is (kStageP > 1; only where that LDS costs no resident workgroup)
        if constexpr (!REPLAY && TP::kStageRows && D <= 10) {
            if (rb.stats == nullptr) {
                auto kern_one = screened_kernel<D, TP, false, false, true>;
                kern = kern_one;
                constexpr int kStageP = (!TP::kHoldRows || D % 4 == 0) ? 1 : ((D % 2 == 0) ? 2 : 4);
                const char *env_stage = getenv("GSSS_STAGE_ROWS");  // "0": off (A/B)
                if (kStageP > 1 && rb.samples != nullptr && rb.keep_rows > 0 && !(env_stage && env_stage[0] == '0')) {
                    const size_t lds_staged = lds + (size_t)(kStageP - 1) * D * kBlock * sizeof(double);
                    if (resident_workgroups(reinterpret_cast<const void *>(kern_one), lds_staged) >= resident_workgroups(reinterpret_cast<const void *>(kern_one), lds)) {
                        // (the attribute first: rows are held back only if the launch can have the LDS for them)
                        const bool can = lds_staged <= 48 * 1024 ||
                                         hipFuncSetAttribute(reinterpret_cast<const void *>(kern_one), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged) == hipSuccess;
                        if (can) {
                            stage_rows = true;
                            lds = lds_staged;
                        } else {
                            (void)hipGetLastError();
                        }
                    }
                }
            }
        }
    }
    if (lds > 48 * 1024) {  // the kernel that IS launched, with the LDS it is launched with (the one-chain-per-lane build got none)
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu B of LDS) failed: %s", lds, hipGetErrorString(ea));
            return GSSS_E_HIP;
        }
    }
    const int64_t n_chunks = (rb.n_chains + per_block - 1) / per_block;
    // a small last round of workgroups is cut into step slices (plan_partial_round, gsss_device.h)
    plan = plan_partial_round(kern, lds, rb, n_chunks, !REPLAY && screen_parks<D, TP>(), st, first);
    RunBlock rbl = rb;
    rbl.one_per_lane = one_per_lane ? 1 : 0;
    rbl.stats_onchip = stats_onchip;
    rbl.stage_rows = stage_rows ? 1 : 0;
    rbl.sched = plan.ws;
    rbl.slice_steps = plan.slice_steps;
    rbl.sched_first = first;
    const int64_t grid = plan.grid;
    if (getenv("GSSS_DEBUG_OCCUPANCY")) {  // (tuning aid: what the runtime says about residency)
        int per_cu = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds);
        fprintf(stderr, "gsss: screened kernel: %zu B of LDS per workgroup, %d workgroups per CU, grid %lld\n", lds, per_cu, (long long)grid);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rbl);
    hipError_t e = hipGetLastError();
    if (plan.ws) (void)hipFreeAsync(plan.ws, st);
    if (e != hipSuccess) {
        set_error("screened kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

}  // namespace gsss

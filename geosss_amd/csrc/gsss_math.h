// gsss_math.h -- double-precision elementary functions specialised for the sampler's hot loop.
//
// The generic device-library sincos/exp carry large-argument reduction and double-double
// corrections the shrinkage loop never needs: theta always lies in [-2 pi, 2 pi] and the
// exponents are differences of log-densities.  These versions are ~1 ulp accurate on their
// stated domains (tests/test_math.py measures them against libm on the host build) and cost
// a third of the instructions.  Everything is plain FMA arithmetic, identical on host and device.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSSS_HD __host__ __device__ __forceinline__
#else
#define GSSS_HD inline
#endif

namespace gsss {
namespace fm {

// A double constant for the addend slot of an FMA.  On the device it is materialised in a scalar register
// pair by two s_mov_b32 right where it is used: scalar instructions issue beside the other wavefronts' vector
// work, whereas hipcc, left alone, builds a single-use 64-bit addend in VECTOR registers (two v_mov_b32 in
// front of every Horner step: +100 % vector issue on the polynomial) or hoists it out of the loop into a
// long-lived register (the sampler loops then spill scalar registers into vector lanes).
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ __forceinline__ double kc()
{
    int lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((int)(uint32_t)(BITS & 0xffffffffull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((int)(uint32_t)(BITS >> 32)));
    return __hiloint2double(hi, lo);
}
#else
template <uint64_t BITS>
inline double kc()
{
    double d;
    const uint64_t b = BITS;
    __builtin_memcpy(&d, &b, sizeof(d));
    return d;
}
#endif

#define GSSS_KC(x) ::gsss::fm::kc<__builtin_bit_cast(uint64_t, (double)(x))>()

// a * b + k with the constant k read straight from its scalar register pair (one VOP3 instruction)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_k(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#else
inline double fma_k(double a, double b, double k) { return fma(a, b, k); }
#endif
#define GSSS_FMAK(a, b, k) ::gsss::fm::fma_k((a), (b), GSSS_KC(k))

// sin / cos kernels on |r| <= pi/4 (classic fdlibm minimax polynomials, |err| < 2^-57)
GSSS_HD double sin_kernel(double r)
{
    const double z = r * r;
    double p = 1.58969099521155010221e-10;
    p = GSSS_FMAK(p, z, -2.50507602534068634195e-08);
    p = GSSS_FMAK(p, z, 2.75573137070700676789e-06);
    p = GSSS_FMAK(p, z, -1.98412698298579493134e-04);
    p = GSSS_FMAK(p, z, 8.33333333332248946124e-03);
    p = GSSS_FMAK(p, z, -1.66666666666666324348e-01);
    return fma(r * z, p, r);
}

GSSS_HD double cos_kernel(double r)
{
    const double z = r * r;
    double p = -1.13596475577881948265e-11;
    p = GSSS_FMAK(p, z, 2.08757232129817482790e-09);
    p = GSSS_FMAK(p, z, -2.75573143513906633035e-07);
    p = GSSS_FMAK(p, z, 2.48015872894767294178e-05);
    p = GSSS_FMAK(p, z, -1.38888888888741095749e-03);
    p = GSSS_FMAK(p, z, 4.16666666666666019037e-02);
    // 1 - z/2 + z^2 p, with the 1 - z/2 rounding error folded back in
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * z * p);
}

GSSS_HD void quadrant_select(int q, double s, double c, double &so, double &co)
{
    const bool swap = q & 1;
    const double a = swap ? c : s;
    const double b = swap ? s : c;
    so = (q & 2) ? -a : a;
    co = ((q + 1) & 2) ? -b : b;
}

// sin and cos of x for |x| <= 8 (three-bit quadrant index, two-term Cody-Waite reduction)
GSSS_HD void sincos_small(double x, double &so, double &co)
{
    const double k = rint(x * 6.36619772367581382433e-01);  // 2/pi
    double r = fma(-k, 1.57079632673412561417e+00, x);       // pi/2 high 33 bits: k * it is exact
    r = fma(-k, 6.07710050650619224932e-11, r);              // pi/2 - high part
    quadrant_select((int)k & 3, sin_kernel(r), cos_kernel(r), so, co);
}

// sin and cos of 2 pi u for u in [0, 1): exact octant reduction in units of turns
GSSS_HD void sincos_2pi(double u, double &so, double &co)
{
    const double t = 4.0 * u;
    const double k = rint(t);
    const double r = (t - k) * 1.57079632679489661923;  // (t - k) is exact
    quadrant_select((int)k & 3, sin_kernel(r), cos_kernel(r), so, co);
}

// exp(x) for |x| < 2^20 without the range clamp and NaN handling of exp_fast: the arguments of the
// mixture's accept test are bounded by construction (differences of log-densities of points on the
// sphere); a non-finite state is caught when a step is set up.  Same polynomial, same rounding.
GSSS_HD double exp_bounded(double x)
{
    const double n = rint(x * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878680989792e-09);
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// exp(x) for any finite x (and -inf): n = round(x / ln 2), degree-13 Taylor on |r| <= ln2/2
// (truncation 4e-18), scaled by 2^n with ldexp (gradual underflow, overflow to +inf)
GSSS_HD double exp_fast(double x)
{
    const double xc = fmin(fmax(x, -800.0), 800.0);
    const double n = rint(xc * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, xc);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;       // 1/13!
    p = fma(p, r, 2.08767569878680989792e-09);  // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);  // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);  // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);  // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);  // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);  // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);  // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);  // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);  // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);  // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return x == x ? ldexp(p, (int)n) : x;  // NaN stays NaN
}

// log(x) for finite normal x > 0 (fdlibm's s = f/(2+f) series, |err| < 1 ulp); x = 0 -> -inf
GSSS_HD double log_fast(double x)
{
    if (!(x > 0.0)) return x == 0.0 ? -INFINITY : NAN;
    int e;
    double m = frexp(x, &e);  // m in [0.5, 1)
    if (m < 7.07106781186547524401e-01) {
        m *= 2.0;
        e -= 1;
    }
    const double f = m - 1.0;  // [-0.2929, 0.4142]
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * GSSS_FMAK(w, GSSS_FMAK(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * GSSS_FMAK(w, GSSS_FMAK(w, GSSS_FMAK(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// ------------------------------------------------------------------------------------------
// Table-driven variants for the throughput kernels: 64 points of the circle and 93 points of log on
// [sqrt(1/2), sqrt(2)] (2.5 KB, staged in LDS by the kernel) cut the polynomial degrees: sincos 21 instead of 38
// instructions, the logarithm of a 32-bit uniform 20 instead of 35 (no division).  Accuracy ~1 ulp like the polynomial
// versions (tests/test_math.py measures both).  Table entries: sc[j] = (sin, cos)(2 pi j / 64),
// lg[j - 90] = (log m_j, 1 / m_j), m_j = j / 128, j = 90 .. 182.
// ------------------------------------------------------------------------------------------
struct Tables {
    const double *sc;  // [64][2]
    const double *lg;  // [93][2]
};
constexpr int kLogTableLo = 90, kLogTableN = 93;
constexpr int kTableDoubles = 2 * 64 + 2 * kLogTableN;

// fills `buf` (kTableDoubles doubles); entry i of the combined list is computed by one caller (i = 0 .. 64 + 129 - 1)
GSSS_HD void table_entry(double *buf, int i)
{
    if (i < 64) {
        double sn, cs;
        sincos_2pi((double)i * (1.0 / 64.0), sn, cs);
        buf[2 * i] = sn;
        buf[2 * i + 1] = cs;
    } else if (i < 64 + kLogTableN) {
        const int j = i - 64;
        const double m = (double)(j + kLogTableLo) * (1.0 / 128.0);
        buf[128 + 2 * j] = log_fast(m);  // (exactly 0 at m = 1)
        buf[128 + 2 * j + 1] = 1.0 / m;
    }
}

// sin and cos of r (|r| <= pi/64 + rounding) around the table point (sa, ca): sin(a + r), cos(a + r)
GSSS_HD void sincos_around(double sa, double ca, double r, double &so, double &co)
{
    const double z = r * r;
    double p = GSSS_FMAK(z, 2.75573192239858906526e-06, -1.98412698412698412698e-04);   // r^9/9!, -r^7/7!
    p = GSSS_FMAK(p, z, 8.33333333333333333333e-03);
    p = GSSS_FMAK(p, z, -1.66666666666666666667e-01);
    const double sr = fma(r * z, p, r);                                                 // sin r
    double q = GSSS_FMAK(z, 2.48015873015873015873e-05, -1.38888888888888888889e-03);   // z^4/8!, -z^3/6!
    q = GSSS_FMAK(q, z, 4.16666666666666666667e-02);
    q = GSSS_FMAK(q, z, -0.5);
    const double cm1 = z * q;                                                           // cos r - 1
    so = sa + fma(ca, sr, sa * cm1);
    co = ca + fma(-sa, sr, ca * cm1);
}

// sin and cos of x, |x| <= 8
GSSS_HD void sincos_tab(double x, const Tables &t, double &so, double &co)
{
    const double k = rint(x * 1.01859163578813017e+01);            // 32 / pi
    double r = fma(-k, 9.81747704208828508854e-02, x);                            // pi / 32, high 33 bits: k * it is exact
    r = fma(-k, 3.79818781656637015582e-12, r);                                   // pi / 32 - high part
    const int j = (int)k & 63;
    sincos_around(t.sc[2 * j], t.sc[2 * j + 1], r, so, co);
}

// sin and cos of 2 pi (w / 2^32) for a 32-bit word w: the nearest table point and an exact signed remainder
GSSS_HD void sincos_word_tab(uint32_t w, const Tables &t, double &so, double &co)
{
    const uint32_t j = (w + (1u << 25)) >> 26;                      // 0 .. 64 (64 wraps to 0)
    const int32_t rem = (int32_t)(w - (j << 26));                   // in [-2^25, 2^25)
    const double r = (double)rem * 1.46291807926715968105e-09;      // 2 pi / 2^32
    const int jj = (int)(j & 63u);
    sincos_around(t.sc[2 * jj], t.sc[2 * jj + 1], r, so, co);
}

// log((w + 1) / 2^32) for a 32-bit word w (the radius uniform of a Box-Muller pair, in (0, 1])
GSSS_HD double log_word_tab(uint32_t w, const Tables &t)
{
    const double v = (double)w + 1.0;                               // 1 .. 2^32, exact
    int e;
    double m = frexp(v, &e);                                        // [0.5, 1)
    if (m < 7.07106781186547524401e-01) {                           // -> [sqrt(1/2), sqrt(2)): log m is small where log v is
        m *= 2.0;
        e -= 1;
    }
    const int j = (int)rint(m * 128.0);                             // nearest table point, 91 .. 181
    const double mj = (double)j * (1.0 / 128.0);
    const double *row = t.lg + 2 * (j - kLogTableLo);
    const double r = (m - mj) * row[1];                             // |r| <= 1/181 (m - mj exact)
    double p = GSSS_FMAK(r, -1.66666666666666666667e-01, 2.0e-01);  // log1p(r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6
    p = GSSS_FMAK(p, r, -0.25);
    p = GSSS_FMAK(p, r, 3.33333333333333333333e-01);
    p = GSSS_FMAK(p, r, -0.5);
    const double l1p = fma(r * r, p, r);
    const double dk = (double)(e - 32);
    return fma(dk, 6.93147180369123816490e-01, fma(dk, 1.90821492927058770002e-10, row[0] + l1p));
}

// ------------------------------------------------------------------------------------------
// Box-Muller pair in SINGLE precision from two 32-bit words of the library stream (radius word: its upper 24 bits ->
// (0, 1]; angle word -> [0, 1) revolutions).  The normals of a step only give the tangent its DIRECTION (mcmc.py:387, sphere.py:29-33): 24 bits
// are plenty, and a pair costs ~50 two-cycle instructions instead of ~60 four-to-five-cycle ones.  Every operation is an
// IEEE single-precision add, multiply, fma, square root or an exact integer / bit operation -- no hardware transcendental --
// so the host (the oracle restates this function, gor_box_muller32) and the device produce the same bits.  Polynomials:
// the classic single-precision minimax sets for log(1 + x) on [sqrt(1/2) - 1, sqrt(2) - 1] and sin / cos on [-pi/4, pi/4]
// (Cephes), ~1e-7.
// ------------------------------------------------------------------------------------------
GSSS_HD uint32_t f32_bits(float x)
{
    uint32_t b;
    __builtin_memcpy(&b, &x, sizeof(b));
    return b;
}
GSSS_HD float bits_f32(uint32_t b)
{
    float x;
    __builtin_memcpy(&x, &b, sizeof(x));
    return x;
}
GSSS_HD void box_muller_f32(uint32_t wr, uint32_t wa, float &z0, float &z1)
{
    // ---- radius: sqrt(-2 ln v), v = ((wr >> 8) + 1) / 2^24 in (0, 1]: the upper 24 bits of the word, exact in single
    // precision (the 2^-24 goes into the exponent count; radii up to 5.8)
    const float vf = (float)((wr >> 8) + 1u);                // 1 .. 2^24
    const uint32_t vb = f32_bits(vf);
    int e = (int)(vb >> 23) - 127 - 24;                      // v = m 2^e, m in [1, 2)
    float m = bits_f32((vb & 0x007FFFFFu) | 0x3F800000u);
    const bool up = m > 1.41421356f;
    m = up ? 0.5f * m : m;                                   // [sqrt(1/2), sqrt(2)]
    e += up ? 1 : 0;
    const float x = m - 1.0f;
    const float z = x * x;
    float p = fmaf(7.0376836292e-2f, x, -1.1514610310e-1f);
    p = fmaf(p, x, 1.1676998740e-1f);
    p = fmaf(p, x, -1.2420140846e-1f);
    p = fmaf(p, x, 1.4249322787e-1f);
    p = fmaf(p, x, -1.6668057665e-1f);
    p = fmaf(p, x, 2.0000714765e-1f);
    p = fmaf(p, x, -2.4999993993e-1f);
    p = fmaf(p, x, 3.3333331174e-1f);
    const float fe = (float)e;
    float ln = fmaf(x * z, p, fe * -2.12194440e-4f);         // ln 2 = 0.693359375 - 2.12194440e-4
    ln = fmaf(-0.5f, z, ln);
    ln = (x + ln) + fe * 0.693359375f;
    const float t = -2.0f * ln;
    const float r = sqrtf(t > 0.0f ? t : 0.0f);
    // ---- angle: 2 pi wa / 2^32 = k pi/4 + a, k = top three bits; reduced to the nearest multiple of pi/2
    const uint32_t k = wa >> 29;
    const float a = (float)(wa & 0x1FFFFFFFu) * 1.46291807926715968e-9f;   // pi/4 / 2^29
    const float y = (k & 1u) ? a - 0.785398163397448f : a;                 // [-pi/4, pi/4]
    const float yy = y * y;
    float sp = fmaf(-1.9515295891e-4f, yy, 8.3321608736e-3f);
    sp = fmaf(sp, yy, -1.6666654611e-1f);
    const float sn = fmaf(y * yy, sp, y);
    float cp = fmaf(2.443315711809948e-5f, yy, -1.388731625493765e-3f);
    cp = fmaf(cp, yy, 4.166664568298827e-2f);
    const float cs = fmaf(yy * yy, cp, fmaf(-0.5f, yy, 1.0f));
    const uint32_t q = ((k + 1u) >> 1) & 3u;                 // quarter turns
    const float c0 = (q & 1u) ? sn : cs, s0 = (q & 1u) ? cs : sn;
    const float c = (q == 1u || q == 2u) ? -c0 : c0;
    const float sg = (q >= 2u) ? -s0 : s0;
    z0 = r * c;
    z1 = r * sg;
}

}  // namespace fm
}  // namespace gsss

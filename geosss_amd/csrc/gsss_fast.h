// gsss_fast.h -- GSSS_MODE_FAST kernels: the throughput path of the shrinkage / rejection
// slice sampler for the lane-per-chain layout.
//
// Three things distinguish it from the step-synchronous kernel in gsss_device.h:
//
//  1. log_prob restricted to the great circle.  On y(theta) = cos(theta) x + sin(theta) u every
//     target of the hot path is a tiny trigonometric form whose coefficients are O(d) dots taken
//     once per step, so a try costs O(K) instead of O(K d):
//        vMF mixture  a_k(theta) = c (mu_k.x) + s (mu_k.u) + logc_k          (distributions.py:156,220)
//        Bingham      q(theta)   = c^2 xAx + c s (xAu + uAx) + s^2 uAu       (distributions.py:86)
//        curve vMF    a_i.y      = c (a_i.x) + s (a_i.u), segment algebra    (spherical_curve.py:10-32)
//  2. The accept test of the mixture is taken in the linear domain: with m = max_k a_k(x),
//        logsumexp_k a_k(theta) > logsumexp_k a_k(x) + log U   <=>   sum_k e^{a_k(theta)-m} > U sum_k e^{a_k(x)-m}
//     so a try needs K exps and no log.
//  3. Two chains per lane with deferred setup.  The number of tries per step varies (mean 5,
//     p99 12), so a step-synchronous wavefront idles half its lanes in the shrink loop.  Here a lane
//     that accepts switches to its second chain at once; the per-step setup (RNG, Box-Muller,
//     projection, coefficients) runs for the whole wavefront only when enough lanes have a chain
//     waiting for it.  Simulated lane utilisation 0.83 vs 0.49 (DESIGN.md "Scheduling").
//
// Elementary functions come from gsss_math.h (bounded-range sincos, Taylor exp).  Results agree
// with the reference to ~1e-14 per step; tests hold them to 1e-10 against the golden chains.
#pragma once
#include "gsss_device.h"
#include "gsss_math.h"

namespace gsss {

constexpr int kChainsPerLane = 2;
constexpr int kFastChainsPerBlock = kBlock * kChainsPerLane;

enum : int32_t { kReady = 0, kPending = 1, kDone = 2 };

// ------------------------------------------------------------------------------------------
// restricted targets (lane layout, D components in registers)
// ------------------------------------------------------------------------------------------
template <int D, int KC>
struct FastVmf {
    static constexpr bool kLinear = true;
    static constexpr int kKind = GSSS_VMF_MIXTURE;
    const double *mu;    // LDS [KC][D]
    const double *logc;  // LDS [KC]
    struct Coef {
        double ax[KC], bu[KC], m;
    };
    __host__ __device__ static size_t lds_doubles() { return (size_t)KC * D + KC; }
    __host__ static bool covers(int d, int k) { return d == D && k == KC; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        for (int i = threadIdx.x; i < KC * D + KC; i += kBlock) lds[i] = tb.blob[i];
        mu = lds;
        logc = lds + KC * D;
    }
    // coefficients of the circle through x along u; returns sum_k e^{a_k(x) - m}
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
        double a0[KC];
        double m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            double ax = 0.0, bu = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double mkj = mu[k * D + j];
                ax = fma(mkj, x[j], ax);
                bu = fma(mkj, u[j], bu);
            }
            cf.ax[k] = ax;
            cf.bu[k] = bu;
            a0[k] = ax + logc[k];
            m = fmax(m, a0[k]);
        }
        cf.m = m;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) s += fm::exp_fast(a0[k] - m);
        return s;
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s) const
    {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) sum += fm::exp_fast(fma(c, cf.ax[k], fma(s, cf.bu[k], logc[k])) - cf.m);
        return sum;
    }
};

template <int D>
struct FastBingham {
    static constexpr bool kLinear = false;
    static constexpr int kKind = GSSS_BINGHAM;
    const double *A;  // LDS [D][D]
    struct Coef {
        double qxx, qxu, quu;
    };
    __host__ __device__ static size_t lds_doubles() { return (size_t)D * D; }
    __host__ static bool covers(int d, int /*k*/) { return d == D; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        for (int i = threadIdx.x; i < D * D; i += kBlock) lds[i] = tb.blob[i];
        A = lds;
    }
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D]) const
    {
        double qxx = 0.0, qxu = 0.0, quu = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double xa = 0.0, ua = 0.0;  // (x A)_j, (u A)_j  (distributions.py:86 contracts rows first)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double aij = A[i * D + j];
                xa = fma(x[i], aij, xa);
                ua = fma(u[i], aij, ua);
            }
            qxx = fma(xa, x[j], qxx);
            qxu = fma(xa, u[j], fma(ua, x[j], qxu));  // xAu + uAx (A need only be symmetric to rounding)
            quu = fma(ua, u[j], quu);
        }
        cf.qxx = qxx;
        cf.qxu = qxu;
        cf.quu = quu;
        return qxx;
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s) const
    {
        return fma(c * c, cf.qxx, fma(c * s, cf.qxu, (s * s) * cf.quu));
    }
};

// ------------------------------------------------------------------------------------------
// per-chain registers
// ------------------------------------------------------------------------------------------
template <int D, class TP, class Dr>
struct FastChain {
    double x[D], u[D];
    typename TP::Coef cf;
    double lo, hi, thr;
    Dr dr;
    int64_t id;       // chain index within this call (addressing); < 0: no chain
    int64_t n_try, n_rej;
    int32_t steps_done, steps_left;
    int32_t until_keep, row;
    int32_t status, err;
    int32_t t;        // proposals made in the current step
};

template <int D, class TP, template <class> class DR>
__global__ void __launch_bounds__(kBlock) fast_kernel(TargetBlock tb, RunBlock a)
{
    using V = LaneVec<D>;
    using Dr = DR<V>;
    using Chain = FastChain<D, TP, Dr>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TP tp;
    tp.stage(lds, tb);
    __syncthreads();

    const int64_t n = a.n_chains;
    const bool shrink = a.sampler == GSSS_SHRINK;
    const int32_t thin = (int32_t)a.thin;

    auto init = [&](Chain &ch, int64_t c) {
        const bool valid = c < n;
        ch.id = valid ? c : -1;
        const int64_t cc = valid ? c : 0;
#pragma unroll
        for (int j = 0; j < D; ++j) ch.x[j] = a.state[(size_t)j * n + cc];
        ch.dr.init(a, cc, D);
        ch.n_try = ch.n_rej = 0;
        ch.steps_done = 0;
        ch.steps_left = valid ? (int32_t)a.n_steps : 0;
        ch.until_keep = thin;
        ch.row = 0;
        ch.err = 0;
        ch.status = ch.steps_left > 0 ? kPending : kDone;
    };

    // everything a step needs before its first try (mcmc.py:387-392)
    auto setup = [&](Chain &ch) {
        ch.dr.begin_step(a.step_offset + (uint64_t)ch.steps_done);
        ch.dr.normals(ch.u, 0);
        {  // u = spherical_projection(z, x), sphere.py:29-33, with reciprocals instead of divisions
            const double rnx = 1.0 / (sqrt(vdot<V>(ch.x, ch.x)) + 1e-100);
            double cz = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) cz = fma(ch.u[j], ch.x[j] * rnx, cz);
#pragma unroll
            for (int j = 0; j < D; ++j) ch.u[j] = fma(-cz, ch.x[j] * rnx, ch.u[j]);
            const double rnw = 1.0 / (sqrt(vdot<V>(ch.u, ch.u)) + 1e-100);
#pragma unroll
            for (int j = 0; j < D; ++j) ch.u[j] *= rnw;
        }
        double u_thr, u_th0;
        ch.dr.step_uniforms(u_thr, u_th0, shrink);
        const double lvl0 = tp.make(ch.cf, ch.x, ch.u);
        bool finite;
        if (TP::kLinear) {
            ch.thr = lvl0 * u_thr;
            finite = lvl0 > 0.0 && lvl0 < INFINITY;
        } else {
            ch.thr = lvl0 + log(u_thr);
            finite = lvl0 > -INFINITY && lvl0 < INFINITY;
        }
        if (shrink) {
            ch.hi = kTwoPi * u_th0;
            ch.lo = ch.hi - kTwoPi;
        } else {
            ch.lo = 0.0;
            ch.hi = kTwoPi;
        }
        ch.t = 0;
        ch.status = kReady;
        if (!finite) {
            ch.err |= GSSS_CHAIN_NONFINITE;
            ch.status = kDone;
        }
    };

    auto attempt = [&](Chain &ch) {
        if (ch.t >= a.max_tries) {
            ch.n_try += ch.t;
            ch.n_rej += ch.t;
            ch.err |= GSSS_CHAIN_MAX_TRIES;
            ch.status = kDone;
            return;
        }
        const double theta = fma(ch.hi - ch.lo, ch.dr.next_try(), ch.lo);  // mcmc.py:395
        ++ch.t;
        double sn, cs;
        fm::sincos_small(theta, sn, cs);
        const double lvl = tp.level(ch.cf, cs, sn);
        const bool exhausted = Dr::kReplay && ch.dr.exhausted;
        if (lvl > ch.thr) {                                                  // mcmc.py:397
#pragma unroll
            for (int j = 0; j < D; ++j) ch.x[j] = fma(sn, ch.u[j], cs * ch.x[j]);  // mcmc.py:396
            ch.n_try += ch.t;
            ch.n_rej += exhausted ? ch.t : ch.t - 1;
            ++ch.steps_done;
            --ch.steps_left;
            if (a.samples != nullptr && --ch.until_keep == 0) {
                ch.until_keep = thin;
#pragma unroll
                for (int j = 0; j < D; ++j) a.samples[((size_t)ch.row * D + j) * n + ch.id] = ch.x[j];
                ++ch.row;
            }
            ch.status = ch.steps_left > 0 ? kPending : kDone;
            if (exhausted) {
                ch.err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
                ch.status = kDone;
            }
            return;
        }
        if (shrink) {                                                        // mcmc.py:400
            if (theta < 0.0)
                ch.lo = theta;
            else
                ch.hi = theta;
        }
        if (exhausted) {
            ch.n_try += ch.t;
            ch.n_rej += ch.t;
            ch.err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            ch.status = kDone;
        }
    };

    auto swap = [](Chain &p, Chain &q) {
        Chain tmp = p;
        p = q;
        q = tmp;
    };

    Chain cur, oth;
    const int64_t base = (int64_t)blockIdx.x * kFastChainsPerBlock + threadIdx.x;
    init(cur, base);
    init(oth, base + kBlock);
    if (cur.status == kPending) setup(cur);
    if (oth.status == kPending) setup(oth);
    if (cur.status != kReady && oth.status == kReady) swap(cur, oth);

    for (;;) {
        if (cur.status == kReady) {
            attempt(cur);
            if (cur.status != kReady && oth.status == kReady) swap(cur, oth);
        }
        const unsigned long long live = __ballot(cur.status != kDone || oth.status != kDone);
        if (live == 0ull) break;
        const unsigned long long waiting = __ballot(cur.status == kPending);  // nothing to try until set up
        const unsigned long long pend = __ballot(cur.status == kPending || oth.status == kPending);
        const int n_live = __popcll(live);
        if (pend != 0ull && (2 * __popcll(waiting) >= n_live || 8 * __popcll(pend) >= 7 * n_live)) {
            if (cur.status == kPending && oth.status != kPending) swap(cur, oth);
            if (oth.status == kPending) setup(oth);
            if (cur.status != kReady && oth.status == kReady) swap(cur, oth);
        }
    }

    auto flush = [&](const Chain &ch) {
        if (ch.id < 0) return;
#pragma unroll
        for (int j = 0; j < D; ++j) a.state[(size_t)j * n + ch.id] = ch.x[j];
        if (a.n_reject) a.n_reject[ch.id] += ch.n_rej;
        if (a.n_tries) a.n_tries[ch.id] += ch.n_try;
        if (a.err && ch.err) a.err[ch.id] |= ch.err;
    };
    flush(cur);
    flush(oth);
}

// host side: launch one instantiation
void set_error(const char *fmt, ...);

template <int D, class TP, template <class> class DR>
int do_fast_run(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    const size_t lds = TP::lds_doubles() * sizeof(double);
    auto kern = fast_kernel<D, TP, DR>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int64_t grid = (rb.n_chains + kFastChainsPerBlock - 1) / kFastChainsPerBlock;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("fast kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

template <int D, class TP>
int do_fast(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    return replay ? do_fast_run<D, TP, ReplayDraws>(tb, rb, st) : do_fast_run<D, TP, PhiloxDraws>(tb, rb, st);
}

// per-target entry points (one translation unit each); GSSS_E_UNSUPPORTED when no instantiation
// covers (d, k).  `probe` = only answer whether a kernel exists.
int launch_fast_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, bool probe, hipStream_t st);
int launch_fast_bingham(const TargetBlock &tb, const RunBlock &rb, bool replay, bool probe, hipStream_t st);

}  // namespace gsss

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
//     keeps one chain in registers and a second one parked in LDS; when its chain accepts it trades
//     the two (ds_wrxchg) and keeps trying; the per-step setup (RNG, Box-Muller, projection,
//     coefficients) runs for the whole wavefront only when enough lanes have a chain waiting for it.
//     Measured lane utilisation 0.72 vs 0.49.
//
// The second half of the file holds the cooperative variant for large d (L lanes per chain).
//
// Elementary functions come from gsss_math.h (bounded-range sincos, Taylor exp).  Results agree
// with the reference to ~1e-14 per step; tests hold them to 1e-10 against the golden chains.
#pragma once
#include <stdio.h>

#include "gsss_device.h"
#include "gsss_math.h"

namespace gsss {

enum : int32_t { kReady = 0, kPending = 1, kDone = 2 };
constexpr double kLogZero = -1.0e5;  // stands in for log(0) where the arithmetic must stay finite

// 1 / (sqrt(s) + 1e-100) of sphere.py:14 as one reciprocal square root; the 1e-100 only matters for
// the zero vector, whose projection the reference maps to zero
__device__ __forceinline__ double inv_norm(double s) { return s > 1e-200 ? rsqrt(s) : 0.0; }

// ------------------------------------------------------------------------------------------
// restricted targets (lane layout, D components in registers)
//   make(cf, x, u, lvl, fresh)  coefficients of the circle through x along u; returns the level
//                               of x (theta = 0), formed from x as the reference forms its threshold.
//                               (`lvl`, `fresh`: the level carried over from the accepting try and the
//                               first-step flag of round 1's carry scheme -- no target uses them any more.)
//   level(cf, c, s)             level of y(theta); accept iff level > threshold
//   kLinear                     level is a density (threshold = level(x) * U) rather than a
//                               log-density (threshold = level(x) + log U)
// ------------------------------------------------------------------------------------------
template <int D, int KC>
struct FastVmf {
    static constexpr bool kLinear = true;
    const double *mu;    // LDS [KC][D]
    const double *logc;  // LDS [KC]
    struct Coef {
        double ax[KC], au[KC], m;
        template <class F>
        __device__ __forceinline__ void each(F &&f)
        {
#pragma unroll
            for (int k = 0; k < KC; ++k) f(ax[k]);
#pragma unroll
            for (int k = 0; k < KC; ++k) f(au[k]);
            f(m);
        }
    };
    static constexpr int kCoefWords = 2 * KC + 1;
    __host__ __device__ static size_t lds_doubles() { return (size_t)KC * D + KC; }
    int K;  // components of the target, K <= KC: the kernels are built for KC, the surplus ones are padded with
            // mu = 0, logc = log(0), so that they add exactly +0.0 to every sum (a mixture of any K <= KC runs the KC kernel)
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        K = tb.k;
        for (int i = threadIdx.x; i < KC * D; i += kBlock) lds[i] = i < K * D ? tb.blob[i] : 0.0;
        // a zero-weight component has logc = -inf (log w_k, distributions.py:220); exp_bounded wants finite
        // arguments, and e^{-1e5} is as much a zero as e^{-inf}
        for (int i = threadIdx.x; i < KC; i += kBlock) lds[KC * D + i] = i < K ? fmax(tb.blob[K * D + i], kLogZero) : kLogZero;
        mu = lds;
        logc = lds + KC * D;
    }
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D], double lvl,
                                           bool fresh) const
    {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double mkj = mu[k * D + j];
                ax = fma(mkj, x[j], ax);
                au = fma(mkj, u[j], au);
            }
            cf.ax[k] = ax;
            cf.au[k] = au;
        }
        // level of x itself, computed from x as the reference does for its threshold (mcmc.py:389); the offset m
        // only keeps the exponentials in range
        (void)fresh;
        double m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KC; ++k) m = fmax(m, cf.ax[k] + logc[k]);
        cf.m = m;
        lvl = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) lvl += fm::exp_fast((cf.ax[k] + logc[k]) - m);
        return lvl;
    }
    // Many components: only the largest term is exponentiated in double; the others are bounded from
    // above with the hardware's single-precision 2^x.  lower = e^{a_max-m} <= S <= lower (1 + r_ub), and
    // the accept test S > thr is decided by the bounds whenever thr is outside (lower, upper] -- the same
    // decision the full sum gives.  Otherwise (thr between the bounds: the other components matter, rare
    // for separated modes) the full double-precision sum is formed.  Returns a value that compares with
    // `thr` exactly like the full sum.
    static constexpr bool kScreened = KC >= 5;
    __device__ __forceinline__ double level_full(const double (&a)[KC]) const
    {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) sum += fm::exp_bounded(a[k]);
        return sum;
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s, double thr) const
    {
        double a[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) a[k] = fma(c, cf.ax[k], fma(s, cf.au[k], logc[k])) - cf.m;
        if (!kScreened) return level_full(a);
        double amax = a[0];
#pragma unroll
        for (int k = 1; k < KC; ++k) amax = fmax(amax, a[k]);
        const double lower = fm::exp_bounded(amax);
        float r = -1.0f;  // sum_k 2^{(a_k - amax) log2 e} - 1 : the terms other than the largest
#pragma unroll
        for (int k = 0; k < KC; ++k) r += __builtin_amdgcn_exp2f((float)(a[k] - amax) * 1.44269504f);
        const double r_ub = (double)fmaxf(r, 0.0f) * 1.0001 + 2e-6;  // covers the float exp, cast and sum errors
        if (lower > thr) return lower;                                // S >= lower > thr: accept
        const double upper = lower * (1.0 + r_ub);
        if (upper <= thr) return upper;                               // S <= upper <= thr: reject
        return level_full(a);
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s) const
    {
        double a[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) a[k] = fma(c, cf.ax[k], fma(s, cf.au[k], logc[k])) - cf.m;
        return level_full(a);
    }
};

template <int D>
struct FastBingham {
    static constexpr bool kLinear = false;
    const double *A;  // LDS [D][D]
    const double *b;  // LDS [D]: BinghamFisher linear term (zeros for a plain Bingham)
    struct Coef {
        double qxx, qxu, quu, bx, bu;
        template <class F>
        __device__ __forceinline__ void each(F &&f)
        {
            f(qxx);
            f(qxu);
            f(quu);
            f(bx);
            f(bu);
        }
    };
    static constexpr int kCoefWords = 5;
    __host__ __device__ static size_t lds_doubles() { return (size_t)D * D + D; }
    bool diagonal;  // A is diagonal (the eigenbasis targets of scripts/bingham.py:131): O(d) coefficients
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        for (int i = threadIdx.x; i < D * D + D; i += kBlock) lds[i] = tb.blob[i];
        A = lds;
        b = lds + D * D;
        diagonal = (tb.k & 1) != 0;
    }
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D], double /*lvl*/,
                                           bool /*fresh*/) const
    {
        double qxx = 0.0, qxu = 0.0, quu = 0.0, bx = 0.0, bu = 0.0;
        if (diagonal) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double ajj = A[j * D + j];
                const double xa = x[j] * ajj, ua = u[j] * ajj;
                qxx = fma(xa, x[j], qxx);
                qxu = fma(xa, u[j], fma(ua, x[j], qxu));
                quu = fma(ua, u[j], quu);
                bx = fma(b[j], x[j], bx);
                bu = fma(b[j], u[j], bu);
            }
            cf.qxx = qxx;
            cf.qxu = qxu;
            cf.quu = quu;
            cf.bx = bx;
            cf.bu = bu;
            return qxx + bx;
        }
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
            bx = fma(b[j], x[j], bx);
            bu = fma(b[j], u[j], bu);
            // d > 10: one column at a time -- left alone the scheduler hoists the LDS reads of many columns (d of them each) and
            // the d = 12 / 14 / 15 kernels spill 50-90 bytes per lane at two wavefronts per SIMD
            if constexpr (D > 10) __builtin_amdgcn_sched_barrier(0);
        }
        cf.qxx = qxx;
        cf.qxu = qxu;
        cf.quu = quu;
        cf.bx = bx;
        cf.bu = bu;
        return qxx + bx;
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s, double /*thr*/) const
    {
        return level(cf, c, s);
    }
    // y^T A y + y.b on the circle (distributions.py:86, :113-114)
    __device__ __forceinline__ double level(const Coef &cf, double c, double s) const
    {
        return fma(c * c, cf.qxx, fma(c * s, cf.qxu, (s * s) * cf.quu)) + fma(c, cf.bx, s * cf.bu);
    }
};

template <int D, int NK>
struct FastCurve {
    static constexpr bool kLinear = false;
    const double *knots;  // LDS [NK][D]
    const double *seg;    // LDS [NK-1][4]: cos(theta_s), sin(theta_s), 1/(sin(theta_s)+1e-10), unused
    double kappa;
    struct Coef {
        double ax[NK], au[NK];  // a_i . x, a_i . u
        template <class F>
        __device__ __forceinline__ void each(F &&f)
        {
#pragma unroll
            for (int i = 0; i < NK; ++i) f(ax[i]);
#pragma unroll
            for (int i = 0; i < NK; ++i) f(au[i]);
        }
    };
    static constexpr int kCoefWords = 2 * NK;
    __host__ __device__ static size_t lds_doubles() { return (size_t)NK * D + 4 * (size_t)(NK - 1); }
    int nseg;  // segments of the target's curve: k - 1 <= NK - 1 (a curve of fewer knots runs the NK kernel: the surplus
               // knot rows are zeros and their segments never take part in the maximum)
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        kappa = tb.kappa;
        nseg = tb.k - 1;
        for (int i = threadIdx.x; i < NK * D; i += kBlock) lds[i] = i < tb.k * D ? tb.blob[i] : 0.0;
        double *sg = lds + NK * D;
        for (int i = threadIdx.x; i < NK - 1; i += kBlock) {  // blob: theta, cos, sin, sin + 1e-10
            const bool real = i < tb.k - 1;
            sg[4 * i + 0] = real ? tb.blob[(size_t)tb.k * D + 4 * i + 1] : 1.0;
            sg[4 * i + 1] = real ? tb.blob[(size_t)tb.k * D + 4 * i + 2] : 0.0;
            sg[4 * i + 2] = real ? 1.0 / tb.blob[(size_t)tb.k * D + 4 * i + 3] : 0.0;
            sg[4 * i + 3] = 0.0;
        }
        knots = lds;
        seg = sg;
    }
    // kappa * (y . nearest point of the curve), spherical_curve.py:10-32 without trigonometry:
    // with A = (a.y) sin(th), B = b.y - (a.y) cos(th), t = atan2(B, A) clipped to [0, th] has
    //   t = 0 when B <= 0,  t = th when A < cos(th) hypot(A, B),  else sin t = B/h, cos t = A/h,
    // and y . near = (sin(th - t) (a.y) + sin(t) (b.y)) / (sin(th) + 1e-10).  The nearest segment is
    // the first one of minimal acos(clip(y.near)) = the first one of maximal clipped y.near.
    __device__ __forceinline__ double level(const Coef &cf, double c, double s, double /*thr*/) const
    {
        return level(cf, c, s);
    }
    // one segment (a, b) of the curve given a.y and b.y: y . nearest (xy) and its clipped value (xc; -inf for a padding segment)
    __device__ __forceinline__ void segment_value(int g, double ay, double by, double &xc, double &xy) const
    {
        const double ct = seg[4 * g], st = seg[4 * g + 1], rden = seg[4 * g + 2];
        const double A = ay * st;
        const double B = fma(-ay, ct, by);
        const double h2 = fma(A, A, B * B);
        const double rh = h2 > 0.0 ? rsqrt(h2) : 0.0;
        const double inner = fma(fma(st, A, -ct * B), ay, B * by) * rh;  // sin(th-t) a.y + sin(t) b.y
        const bool at_a = B < 0.0 || (B == 0.0 && A >= 0.0);
        const bool at_b = A * rh < ct;
        const double num = at_a ? st * ay : (at_b ? st * by : inner);
        xy = num * rden;
        xc = g < nseg ? fmin(fmax(xy, -1.0), 1.0) : -INFINITY;
    }
    // keeps the first maximum of the clipped y . nearest
    __device__ __forceinline__ void segment(int g, double ay, double by, double &best, double &best_dot) const
    {
        double xc, xy;
        segment_value(g, ay, by, xc, xy);
        if (xc > best) {
            best = xc;
            best_dot = xy;
        }
    }
    __device__ __forceinline__ double level(const Coef &cf, double c, double s) const
    {
        double best = -INFINITY, best_dot = 0.0;
        double ay = fma(c, cf.ax[0], s * cf.au[0]);
#pragma unroll
        for (int g = 0; g + 1 < NK; ++g) {
            const double by = fma(c, cf.ax[g + 1], s * cf.au[g + 1]);
            segment(g, ay, by, best, best_dot);
            ay = by;
        }
        return kappa * best_dot;
    }
    // level() one segment after the other (the rare double-precision decisions of the screened kernels: left alone the
    // scheduler interleaves all segments and holds their constants and intermediates in ~100 registers)
    __device__ __forceinline__ double level_serial(const Coef &cf, double c, double s) const
    {
        double best = -INFINITY, best_dot = 0.0;
        double ay = fma(c, cf.ax[0], s * cf.au[0]);
#pragma unroll
        for (int g = 0; g + 1 < NK; ++g) {
            const double by = fma(c, cf.ax[g + 1], s * cf.au[g + 1]);
            segment(g, ay, by, best, best_dot);
            ay = by;
            __builtin_amdgcn_sched_barrier(0);
        }
        return kappa * best_dot;
    }
    __device__ __forceinline__ double make(Coef &cf, const double (&x)[D], const double (&u)[D], double /*lvl*/,
                                           bool /*fresh*/) const
    {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            double ax = 0.0, au = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double kij = knots[i * D + j];
                ax = fma(kij, x[j], ax);
                au = fma(kij, u[j], au);
            }
            cf.ax[i] = ax;
            cf.au[i] = au;
        }
        return level(cf, 1.0, 0.0);
    }
};

// ------------------------------------------------------------------------------------------
// per-chain state: only what must survive between tries.  RNG counters are rebuilt from
// (id, steps_done, t) when a draw is needed.  A lane keeps its current chain in registers and
// its other chain in an LDS slot; the two trade places with one ds_wrxchg per 64-bit word.
// ------------------------------------------------------------------------------------------
template <int D, class TP>
struct FastChain {
    double x[D], u[D];
    typename TP::Coef cf;
    double lo, hi, thr, lvl;
    uint32_t n_try;      // proposals made in this launch (saturating)
    int32_t steps_done, row;
    int32_t t;           // proposals made in the current step (< 2^26)
    int32_t status, err;
    int32_t cursor;      // replay: draws consumed
    // 64-bit words a parked chain occupies in LDS: the doubles, (steps_done,row), (n_try, t|status|err)
    // [+ (cursor,0) for replay].  19 words for d = 3, K = 3 -> 4 workgroups per CU.
    static constexpr int kWordsNoReplay = 2 * D + TP::kCoefWords + 4 + 2;
    static constexpr int kWords = kWordsNoReplay + 1;
};

__device__ __forceinline__ void lds_trade(double &v, unsigned long long *slot)
{
    const unsigned long long o = __hip_atomic_exchange(slot, (unsigned long long)__double_as_longlong(v),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    v = __longlong_as_double((long long)o);
}
__device__ __forceinline__ void lds_trade(int64_t &v, unsigned long long *slot)
{
    v = (int64_t)__hip_atomic_exchange(slot, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_trade(int32_t &a, int32_t &b, unsigned long long *slot)
{
    const unsigned long long mine = (unsigned long long)(uint32_t)a | ((unsigned long long)(uint32_t)b << 32);
    const unsigned long long o = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    a = (int32_t)(uint32_t)o;
    b = (int32_t)(uint32_t)(o >> 32);
}

// a second chain per lane is parked in LDS only while that leaves room for >= 2 workgroups per CU
template <int D, class TP>
__host__ __device__ constexpr bool fast_parks()
{
    return (size_t)FastChain<D, TP>::kWords * kBlock * sizeof(double) <= 72 * 1024;
}
template <int D, class TP>
__host__ __device__ constexpr int fast_chains_per_block()
{
    return fast_parks<D, TP>() ? 2 * kBlock : kBlock;
}
template <int D, class TP, bool REPLAY>
__host__ __device__ constexpr size_t fast_lds_doubles()
{
    return TP::lds_doubles() + kTabLds +
           (fast_parks<D, TP>() ? (size_t)(REPLAY ? FastChain<D, TP>::kWords : FastChain<D, TP>::kWordsNoReplay) * kBlock
                                : 0);
}

// NUMPY: the draws come from numpy's own PCG64 / ziggurat stream (rng_state, NumpyDraws) instead of the replay buffer, at
// the replay path's consumption points -- the reference's own order, a uniform per try that is made and none for one that is
// not.  A generator per chain: one chain per lane (nothing is parked), the ziggurat tables where the parked chains would be.
template <int D, class TP, bool REPLAY, bool STATS = false, bool NUMPY = false>
__global__ void __launch_bounds__(kBlock) fast_kernel(TargetBlock tb, RunBlock a)
{
    static_assert(!NUMPY || REPLAY, "numpy's stream is a sequential source: it is read where the replay buffer is");
    using V = LaneVec<D>;
    using Chain = FastChain<D, TP>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TP tp;
    tp.stage(lds, tb);
    const fm::Tables tab = stage_tables(lds + TP::lds_doubles());
    // word w of this lane's parked chain lives at park[w * kBlock]: conflict-free across lanes
    unsigned long long *park = reinterpret_cast<unsigned long long *>(lds + TP::lds_doubles() + kTabLds) + threadIdx.x;
    NumpyDraws<V> nd;
    if constexpr (NUMPY) nd.stage(lds + TP::lds_doubles() + kTabLds);
    __syncthreads();

    const int32_t n = (int32_t)a.n_chains;
    const int32_t n_steps = (int32_t)a.n_steps;
    const bool shrink = a.sampler == GSSS_SHRINK;
    const int32_t thin = (int32_t)a.thin;
    const int32_t max_tries = a.max_tries < (1 << 26) ? a.max_tries : (1 << 26) - 1;  // t shares a word with the flags
    constexpr uint32_t kTryBase = 1u + (uint32_t)((D + 3) / 4);
    constexpr bool kPark = fast_parks<D, TP>() && !NUMPY;
    constexpr int kPerBlock = kPark ? 2 * kBlock : kBlock;
    // packed: lane l of block b owns chains b*P + l and (parking targets) b*P + 256 + l.
    // spread (small ensembles): one chain per wavefront, owned by its lane 0; nothing is parked.
    const bool spread = a.spread != 0;
    const int32_t id0 = spread ? ((threadIdx.x % 64 == 0) ? (int32_t)blockIdx.x * (kBlock / 64) + (int32_t)threadIdx.x / 64 : n)
                               : (int32_t)blockIdx.x * kPerBlock + (int32_t)threadIdx.x;
    const int32_t id1 = spread ? n : id0 + kBlock;

    Chain cur;
    int32_t slot = 0;               // which of the lane's two chains `cur` is
    int32_t parked_status = kDone;  // status of the chain in LDS

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
        if (sum < cur.n_try) cur.err |= GSSS_CHAIN_COUNTER_SATURATED;  // > 2^32-1 proposals in one launch
        cur.n_try = sum < cur.n_try ? 0xFFFFFFFFu : sum;
    };
    auto init = [&]() {
        const int32_t c = chain_id();
        const bool valid = c < n;
        const int32_t cc = valid ? c : 0;
#pragma unroll
        for (int j = 0; j < D; ++j) cur.x[j] = a.state[(size_t)j * n + cc];
        cur.n_try = 0u;
        cur.steps_done = 0;
        cur.row = 0;
        cur.err = 0;
        cur.cursor = 0;
        cur.lvl = 0.0;
        cur.t = 0;
        cur.status = (valid && n_steps > 0) ? kPending : kDone;
        if constexpr (NUMPY) nd.init(a, cc, D);
    };

    // everything a step needs before its first try (mcmc.py:387-392)
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
        {  // u = spherical_projection(z, x), sphere.py:29-33, with reciprocals instead of divisions
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
        const double lvl0 = tp.make(cur.cf, cur.x, cur.u, cur.lvl, cur.steps_done == 0);
        bool finite;
        if (TP::kLinear) {
            cur.thr = lvl0 * u_thr;
            finite = lvl0 > 0.0 && lvl0 < INFINITY;
        } else {
            cur.thr = lvl0 + fm::log_fast(u_thr);
            finite = lvl0 > -INFINITY && lvl0 < INFINITY;
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

    // up to two proposals (one Philox block feeds both: the stream hands tries out in pairs)
    auto attempt = [&]() {
        if (cur.t >= max_tries) {
            count_tries();
            cur.err |= GSSS_CHAIN_MAX_TRIES;
            cur.status = kDone;
            return;
        }
        double u_pair[2];
        if (!REPLAY) {  // philox-v3: a try's uniform is one 32-bit word, block j feeds tries 4j .. 4j + 3; this attempt makes tries t, t + 1 (t even)
            uint32_t w[4];
            philox().words(kTryBase + (uint32_t)(cur.t >> 2), w);
            u_pair[0] = try_uniform((cur.t & 2) ? w[2] : w[0]);
            u_pair[1] = try_uniform((cur.t & 2) ? w[3] : w[1]);
        }
        bool accepted = false;
        double sn = 0.0, cs = 1.0, lvl = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!accepted && (h == 0 || cur.t < max_tries) && !(REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED))) {
                const double uu = REPLAY ? replay_take() : u_pair[h];
                const double theta = fma(cur.hi - cur.lo, uu, cur.lo);  // mcmc.py:395
                ++cur.t;
                fm::sincos_tab(theta, tab, sn, cs);
                lvl = tp.level(cur.cf, cs, sn, cur.thr);
                accepted = lvl > cur.thr;                               // mcmc.py:397
                if (!accepted && shrink) {                              // mcmc.py:400
                    if (theta < 0.0)
                        cur.lo = theta;
                    else
                        cur.hi = theta;
                }
            }
        }
        const bool exhausted = REPLAY && (cur.err & GSSS_CHAIN_REPLAY_EXHAUSTED);
        if (accepted) {
#pragma unroll
            for (int j = 0; j < D; ++j) cur.x[j] = fma(sn, cur.u[j], cs * cur.x[j]);  // mcmc.py:396
            cur.lvl = lvl;
            count_tries();
            ++cur.steps_done;
            if ((a.samples != nullptr || STATS) && cur.steps_done == (cur.row + 1) * thin) {
                if (a.samples != nullptr) {
#pragma unroll
                    for (int j = 0; j < D; ++j) a.samples[sample_index(a, cur.row, j, D, chain_id())] = cur.x[j];
                }
                if constexpr (STATS) stats_update<D>(a, chain_id(), cur.x);
                ++cur.row;
            }
            cur.status = (cur.steps_done < n_steps && !exhausted) ? kPending : kDone;
        } else if (exhausted) {
            count_tries();
            cur.status = kDone;
        }
    };

    // current chain <-> parked chain
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
        cur.cf.each(word);
        word(cur.lo);
        word(cur.hi);
        word(cur.thr);
        word(cur.lvl);
        lds_trade(cur.steps_done, cur.row, p);
        p += kBlock;
        if (REPLAY) {
            int32_t zero = 0;
            lds_trade(cur.cursor, zero, p);
            p += kBlock;
        }
        // (n_try, t | status << 26 | err << 28); the status also stays in a register for the wave-level votes
        const int32_t st = cur.status;
        int32_t nt = (int32_t)cur.n_try, packed = cur.t | (cur.status << 26) | (int32_t)((uint32_t)cur.err << 28);
        lds_trade(nt, packed, p);
        cur.n_try = (uint32_t)nt;
        cur.t = packed & 0x3FFFFFF;
        cur.status = (packed >> 26) & 3;
        cur.err = (int32_t)((uint32_t)packed >> 28);
        parked_status = st;
        slot ^= 1;
    };

    auto flush = [&]() {
        const int32_t c = chain_id();
        if (c >= n) return;
#pragma unroll
        for (int j = 0; j < D; ++j) a.state[(size_t)j * n + c] = cur.x[j];
        // every accepted step has exactly one non-rejected proposal (counters of a chain that
        // stopped with an error bit are not specified beyond that bit)
        if (a.n_reject) a.n_reject[c] += (int64_t)cur.n_try - cur.steps_done;
        if (a.n_tries) a.n_tries[c] += (int64_t)cur.n_try;
        if (a.err && cur.err) a.err[c] |= cur.err;
        if constexpr (NUMPY) nd.finish(a, c, true);
    };

    // chain of slot 1 is initialised, set up and parked; then the chain of slot 0
    if (kPark) {
        slot = 1;
        init();
        if (cur.status == kPending) setup();
        // plain stores: the slot holds nothing yet
        unsigned long long *p = park;
        auto put = [&](double v) {
            *p = (unsigned long long)__double_as_longlong(v);
            p += kBlock;
        };
        auto put2 = [&](int32_t lo32, int32_t hi32) {
            *p = (unsigned long long)(uint32_t)lo32 | ((unsigned long long)(uint32_t)hi32 << 32);
            p += kBlock;
        };
#pragma unroll
        for (int j = 0; j < D; ++j) put(cur.x[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) put(cur.u[j]);
        cur.cf.each([&](double &v) { put(v); });
        put(cur.lo);
        put(cur.hi);
        put(cur.thr);
        put(cur.lvl);
        put2(cur.steps_done, cur.row);
        if (REPLAY) put2(cur.cursor, 0);
        put2((int32_t)cur.n_try, cur.t | (cur.status << 26) | (int32_t)((uint32_t)cur.err << 28));
        parked_status = cur.status;
    }
    slot = 0;
    init();
    if (cur.status == kPending) setup();
    if (cur.status != kReady && parked_status == kReady) trade();

    for (;;) {
        if (cur.status == kReady) {
            attempt();
            if (cur.status != kReady && parked_status == kReady) trade();
        }
        const unsigned long long live = __ballot(cur.status != kDone || parked_status != kDone);
        if (live == 0ull) break;
        const unsigned long long waiting = __ballot(cur.status == kPending);  // nothing to try until set up
        const unsigned long long pend = __ballot(cur.status == kPending || parked_status == kPending);
        const int n_live = __popcll(live);
        if (pend != 0ull && (2 * __popcll(waiting) >= n_live || 8 * __popcll(pend) >= 7 * n_live)) {
            if (cur.status != kPending && parked_status == kPending) trade();  // bring the waiting chain in
            if (cur.status == kPending) setup();
        }
    }
    flush();
    if (kPark) {
        trade();
        flush();
    }
}

// host side: launch one instantiation
void set_error(const char *fmt, ...);

template <int D, class TP, bool REPLAY>
int do_fast_run(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    const size_t lds = fast_lds_doubles<D, TP, REPLAY>() * sizeof(double);
    auto kern = fast_kernel<D, TP, REPLAY, false>;
    if constexpr (!REPLAY) {  // running statistics: a build of its own (the plain kernel carries none of it)
        if (rb.stats != nullptr) kern = fast_kernel<D, TP, false, true>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int per_block = rb.spread ? kBlock / 64 : fast_chains_per_block<D, TP>();
    const int64_t grid = (rb.n_chains + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("fast kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

// numpy's stream, one lane per chain (fast_kernel<..., NUMPY>)
template <int D, class TP>
int do_fast_numpy(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    const size_t lds = (TP::lds_doubles() + kTabLds + NumpyDraws<LaneVec<D>>::kLdsDoubles) * sizeof(double);
    auto kern = rb.stats != nullptr ? fast_kernel<D, TP, true, true, true> : fast_kernel<D, TP, true, false, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int per_block = rb.spread ? kBlock / 64 : kBlock;
    const int64_t grid = (rb.n_chains + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("fast kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

template <int D, class TP>
int do_wave(const TargetBlock &tb, const RunBlock &rb, hipStream_t st);

template <int D, class TP>
int do_fast(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    if (replay) return do_fast_run<D, TP, true>(tb, rb, st);
    if constexpr (D <= 16) {
        if (rb.spread) return do_wave<D, TP>(tb, rb, st);  // small ensemble: one wavefront per chain
    }
    if (rb.rng_state != nullptr) return do_fast_numpy<D, TP>(tb, rb, st);  // a generator per chain: one lane per chain
    return do_fast_run<D, TP, false>(tb, rb, st);
}

// per-target entry points (one translation unit each); GSSS_E_UNSUPPORTED when no instantiation
// covers (d, k).  `probe` != nullptr: launch nothing, only answer whether a kernel exists and name it.
struct FastProbe {
    char name[160];  // the instantiation that would run, e.g. "fast_kernel<3, FastVmf<3, 3>>"
    bool lane;       // lane-per-chain layout (small ensembles then run wave_kernel of the same shape)
};
int launch_fast_vmf(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st);
int launch_fast_bingham(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st);
int launch_fast_curve(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, hipStream_t st);
int launch_curvespec(const TargetBlock &tb, const RunBlock &rb, bool replay, FastProbe *probe, bool lane, hipStream_t st);
#define GSSS_PROBE(LANE, ...)                                         \
    do {                                                              \
        snprintf(probe->name, sizeof(probe->name), __VA_ARGS__);      \
        probe->lane = LANE;                                           \
        return GSSS_OK;                                               \
    } while (0)

}  // namespace gsss

// ==========================================================================================
// Cooperative fast kernels: large d.  L lanes share one chain (CoopVec), the O(d) work of a step
// (normals, projection, dots with the target's parameter vectors, state update) is spread over the
// lanes, the O(1)-per-try restricted form is evaluated redundantly by every lane of the group (all
// lanes hold identical bits after the xor-butterfly, so the group never diverges internally).
// The coefficients a_i.x follow the exact recurrence a_i.x' = c a_i.x + s a_i.u between refreshes.
// ==========================================================================================
namespace gsss {

constexpr int kCoefRefresh = 64;  // recompute a_i.x from x every this many steps (bounds rounding drift)

// LDS doubles of a cooperative target's parameters (Bingham's depend on d)
template <class TP>
__host__ __device__ inline size_t coop_param_doubles(int d)
{
    if constexpr (TP::kQuadratic)
        return TP::lds_doubles(d);
    else
        return TP::lds_doubles();
}

template <class V, int NK>
struct CoopCurve {
    using Scalar = FastCurve<1, NK>;  // only its level() and Coef are used
    static constexpr bool kLinear = false;
    static constexpr int kVectors = NK;
    static constexpr bool kQuadratic = false;
    static constexpr int kScratchPerGroup = 0;
    Scalar sc;
    const double *rows;  // LDS [NK][DPAD]
    __host__ __device__ static size_t lds_doubles() { return (size_t)NK * V::DPAD + 4 * (size_t)(NK - 1); }
    int nseg;  // k - 1 segments, k <= NK knots (surplus rows zero, surplus segments out of the maximum)
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        nseg = tb.k - 1;
        lds_fill(lds, tb.k, V::DPAD, tb.blob, tb.d);
        for (int i = threadIdx.x + tb.k * V::DPAD; i < NK * V::DPAD; i += kBlock) lds[i] = 0.0;
        double *sg = lds + (size_t)NK * V::DPAD;
        for (int i = threadIdx.x; i < NK - 1; i += kBlock) {
            const bool real = i < tb.k - 1;
            sg[4 * i + 0] = real ? tb.blob[(size_t)tb.k * tb.d + 4 * i + 1] : 1.0;
            sg[4 * i + 1] = real ? tb.blob[(size_t)tb.k * tb.d + 4 * i + 2] : 0.0;
            sg[4 * i + 2] = real ? 1.0 / tb.blob[(size_t)tb.k * tb.d + 4 * i + 3] : 0.0;
            sg[4 * i + 3] = 0.0;
        }
        rows = lds;
        sc.knots = lds;
        sc.seg = sg;
        sc.kappa = tb.kappa;
        sc.nseg = tb.k - 1;
    }
    // Lane g of a group evaluates segment g (lanes >= NK-1 sit out); an xor-butterfly then picks the
    // first segment of maximal clipped y.near -- the same choice as the sequential scan.
    static constexpr bool kDistributed = V::L >= 16;
    static_assert(NK - 1 <= 16, "segments must fit one row of 16 lanes");
    struct Mine {  // what lane g needs of the coefficients: those of knots g and g+1 (its segment)
        double ax0, au0, ax1, au1, ct, st, rden;
    };
    __device__ __forceinline__ Mine mine_init(int g) const
    {
        const int gs = g < NK - 1 ? g : 0;
        return Mine{0.0, 0.0, 0.0, 0.0, sc.seg[4 * gs], sc.seg[4 * gs + 1], sc.seg[4 * gs + 2]};
    }
    __device__ __forceinline__ void take_au(Mine &m, int g, int r, double v) const
    {
        if (r == g) m.au0 = v;
        if (r == g + 1) m.au1 = v;
    }
    __device__ __forceinline__ void take_ax(Mine &m, int g, int r, double v) const
    {
        if (r == g) m.ax0 = v;
        if (r == g + 1) m.ax1 = v;
    }
    __device__ __forceinline__ void advance(Mine &m, double c, double s) const  // a.x' = c a.x + s a.u
    {
        m.ax0 = fma(c, m.ax0, s * m.au0);
        m.ax1 = fma(c, m.ax1, s * m.au1);
    }
    __device__ __forceinline__ double level_distributed(const Mine &m, int g, double c, double s) const
    {
        const double ay = fma(c, m.ax0, s * m.au0), by = fma(c, m.ax1, s * m.au1);
        const double A = ay * m.st;
        const double B = fma(-ay, m.ct, by);
        const double h2 = fma(A, A, B * B);
        const double rh = h2 > 0.0 ? rsqrt(h2) : 0.0;
        const double inner = fma(fma(m.st, A, -m.ct * B), ay, B * by) * rh;
        const bool at_a = B < 0.0 || (B == 0.0 && A >= 0.0);
        const bool at_b = A * rh < m.ct;
        const double num = at_a ? m.st * ay : (at_b ? m.st * by : inner);
        double xy = num * m.rden;
        double xc = g < nseg ? fmin(fmax(xy, -1.0), 1.0) : -INFINITY;
        int idx = g;
        // (xc, idx) is totally ordered (larger xc first, then smaller idx): the winner of a row of 16 lanes
        // does not depend on the order of the pairwise steps; the segments live in lanes 0 .. NK-2 of row 0
        auto step = [&](double oxc, double oxy, int oidx) {
            const bool take = oxc > xc || (oxc == xc && oidx < idx);
            xc = take ? oxc : xc;
            xy = take ? oxy : xy;
            idx = take ? oidx : idx;
        };
        step(dpp_move<kDppXor1>(xc), dpp_move<kDppXor1>(xy), dpp_move<kDppXor1>(idx));
        step(dpp_move<kDppXor2>(xc), dpp_move<kDppXor2>(xy), dpp_move<kDppXor2>(idx));
        step(dpp_move<kDppHalfMirror>(xc), dpp_move<kDppHalfMirror>(xy), dpp_move<kDppHalfMirror>(idx));
        step(dpp_move<kDppMirror>(xc), dpp_move<kDppMirror>(xy), dpp_move<kDppMirror>(idx));
        if (V::L == 64) xy = lane_broadcast(xy, 0);  // one chain per wave: rows 1-3 hold no segment
        return sc.kappa * xy;
    }
    __device__ __forceinline__ double level(const typename Scalar::Coef &cf, double c, double s) const
    {
        return sc.level(cf, c, s);
    }
    // level of x itself, given fresh coefficients
    __device__ __forceinline__ double level0(typename Scalar::Coef &cf, double /*carried*/, bool /*fresh*/) const
    {
        return sc.level(cf, 1.0, 0.0);
    }
};

template <class V, int KC>
struct CoopVmf {
    using Scalar = FastVmf<1, KC>;
    static constexpr bool kLinear = true;
    static constexpr int kVectors = KC;
    static constexpr bool kQuadratic = false;
    static constexpr int kScratchPerGroup = 0;
    static constexpr bool kDistributed = false;
    struct Mine {};
    __device__ __forceinline__ Mine mine_init(int) const { return Mine{}; }
    __device__ __forceinline__ void take_au(Mine &, int, int, double) const {}
    __device__ __forceinline__ void take_ax(Mine &, int, int, double) const {}
    __device__ __forceinline__ void advance(Mine &, double, double) const {}
    __device__ __forceinline__ double level_distributed(const Mine &, int, double, double) const { return 0.0; }
    Scalar sc;
    const double *rows;  // LDS [KC][DPAD]
    __host__ __device__ static size_t lds_doubles() { return (size_t)KC * V::DPAD + KC; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        // K = tb.k <= KC components; the surplus rows are zeros with logc = log 0 (exact zeros in every sum)
        lds_fill(lds, tb.k, V::DPAD, tb.blob, tb.d);
        for (int i = threadIdx.x + tb.k * V::DPAD; i < KC * V::DPAD; i += kBlock) lds[i] = 0.0;
        double *lc = lds + (size_t)KC * V::DPAD;
        for (int i = threadIdx.x; i < KC; i += kBlock)
            lc[i] = i < tb.k ? fmax(tb.blob[(size_t)tb.k * tb.d + i], kLogZero) : kLogZero;
        sc.K = tb.k;
        rows = lds;
        sc.mu = lds;
        sc.logc = lc;
    }
    __device__ __forceinline__ double level(const typename Scalar::Coef &cf, double c, double s) const
    {
        return sc.level(cf, c, s);
    }
    __device__ __forceinline__ double level0(typename Scalar::Coef &cf, double carried, bool fresh) const
    {
        (void)fresh;
        double m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KC; ++k) m = fmax(m, cf.ax[k] + sc.logc[k]);
        cf.m = m;
        carried = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) carried += fm::exp_fast((cf.ax[k] + sc.logc[k]) - m);
        return carried;
    }
};

// Bingham / BinghamFisher for large d: the five coefficients of q(theta) (FastBingham::Coef) are group
// sums; a general A needs (xA)_j and (uA)_j for the lane's own columns j, i.e. all of x and u, which the
// group publishes in an LDS row first; a diagonal A needs nothing but the lane's own slots.
template <class V>
struct CoopBingham {
    using Scalar = FastBingham<1>;
    static constexpr bool kLinear = false;
    static constexpr int kVectors = 0;
    static constexpr bool kQuadratic = true;
    static constexpr bool kDistributed = false;
    static constexpr int kScratchPerGroup = 2 * V::DPAD + 2;
    struct Mine {};
    __device__ __forceinline__ Mine mine_init(int) const { return Mine{}; }
    __device__ __forceinline__ void take_au(Mine &, int, int, double) const {}
    __device__ __forceinline__ void take_ax(Mine &, int, int, double) const {}
    __device__ __forceinline__ void advance(Mine &, double, double) const {}
    __device__ __forceinline__ double level_distributed(const Mine &, int, double, double) const { return 0.0; }
    Scalar sc;
    const double *A;     // LDS [d][DPAD]
    const double *b;     // LDS [DPAD]
    const double *rows;  // unused
    int d;
    bool diagonal;
    __host__ __device__ static size_t lds_doubles(int d) { return (size_t)(d + 1) * V::DPAD; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        d = tb.d;
        diagonal = (tb.k & 1) != 0;
        lds_fill(lds, tb.d + 1, V::DPAD, tb.blob, tb.d);
        A = lds;
        b = lds + (size_t)tb.d * V::DPAD;
        rows = lds;
    }
    __device__ __forceinline__ void make_coop(typename Scalar::Coef &cf, const double (&x)[V::N],
                                              const double (&u)[V::N], int g, double *scratch) const
    {
        double qxx = 0.0, qxu = 0.0, quu = 0.0, bx = 0.0, bu = 0.0;
        if (diagonal) {
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                const int cj = V::comp(g, j);
                const double ajj = cj < d ? A[(size_t)cj * V::DPAD + cj] : 0.0;
                const double xa = x[j] * ajj, ua = u[j] * ajj;
                qxx = fma(xa, x[j], qxx);
                qxu = fma(xa, u[j], fma(ua, x[j], qxu));
                quu = fma(ua, u[j], quu);
            }
        } else {
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                scratch[V::comp(g, j)] = x[j];
                scratch[V::DPAD + 1 + V::comp(g, j)] = u[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double xa[V::N], ua[V::N];
#pragma unroll
            for (int j = 0; j < V::N; ++j) xa[j] = ua[j] = 0.0;
            for (int i = 0; i < d; ++i) {
                const double xi = scratch[i], ui = scratch[V::DPAD + 1 + i];
#pragma unroll
                for (int j = 0; j < V::N; ++j) {
                    const double aij = A[(size_t)i * V::DPAD + V::comp(g, j)];
                    xa[j] = fma(xi, aij, xa[j]);
                    ua[j] = fma(ui, aij, ua[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                qxx = fma(xa[j], x[j], qxx);
                qxu = fma(xa[j], u[j], fma(ua[j], x[j], qxu));
                quu = fma(ua[j], u[j], quu);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < V::N; ++j) {
            const double bj = b[V::comp(g, j)];
            bx = fma(bj, x[j], bx);
            bu = fma(bj, u[j], bu);
        }
        cf.qxx = V::reduce(qxx);
        cf.qxu = V::reduce(qxu);
        cf.quu = V::reduce(quu);
        cf.bx = V::reduce(bx);
        cf.bu = V::reduce(bu);
    }
    __device__ __forceinline__ double level(const typename Scalar::Coef &cf, double c, double s) const
    {
        return sc.level(cf, c, s);
    }
    __device__ __forceinline__ double level0(typename Scalar::Coef &cf, double, bool) const { return cf.qxx + cf.bx; }
};

template <class V, class TP, bool REPLAY, bool STATS = false>
__global__ void __launch_bounds__(kBlock, V::N >= 16 ? 2 : 4) coopfast_kernel(TargetBlock tb, RunBlock a)
{
    using Coef = typename TP::Scalar::Coef;
    constexpr int NV = TP::kVectors;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TP tp;
    tp.stage(lds, tb);
    double *scratch = lds + coop_param_doubles<TP>(tb.d) + (size_t)TP::kScratchPerGroup * (threadIdx.x / V::L);
    __syncthreads();

    const int d = tb.d;
    const int g = threadIdx.x % V::L;
    const int64_t n = a.n_chains;
    // this workgroup's work: the whole launch of the chunk it was launched for, or -- sliced (SliceSched, gsss_device.h) -- the
    // (chunk, step slice) of the ticket it draws
    __shared__ uint32_t sched_word[4];
    const bool sliced = a.sched != nullptr;
    uint32_t chunk = blockIdx.x;
    int32_t s_begin = 0, len = (int32_t)a.n_steps;  // (fast mode: < 2^31 steps per launch)
    bool timed_out = false;
    if (sliced && !SliceSched::take(a, kBlock / V::L, sched_word, chunk, s_begin, len, timed_out)) return;
    const uint64_t step0 = a.step_offset + (uint64_t)s_begin;  // global id of the first step run here
    const int64_t c_raw = (int64_t)chunk * (kBlock / V::L) + threadIdx.x / V::L;
    const bool active = c_raw < n;
    const int64_t c = active ? c_raw : n - 1;
    const bool shrink = a.sampler == GSSS_SHRINK;
    const uint32_t try_base = 1u + (uint32_t)((d + 3) >> 2);

    double x[V::N];
#pragma unroll
    for (int i = 0; i < V::N; ++i) {
        const int cc = V::comp(g, i);
        x[i] = (cc < d) ? a.state[(size_t)cc * n + c] : 0.0;
    }
    auto pdot = [&](const double (&v)[V::N], int r) {  // v . (parameter row r)
        const double *row = tp.rows + (size_t)r * V::DPAD;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) s = fma(row[V::comp(g, i)], v[i], s);
        return V::reduce(s);
    };

    PhiloxDraws<V> dr;
    dr.init(a, c, d);
    int64_t cursor = 0;  // replay
    const double *rp = REPLAY ? a.replay + (size_t)c * a.replay_stride : nullptr;
    int err = 0;
    auto take = [&]() -> double {
        if (cursor >= a.replay_stride) {
            err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            return 0.5;
        }
        return rp[cursor++];
    };

    Coef cf;
    auto my = tp.mine_init(g);
    double lvl = 0.0;
    int64_t n_try = 0, steps_done = 0;
    // (a chain that is alive has made every step so far: retained rows follow from the first step run here)
    int64_t until_keep = a.thin - s_begin % a.thin, row = s_begin / a.thin;
    bool stopped = false;  // sliced: the chain stopped with an error flag in an earlier slice of this launch
    if (sliced) {
        stopped = SliceSched::dead(a, kBlock / V::L)[c] != 0;
        if (timed_out && !stopped) {  // cannot happen (SliceSched::take); never compute from a state that is not there
            err |= GSSS_CHAIN_MAX_TRIES | GSSS_CHAIN_COUNTER_SATURATED;
            stopped = true;
        }
    }

    for (int32_t s = 0; s < len && !err && !stopped; ++s) {
        dr.begin_step(step0 + (uint64_t)s);
        double u[V::N], u_thr, u_th0;
        if (REPLAY) {
            const bool ok = cursor + d <= a.replay_stride;
#pragma unroll
            for (int i = 0; i < V::N; ++i) {
                const int cc = V::comp(g, i);
                u[i] = (cc < d) ? (ok ? rp[cursor + cc] : 0.5) : 0.0;
            }
            if (ok)
                cursor += d;
            else {
                cursor = a.replay_stride;
                err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            }
            u_thr = take();
            u_th0 = shrink ? take() : 0.0;
        } else {
            dr.normals(u, g);
            dr.block(0u, u_thr, u_th0);
        }
        bool x_ok;  // a NaN / Inf state: flagged (the curve's clipped level would swallow it)
        {   // u = spherical_projection(z, x)   (sphere.py:29-33)
            const double xx = vdot<V>(x, x);
            x_ok = xx < INFINITY;
            const double rnx = inv_norm(xx);
            const double cz = vdot<V>(u, x) * rnx;
#pragma unroll
            for (int i = 0; i < V::N; ++i) u[i] = fma(-cz * rnx, x[i], u[i]);
            const double rnw = inv_norm(vdot<V>(u, u));
#pragma unroll
            for (int i = 0; i < V::N; ++i) u[i] *= rnw;
        }
        // (carried coefficients are formed from x again at the first step run here and wherever the GLOBAL step id is a multiple of
        // kCoefRefresh -- where a sliced launch cuts, so that slicing does not change a bit)
        const bool refresh = s == 0 || ((step0 + (uint64_t)s) % kCoefRefresh) == 0;
        if constexpr (TP::kQuadratic) {
            tp.make_coop(cf, x, u, g, scratch);
        } else if constexpr (TP::kDistributed) {  // every lane keeps only the coefficients of its own segment
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                tp.take_au(my, g, r, pdot(u, r));
                if (refresh) tp.take_ax(my, g, r, pdot(x, r));
            }
        } else {
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                cf.au[r] = pdot(u, r);
                if (refresh) cf.ax[r] = pdot(x, r);
            }
        }
        double lvl0;
        if constexpr (TP::kDistributed)
            lvl0 = tp.level_distributed(my, g, 1.0, 0.0);
        else
            lvl0 = tp.level0(cf, lvl, s == 0);
        double thr;
        bool finite;
        if (TP::kLinear) {
            thr = lvl0 * u_thr;
            finite = lvl0 > 0.0 && lvl0 < INFINITY;
        } else {
            thr = lvl0 + fm::log_fast(u_thr);
            finite = lvl0 > -INFINITY && lvl0 < INFINITY;
        }
        if (!finite || !x_ok) {
            err |= GSSS_CHAIN_NONFINITE;
            break;
        }
        double lo, hi;
        if (shrink) {
            hi = kTwoPi * u_th0;
            lo = hi - kTwoPi;
        } else {
            lo = 0.0;
            hi = kTwoPi;
        }
        int t = 0;
        bool accepted = false;
        double sn = 0.0, cs = 1.0;
        while (!accepted) {
            if (t >= a.max_tries) {
                err |= GSSS_CHAIN_MAX_TRIES;
                break;
            }
            // philox-v3: a try's uniform is one 32-bit word; block j feeds tries 4j .. 4j + 3 (a replayed stream hands over doubles)
            constexpr int kPerBlock = REPLAY ? 2 : 4;
            uint32_t wt[4] = {0u, 0u, 0u, 0u};
            if (!REPLAY) dr.words(try_base + (uint32_t)(t >> 2), wt);
#pragma unroll
            for (int h = 0; h < kPerBlock; ++h) {
                if (!accepted && (h == 0 || t < a.max_tries) && !(REPLAY && (err & GSSS_CHAIN_REPLAY_EXHAUSTED))) {
                    const double uu = REPLAY ? take() : try_uniform(wt[h]);
                    const double theta = fma(hi - lo, uu, lo);
                    ++t;
                    fm::sincos_small(theta, sn, cs);
                    if constexpr (TP::kDistributed)
                        lvl = tp.level_distributed(my, g, cs, sn);
                    else
                        lvl = tp.level(cf, cs, sn);
                    accepted = lvl > thr;
                    if (!accepted && shrink) {
                        if (theta < 0.0)
                            lo = theta;
                        else
                            hi = theta;
                    }
                }
            }
            if (REPLAY && (err & GSSS_CHAIN_REPLAY_EXHAUSTED)) break;
        }
        n_try += t;
        if (accepted) {
#pragma unroll
            for (int i = 0; i < V::N; ++i) x[i] = fma(sn, u[i], cs * x[i]);
            if constexpr (TP::kQuadratic) {
                // coefficients are rebuilt from x and u every step
            } else if constexpr (TP::kDistributed) {
                tp.advance(my, cs, sn);
            } else {
#pragma unroll
                for (int r = 0; r < NV; ++r) cf.ax[r] = fma(cs, cf.ax[r], sn * cf.au[r]);
            }
            ++steps_done;
            if ((a.samples != nullptr || STATS) && --until_keep == 0) {
                until_keep = a.thin;
                if (active && a.samples != nullptr) {
#pragma unroll
                    for (int i = 0; i < V::N; ++i) {
                        const int cc = V::comp(g, i);
                        if (cc < d) a.samples[sample_index(a, row, cc, d, c)] = x[i];
                    }
                }
                if constexpr (STATS) {
                    if (active) stats_update_group<V>(a, c, g, d, x);  // (`active` is the same in every lane of a group)
                }
                ++row;
            }
        }
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < V::N; ++i) {
            const int cc = V::comp(g, i);
            if (cc < d) a.state[(size_t)cc * n + c] = x[i];
        }
        if (g == 0) {
            if (a.n_reject) a.n_reject[c] += n_try - steps_done;
            if (a.n_tries) a.n_tries[c] += n_try;
            if (a.err && err) a.err[c] |= err;
            if (sliced && err) SliceSched::dead(a, kBlock / V::L)[c] = 1;  // (every error flag of this kernel stops the chain)
        }
    }
    if (sliced) SliceSched::publish(a, sched_word);
}

template <class V, class TP>
int do_coopfast(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    if (rb.rng_state != nullptr) {  // the cooperative fast kernels draw from Philox only
        set_error("in fast mode the numpy stream is served for lane-per-chain shapes only (this shape runs the "
                  "cooperative fast kernel); use GSSS_MODE_EXACT");
        return GSSS_E_UNSUPPORTED;
    }
    const size_t lds = (coop_param_doubles<TP>(tb.d) + (size_t)TP::kScratchPerGroup * (kBlock / V::L)) * sizeof(double);
    if (lds > 160 * 1024) {
        set_error("target parameters need %zu B of LDS", lds);
        return GSSS_E_UNSUPPORTED;
    }
    auto kern = replay ? coopfast_kernel<V, TP, true> : coopfast_kernel<V, TP, false>;
    if (rb.stats != nullptr) {  // running statistics: a build of its own (the plain kernel carries none of it)
        if (replay) {
            set_error("running statistics are not accumulated from a replayed stream by the cooperative kernels");
            return GSSS_E_UNSUPPORTED;
        }
        kern = coopfast_kernel<V, TP, false, true>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int64_t per_block = kBlock / V::L;
    const int64_t n_chunks = (rb.n_chains + per_block - 1) / per_block;
    // more chunks than the chip holds at once: one workgroup per (chunk, step slice), tickets (SliceSched, gsss_device.h)
    const SlicePlan plan = plan_slices(kern, lds, rb, n_chunks, !replay, st);
    RunBlock rbl = rb;
    rbl.sched = plan.ws;
    rbl.slice_steps = plan.slice_steps;
    hipLaunchKernelGGL(kern, dim3((unsigned)plan.grid), dim3(kBlock), lds, st, tb, rbl);
    hipError_t e = hipGetLastError();
    if (plan.ws) (void)hipFreeAsync(plan.ws, st);
    if (e != hipSuccess) {
        set_error("cooperative fast kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

}  // namespace gsss

// ==========================================================================================
// One wavefront per chain, speculative tries: the latency path for small ensembles.
//
// In a spread launch a chain has a whole wavefront to itself and is bound by the LATENCY of its
// sequential shrink loop.  But the bracket sequence of a step does not depend on any log-density:
// while tries are rejected, theta_t and the shrunken bracket follow from the uniforms alone
// (mcmc.py:395, 400: the side that shrinks is the sign of theta).  So the wave draws the uniforms of
// the first tries at once (one Philox block per lane), replays the bracket recurrence, lets lane t evaluate
// try t, and a ballot finds the first accepted one -- the same try the sequential loop would stop at,
// hence the same chain bit for bit.  Branch-light: no data-dependent loop unless a whole batch is rejected.
// ==========================================================================================
namespace gsss {

constexpr int kSpecTries = 8;  // measured: 4 -> 5.7e5, 8 -> 6.5e5, 16 -> 4.9e5 steps/s for one chain

__device__ __forceinline__ double lane_broadcast_dyn(double v, int lane)  // `lane` wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// NUMPY = true: the draws come from numpy's own PCG64 / ziggurat stream (sequential by nature: every lane
// runs the identical generator; lane t snapshots the generator after try t's uniform, and the accepted
// try's snapshot becomes the stream position, so exactly the reference's numbers are consumed).
template <int D, class TP, bool NUMPY, bool STATS = false>
__global__ void __launch_bounds__(kBlock) wave_kernel(TargetBlock tb, RunBlock a)
{
    using V = LaneVec<D>;
    using Coef = typename TP::Coef;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TP tp;
    tp.stage(lds, tb);
    const fm::Tables tab = stage_tables(lds + TP::lds_doubles());
    NumpyDraws<V> nd;
    if (NUMPY) nd.stage(lds + TP::lds_doubles() + kTabLds);
    __syncthreads();

    const int lane = threadIdx.x % 64;
    const int64_t n = a.n_chains;
    const int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    if (c >= n) return;  // whole wavefront
    const bool shrink = a.sampler == GSSS_SHRINK;
    constexpr uint32_t kNormalBlocks = (uint32_t)((D + 3) / 4);
    constexpr uint32_t kTryBase = 1u + kNormalBlocks;
    constexpr int kPairs = (D + 1) / 2;

    double x[D];  // wave-uniform values: every lane holds the chain's state
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = a.state[(size_t)j * n + c];
    PhiloxDraws<V, true> dr;
    dr.tab = tab;
    if (NUMPY)
        nd.init(a, c, D);
    else
        dr.init(a, c, D);

    Coef cf;
    double lvl = 0.0;
    int64_t n_try = 0, steps_done = 0, until_keep = a.thin, row = 0;
    int err = 0;

    for (int64_t s = 0; s < a.n_steps; ++s) {
        double u[D], u_thr, u_th0, pair_u0 = 0.0, pair_u1 = 0.0;
        uint32_t try_w[4] = {0u, 0u, 0u, 0u};  // Philox: the words of this lane's block (lanes 0-7: the first tries' blocks)
        uint32_t tangent_word = 0u;  // d = 3, Philox: word 3 of the step's block 0 (in lane 8)
        if (NUMPY) {
            nd.normals(u, 0);                            // mcmc.py:387
            u_thr = nd.next_double();                    // mcmc.py:389
            u_th0 = shrink ? nd.next_double() : 0.0;     // mcmc.py:391
        } else {
        dr.begin_step(a.step_offset + (uint64_t)s);
        // ---- every RNG block of the step in one go: lanes 0-7 the tries' blocks, lane 8 block 0, lanes 9.. the normals'
        uint32_t w[4];
        {
            const uint32_t blk = lane < 8 ? kTryBase + (uint32_t)lane : (lane == 8 ? 0u : (uint32_t)(lane - 8));
            dr.words(blk, w);
        }
        pair_u0 = u53(w[0], w[1]);  // lane 8: U_thr, U_theta0 (lanes 0-7: tries 4l .. 4l + 3, one word each -- try_w)
        pair_u1 = u53(w[2], w[3]);
        try_w[0] = w[0];
        try_w[1] = w[1];
        try_w[2] = w[2];
        try_w[3] = w[3];
        tangent_word = w[3];
        u_thr = lane_broadcast(pair_u0, 8);
        u_th0 = lane_broadcast(pair_u1, 8);
        if constexpr (D == 3)  // S^2: block 0 carries U_threshold, U_theta0 (32 bits: word 2) and the tangent angle (word 3)
            u_th0 = (double)(uint32_t)__builtin_amdgcn_readlane((int)w[2], 8) * 0x1.0p-32;
        // ---- normals: Box-Muller pair p on lane p (words of block 1 + p/2, held by lane 9 + p/2)
        if constexpr (D != 3)
        {
            double z0 = 0.0, z1 = 0.0;
            uint32_t wr = 0u, wa = 0u;
#pragma unroll
            for (int p = 0; p < kPairs; ++p) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)w[2 * (p & 1)], 9 + p / 2);
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)w[2 * (p & 1) + 1], 9 + p / 2);
                if (lane == p) {
                    wr = r;
                    wa = g;
                }
            }
            box_muller32(wr, wa, tab, z0, z1);
#pragma unroll
            for (int p = 0; p < kPairs; ++p) {
                u[2 * p] = lane_broadcast(z0, p);
                if (2 * p + 1 < D) u[2 * p + 1] = lane_broadcast(z1, p);
            }
        }
        }
        bool x_ok;  // a NaN / Inf state: flagged (the curve's clipped level would swallow it)
        if constexpr (D == 3 && !NUMPY) {  // Philox stream on S^2: the unit tangent from word 3 of block 0 (lane 8), tangent3
            const double xx = vdot<V>(x, x);
            x_ok = xx < INFINITY;
            const double rnx = inv_norm(xx);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)tangent_word, 8);
            double sn, cs;
            fm::sincos_word_tab(w0, tab, sn, cs);
            tangent3(x[0] * rnx, x[1] * rnx, x[2] * rnx, sn, cs, u[0], u[1], u[2]);
        } else
        {   // u = spherical_projection(z, x)
            const double xx = vdot<V>(x, x);
            x_ok = xx < INFINITY;
            const double rnx = inv_norm(xx);
            double cz = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) cz = fma(u[j], x[j] * rnx, cz);
#pragma unroll
            for (int j = 0; j < D; ++j) u[j] = fma(-cz, x[j] * rnx, u[j]);
            const double rnw = inv_norm(vdot<V>(u, u));
#pragma unroll
            for (int j = 0; j < D; ++j) u[j] *= rnw;
        }
        const double lvl0 = tp.make(cf, x, u, lvl, s == 0);
        double thr;
        bool finite;
        if (TP::kLinear) {
            thr = lvl0 * u_thr;
            finite = lvl0 > 0.0 && lvl0 < INFINITY;
        } else {
            thr = lvl0 + fm::log_fast(u_thr);
            finite = lvl0 > -INFINITY && lvl0 < INFINITY;
        }
        if (!finite || !x_ok) {
            err |= GSSS_CHAIN_NONFINITE;
            break;
        }
        double lo, hi;
        if (shrink) {
            hi = kTwoPi * u_th0;
            lo = hi - kTwoPi;
        } else {
            lo = 0.0;
            hi = kTwoPi;
        }
        // ---- batches of kSpecTries speculative tries
        int t_base = 0;
        bool accepted = false;
        for (;;) {
            if (t_base >= a.max_tries) {
                err |= GSSS_CHAIN_MAX_TRIES;
                break;
            }
            // philox-v3: try t is word t % 4 of block kTryBase + t / 4; lane l < 8 holds block t_base / 4 + l (a batch reads lanes 0, 1)
            uint32_t tw[4] = {try_w[0], try_w[1], try_w[2], try_w[3]};
            if (!NUMPY && t_base > 0)  // 12 % of the steps: a further batch draws its own blocks
                dr.words(kTryBase + (uint32_t)(t_base >> 2) + (uint32_t)(lane & 7), tw);
            // bracket recurrence, identical on every lane; lane t keeps theta_t
            double my_theta = 0.0;
            uint64_t my_sh = 0, my_sl = 0;  // numpy stream: generator state right after try t's uniform
#pragma unroll
            for (int t = 0; t < kSpecTries; ++t) {
                double ut;
                if (NUMPY) {
                    ut = nd.next_double();
                    if (lane == t) {
                        my_sh = nd.sh;
                        my_sl = nd.sl;
                    }
                } else {
                    ut = try_uniform((uint32_t)__builtin_amdgcn_readlane((int)tw[t & 3], t >> 2));
                }
                const double theta = fma(hi - lo, ut, lo);  // mcmc.py:395
                if (lane == t) my_theta = theta;
                if (shrink) {                               // mcmc.py:400, assuming try t is rejected
                    if (theta < 0.0)
                        lo = theta;
                    else
                        hi = theta;
                }
            }
            double sn, cs;
            fm::sincos_tab(my_theta, tab, sn, cs);
            const double my_lvl = tp.level(cf, cs, sn, thr);
            const bool ok = lane < kSpecTries && t_base + lane < a.max_tries && my_lvl > thr;  // mcmc.py:397
            const unsigned long long mask = __ballot(ok);
            if (mask != 0ull) {
                const int T = (int)__builtin_ctzll(mask);  // the first accepted try: where the sequential loop stops
                const double acs = lane_broadcast_dyn(cs, T), asn = lane_broadcast_dyn(sn, T);
                lvl = lane_broadcast_dyn(my_lvl, T);
                if (NUMPY) {  // the stream stands right after the accepted try's uniform
                    nd.sh = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_sh >> 32), T) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_sh, T);
                    nd.sl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_sl >> 32), T) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_sl, T);
                }
#pragma unroll
                for (int j = 0; j < D; ++j) x[j] = fma(asn, u[j], acs * x[j]);  // mcmc.py:396
                n_try += t_base + T + 1;
                accepted = true;
                break;
            }
            const int left = a.max_tries - t_base;
            if (left <= kSpecTries) {
                n_try += a.max_tries;
                err |= GSSS_CHAIN_MAX_TRIES;
                break;
            }
            t_base += kSpecTries;
        }
        if (!accepted) break;
        ++steps_done;
        if ((a.samples != nullptr || STATS) && --until_keep == 0) {
            until_keep = a.thin;
            if (lane < D && a.samples != nullptr) {
#pragma unroll
                for (int j = 0; j < D; ++j)
                    if (lane == j) a.samples[sample_index(a, row, j, D, c)] = x[j];
            }
            if constexpr (STATS) {
                if (lane == 0) stats_update<D>(a, c, x);
            }
            ++row;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.state[(size_t)j * n + c] = x[j];
        if (a.n_reject) a.n_reject[c] += n_try - steps_done;
        if (a.n_tries) a.n_tries[c] += n_try;
        if (a.err && err) a.err[c] |= err;
    }
    if (NUMPY) nd.finish(a, c, lane == 0);
}

template <int D, class TP>
int do_wave(const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    static_assert(16 % kSpecTries == 0 && (D + 1) / 2 <= 8 && (D + 3) / 4 <= 55, "Box-Muller pairs must fit the lanes reserved for them");
    const bool numpy = rb.rng_state != nullptr;
    const size_t lds = (TP::lds_doubles() + kTabLds + (numpy ? NumpyDraws<LaneVec<D>>::kLdsDoubles : 0)) * sizeof(double);
    auto kern = numpy ? wave_kernel<D, TP, true, false> : wave_kernel<D, TP, false, false>;
    if (rb.stats != nullptr) kern = numpy ? wave_kernel<D, TP, true, true> : wave_kernel<D, TP, false, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int64_t grid = (rb.n_chains + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wave kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

}  // namespace gsss

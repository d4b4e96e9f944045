// Registration targets of geosss/registration.py on the device: CoherentPointDrift (:186-293) and
// GaussianMixtureModel (:62-118) scored on a unit quaternion q (d = 4), one lane per chain:
//
//   log_prob(q) = beta * sum_l w_l * logsumexp( { log w_i - |y_l - T_q(x_i)|^2 / (2 sigma^2) + const : i in kNN(l) } [+ outlier] )
//
// R(q) = Rotation.from_quat(q).as_matrix() (pointcloud.py:101-115: scalar last, q normalised); T_q(x) = R x for a
// PointCloud source (3-D target) or the first two rows of R x for a RotationProjection source (2-D target),
// pointcloud.py:252-264, 280-293.  The reference builds a KD tree of the transformed source per evaluation and asks for
// the k nearest; here every lane scans all source points (LDS broadcast reads) and keeps the KMAX smallest squared
// distances in a register-resident sorted list (min / max insertion network) -- the same k nearest, no tree.
// With a 3-D target the target point is rotated into the source frame instead (|y - R x| = |R^T y - x|): 6 flops a pair.
#include "gsss_launch.h"
#include "gsss_mh.h"

namespace gsss {

// tb.k packs the sizes: low 16 bits = source points, high bits = target points (both <= 65535)
__host__ __device__ inline int cpd_ns(int k) { return k & 0xFFFF; }
__host__ __device__ inline int cpd_nt(int k) { return (k >> 16) & 0xFFFF; }
constexpr int kCpdConsts = 8;

template <class V, int KMAX, bool UNIFORM_W>
struct CpdTargetT {
    static_assert(V::L == 1 && V::N == 4, "one lane per chain, quaternion state");
    const double *src;  // LDS [ns][3]
    const double *lw;   // LDS [ns]: log w_i + log_constant
    const double *tgt;  // LDS [nt][3] (third component 0 for a 2-D target)
    const double *tw;   // LDS [nt]
    int ns, nt, dt, kn;
    bool outlier;
    double log_out, half_inv_s2, beta;

    __host__ __device__ static size_t lds_doubles(int k, int /*d*/)
    {
        return 4 * (size_t)cpd_ns(k) + 4 * (size_t)cpd_nt(k) + kCpdConsts;
    }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        ns = cpd_ns(tb.k);
        nt = cpd_nt(tb.k);
        const int total = 4 * ns + 4 * nt + kCpdConsts;
        for (int i = threadIdx.x; i < total; i += kBlock) lds[i] = tb.blob[i];
        src = lds;
        lw = lds + 3 * ns;
        tgt = lw + ns;
        tw = tgt + 3 * nt;
        const double *c = tb.blob + 4 * ns + 4 * nt;  // consts straight from global memory (uniform)
        log_out = c[0];
        half_inv_s2 = c[1];
        beta = c[2];
        outlier = c[3] != 0.0;
        dt = (int)c[4];
        kn = (int)c[5];
    }
    // (d^2 [, lw]) into the sorted list: the list keeps its KMAX smallest keys
    __device__ __forceinline__ void insert(double (&key)[KMAX], double (&val)[KMAX], double k, double v) const
    {
#pragma unroll
        for (int a = 0; a < KMAX; ++a) {
            if constexpr (UNIFORM_W) {
                const double lo = fmin(key[a], k);
                k = fmax(key[a], k);
                key[a] = lo;
            } else {
                const bool sw = k < key[a];
                const double ko = sw ? key[a] : k, vo = sw ? val[a] : v;
                key[a] = sw ? k : key[a];
                val[a] = sw ? v : val[a];
                k = ko;
                v = vo;
            }
        }
    }
    __device__ double logp(const double (&q)[4], int /*g*/, double * /*scratch*/) const
    {
        const double nq = sqrt(fma(q[0], q[0], fma(q[1], q[1], fma(q[2], q[2], q[3] * q[3]))));
        const double x = q[0] / nq, y = q[1] / nq, z = q[2] / nq, w = q[3] / nq;
        const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
        const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
        const double r00 = x2 - y2 - z2 + w2, r01 = 2.0 * (xy - zw), r02 = 2.0 * (xz + yw);
        const double r10 = 2.0 * (xy + zw), r11 = -x2 + y2 - z2 + w2, r12 = 2.0 * (yz - xw);
        const double r20 = 2.0 * (xz - yw), r21 = 2.0 * (yz + xw), r22 = -x2 - y2 + z2 + w2;
        const double lw0 = lw[0];
        double total = 0.0;
        for (int l = 0; l < nt; ++l) {
            const double t0 = tgt[3 * l], t1 = tgt[3 * l + 1], t2 = tgt[3 * l + 2];
            double key[KMAX], val[KMAX];
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                key[a] = INFINITY;
                val[a] = -INFINITY;
            }
            if (dt == 3) {
                // the target point in the source frame: R^T y
                const double u0 = fma(r00, t0, fma(r10, t1, r20 * t2)), u1 = fma(r01, t0, fma(r11, t1, r21 * t2));
                const double u2 = fma(r02, t0, fma(r12, t1, r22 * t2));
                for (int i = 0; i < ns; ++i) {
                    const double d0 = u0 - src[3 * i], d1 = u1 - src[3 * i + 1], d2 = u2 - src[3 * i + 2];
                    insert(key, val, fma(d0, d0, fma(d1, d1, d2 * d2)), UNIFORM_W ? 0.0 : lw[i]);
                }
            } else {
                for (int i = 0; i < ns; ++i) {
                    const double s0 = src[3 * i], s1 = src[3 * i + 1], s2 = src[3 * i + 2];
                    const double d0 = t0 - fma(r00, s0, fma(r01, s1, r02 * s2)), d1 = t1 - fma(r10, s0, fma(r11, s1, r12 * s2));
                    insert(key, val, fma(d0, d0, d1 * d1), UNIFORM_W ? 0.0 : lw[i]);
                }
            }
            // logsumexp over the kn nearest (and the outlier column), registration.py:232-245
            double amax = outlier ? log_out : -INFINITY;
            double term[KMAX];
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                term[a] = a < kn ? fma(-half_inv_s2, key[a], UNIFORM_W ? lw0 : val[a]) : -INFINITY;
                amax = fmax(amax, term[a]);
            }
            double sum = outlier ? fm::exp_fast(log_out - amax) : 0.0;
#pragma unroll
            for (int a = 0; a < KMAX; ++a)
                if (a < kn) sum += fm::exp_fast(term[a] - amax);
            total = fma(tw[l], amax + fm::log_fast(sum), total);
        }
        return beta * total;
    }
    // Registration.gradient (registration.py:55-60): J^T vec(beta * _grad_R), _grad_R of :120-160 / :252-293 (posterior
    // weights gamma clipped to [e^-20, 1]) and the Jacobian d(M / r)/dq of pointcloud.py:135-204 (the two minus signs of the
    // reference cancel).  Needs the neighbours' indices: a (distance, index) insertion network.
    __device__ void grad(const double (&q)[4], int /*g*/, double * /*scratch*/, double (&out)[4]) const
    {
        const double nq = sqrt(fma(q[0], q[0], fma(q[1], q[1], fma(q[2], q[2], q[3] * q[3]))));
        const double x = q[0] / nq, y = q[1] / nq, z = q[2] / nq, w = q[3] / nq;
        const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
        const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
        const double r00 = x2 - y2 - z2 + w2, r01 = 2.0 * (xy - zw), r02 = 2.0 * (xz + yw);
        const double r10 = 2.0 * (xy + zw), r11 = -x2 + y2 - z2 + w2, r12 = 2.0 * (yz - xw);
        const double r20 = 2.0 * (xz - yw), r21 = 2.0 * (yz + xw), r22 = -x2 - y2 + z2 + w2;
        double g00 = 0.0, g01 = 0.0, g02 = 0.0, g10 = 0.0, g11 = 0.0, g12 = 0.0, g20 = 0.0, g21 = 0.0, g22 = 0.0;
        const double inv_s2 = 2.0 * half_inv_s2;
        for (int l = 0; l < nt; ++l) {
            const double t0 = tgt[3 * l], t1 = tgt[3 * l + 1], t2 = tgt[3 * l + 2];
            double key[KMAX];
            int idx[KMAX];
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                key[a] = INFINITY;
                idx[a] = 0;
            }
            const double u0 = fma(r00, t0, fma(r10, t1, r20 * t2)), u1 = fma(r01, t0, fma(r11, t1, r21 * t2));
            const double u2 = fma(r02, t0, fma(r12, t1, r22 * t2));
            for (int i = 0; i < ns; ++i) {
                const double s0 = src[3 * i], s1 = src[3 * i + 1], s2 = src[3 * i + 2];
                double k;
                if (dt == 3) {
                    const double d0 = u0 - s0, d1 = u1 - s1, d2 = u2 - s2;
                    k = fma(d0, d0, fma(d1, d1, d2 * d2));
                } else {
                    const double d0 = t0 - fma(r00, s0, fma(r01, s1, r02 * s2)), d1 = t1 - fma(r10, s0, fma(r11, s1, r12 * s2));
                    k = fma(d0, d0, d1 * d1);
                }
                int v = i;
#pragma unroll
                for (int a = 0; a < KMAX; ++a) {
                    const bool sw = k < key[a];
                    const double ko = sw ? key[a] : k;
                    const int vo = sw ? idx[a] : v;
                    key[a] = sw ? k : key[a];
                    idx[a] = sw ? v : idx[a];
                    k = ko;
                    v = vo;
                }
            }
            double amax = outlier ? log_out : -INFINITY;
            double term[KMAX];
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                term[a] = a < kn ? fma(-half_inv_s2, key[a], lw[idx[a]]) : -INFINITY;
                amax = fmax(amax, term[a]);
            }
            double sum = outlier ? fm::exp_fast(log_out - amax) : 0.0;
#pragma unroll
            for (int a = 0; a < KMAX; ++a)
                if (a < kn) sum += fm::exp_fast(term[a] - amax);
            const double lse = amax + fm::log_fast(sum);
#pragma unroll
            for (int a = 0; a < KMAX; ++a) {
                if (a < kn) {
                    const double lg = fmin(fmax(term[a] - lse, -20.0), 0.0);
                    const double coeff = tw[l] * fm::exp_fast(lg) * inv_s2;
                    const double s0 = src[3 * idx[a]], s1 = src[3 * idx[a] + 1], s2 = src[3 * idx[a] + 2];
                    const double d0 = t0 - fma(r00, s0, fma(r01, s1, r02 * s2)), d1 = t1 - fma(r10, s0, fma(r11, s1, r12 * s2));
                    const double d2 = dt == 3 ? t2 - fma(r20, s0, fma(r21, s1, r22 * s2)) : 0.0;
                    g00 = fma(coeff * d0, s0, g00); g01 = fma(coeff * d0, s1, g01); g02 = fma(coeff * d0, s2, g02);
                    g10 = fma(coeff * d1, s0, g10); g11 = fma(coeff * d1, s1, g11); g12 = fma(coeff * d1, s2, g12);
                    g20 = fma(coeff * d2, s0, g20); g21 = fma(coeff * d2, s1, g21); g22 = fma(coeff * d2, s2, g22);
                }
            }
        }
        // dR/dq_c = dM/dq_c / r - 2 q_c M / r^2 with the quaternion as given (r = |q|^2 + 1e-300)
        const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        const double r = qw * qw + qx * qx + qy * qy + qz * qz + 1e-300;
        const double m00 = qw * qw + qx * qx - qy * qy - qz * qz, m01 = 2.0 * (qx * qy - qw * qz), m02 = 2.0 * (qw * qy + qx * qz);
        const double m10 = 2.0 * (qw * qz + qx * qy), m11 = qw * qw - qx * qx + qy * qy - qz * qz, m12 = 2.0 * (qy * qz - qw * qx);
        const double m20 = 2.0 * (qx * qz - qw * qy), m21 = 2.0 * (qw * qx + qy * qz), m22 = qw * qw - qx * qx - qy * qy + qz * qz;
        const double mg = m00 * g00 + m01 * g01 + m02 * g02 + m10 * g10 + m11 * g11 + m12 * g12 + m20 * g20 + m21 * g21 + m22 * g22;
        // sum_ji dM_ji/dq_c g_ji for c = x, y, z, w
        const double dx = 2.0 * (qx * g00 + qy * g01 + qz * g02 + qy * g10 - qx * g11 - qw * g12 + qz * g20 + qw * g21 - qx * g22);
        const double dy = 2.0 * (-qy * g00 + qx * g01 + qw * g02 + qx * g10 + qy * g11 + qz * g12 - qw * g20 + qz * g21 - qy * g22);
        const double dz = 2.0 * (-qz * g00 - qw * g01 + qx * g02 + qw * g10 - qz * g11 + qy * g12 + qx * g20 + qy * g21 + qz * g22);
        const double dw = 2.0 * (qw * g00 - qz * g01 + qy * g02 + qz * g10 + qw * g11 - qx * g12 - qy * g20 + qx * g21 + qw * g22);
        const double f = 2.0 * mg / (r * r);
        out[0] = beta * (dx / r - qx * f);
        out[1] = beta * (dy / r - qy * f);
        out[2] = beta * (dz / r - qz * f);
        out[3] = beta * (dw / r - qw * f);
    }
    static constexpr int kScratchPerChain = 0;
};

template <class V> using Cpd8U = CpdTargetT<V, 8, true>;
template <class V> using Cpd8W = CpdTargetT<V, 8, false>;
template <class V> using Cpd24U = CpdTargetT<V, 24, true>;
template <class V> using Cpd24W = CpdTargetT<V, 24, false>;

template <template <class> class TT>
static int cpd_run(int draws, const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    if (draws == kDrawsReplay) return do_run<VL4, TT, ReplayDraws>(tb, rb, st);
    if (draws == kDrawsNumpy) return do_run<VL4, TT, NumpyDraws>(tb, rb, st);
    return do_run<VL4, TT, PhiloxDraws>(tb, rb, st);
}
template <template <class> class TT>
static int cpd_rwmh(int draws, int sampler, const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st)
{
    return mh_dispatch<VL4, TT>(draws, sampler, tb, rb, mb, st);
}

// variant: 0 = 8 neighbours / uniform weights, 1 = 8 / weighted, 2 = 24 / uniform, 3 = 24 / weighted
int launch_cpd_run(int variant, int draws, const TargetBlock &tb, const RunBlock &rb, hipStream_t st)
{
    switch (variant) {
    case 0: return cpd_run<Cpd8U>(draws, tb, rb, st);
    case 1: return cpd_run<Cpd8W>(draws, tb, rb, st);
    case 2: return cpd_run<Cpd24U>(draws, tb, rb, st);
    default: return cpd_run<Cpd24W>(draws, tb, rb, st);
    }
}
int launch_cpd_mh(int variant, int draws, int sampler, const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st)
{
    switch (variant) {
    case 0: return cpd_rwmh<Cpd8U>(draws, sampler, tb, rb, mb, st);
    case 1: return cpd_rwmh<Cpd8W>(draws, sampler, tb, rb, mb, st);
    case 2: return cpd_rwmh<Cpd24U>(draws, sampler, tb, rb, mb, st);
    default: return cpd_rwmh<Cpd24W>(draws, sampler, tb, rb, mb, st);
    }
}
int launch_cpd_logprob(int variant, const TargetBlock &tb, const double *x, int64_t n, double *out, bool grad, hipStream_t st)
{
    switch (variant) {
    case 0: return do_logprob<VL4, Cpd8U>(tb, x, n, out, grad, st);
    case 1: return do_logprob<VL4, Cpd8W>(tb, x, n, out, grad, st);
    case 2: return do_logprob<VL4, Cpd24U>(tb, x, n, out, grad, st);
    default: return do_logprob<VL4, Cpd24W>(tb, x, n, out, grad, st);
    }
}

}  // namespace gsss

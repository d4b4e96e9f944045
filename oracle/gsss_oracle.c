/*
 * gsss_oracle.c -- CPU ORACLE for the geodesic slice-sampler hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (geosss_amd/, include/) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and there only as the checker / the CPU number printed beside
 * the GPU number.
 *
 * It is a plain-C, double precision, one-chain-at-a-time restatement of the reference
 * algorithm, function by function, each citing the reference file:line it follows
 * (paths relative to /root/reference):
 *
 *   gor_radial_projection       geosss/sphere.py:10-18
 *   gor_orthogonal_projection   geosss/sphere.py:21-26
 *   gor_spherical_projection    geosss/sphere.py:29-33
 *   gor_distance                geosss/sphere.py:64-68
 *   gor_distance_slerp          geosss/spherical_curve.py:10-32
 *   gor_find_nearest            geosss/spherical_curve.py:95-102
 *   gor_logprob (vMF mixture)   geosss/distributions.py:156-157, 218-221
 *   gor_logprob (Bingham)       geosss/distributions.py:78-86 (+ BinghamFisher :106-114)
 *   gor_logprob (curve vMF)     geosss/distributions.py:272-275
 *   gor_step (shrink)           geosss/mcmc.py:382-401
 *   gor_step (reject)           geosss/mcmc.py:357-374
 *   gor_run                     geosss/mcmc.py:55-77 (the sampling loop, many chains)
 *   gor_logprob (CPD / GMM)     geosss/registration.py:47-53, 103-118, 215-250; pointcloud.py:101-115, 252-264, 280-293
 *   gor_gradient                geosss/distributions.py:88-89, 159-160, 223-227, 277-278
 *   gor_mh_run (RWMH)           geosss/mcmc.py:138-167 with AdaptiveStepsize :80-115
 *   gor_mh_run (spherical HMC)  geosss/mcmc.py:236-318
 *   gor_mh_run (independence sampler, RWMH / independence mixture)  geosss/mcmc.py:179-234
 *
 * Parity pin: the reference's own tests hold no golden vectors for this path
 * (SURVEY.md §4); this oracle is pinned instead by the .npz files in tests/golden/, which
 * tests/golden/make_golden.py produced by running the reference itself in the build
 * container (tests/test_oracle_golden.py checks every one of them).
 *
 * Third-party pieces the reference calls on this path:
 *   scipy.special.logsumexp (scipy 1.15.3)  -> gor_logsumexp restates its algorithm:
 *       a_max + log(m) + log1p(sum_{a_i != a_max} exp(a_i - a_max) / m), m = #ties at max
 *   scipy.special.i0                         -> NOT restated: log i0(kappa_k) is an
 *       x-independent constant; the Python side passes it in (`lognorm`), computed with
 *       scipy exactly as distributions.py:157 does.
 *   numpy PCG64/ziggurat (numpy 2.2.6)       -> three draw sources: (1) replay of a recorded
 *       stream (bit parity with the reference chain); (2) the counter-based Philox4x32-10 stream
 *       specified in DESIGN.md §3, which the HIP kernels implement identically; (3) numpy's
 *       own stream restated -- PCG64 XSL-RR 128/64 (numpy/random/src/pcg64/pcg64.h) feeding
 *       Generator.random / uniform / standard_normal (256-block ziggurat of
 *       numpy/random/src/distributions/distributions.c, tables read out of the installed numpy by
 *       tools/extract_numpy_ziggurat.py) -- so that `default_rng(seed)` chains are reproduced from
 *       the seed alone; tests/test_numpy_stream.py matches it against numpy bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "numpy_ziggurat_tables.h"

#define GOR_VMF_MIXTURE 1
#define GOR_BINGHAM 2
#define GOR_CURVE_VMF 3
#define GOR_CPD 4 /* registration.py: CoherentPointDrift / GaussianMixtureModel on unit quaternions (d = 4) */

#define GOR_SHRINK 0
#define GOR_REJECT 1

#define GOR_ERR_MAX_TRIES 1
#define GOR_ERR_NONFINITE 2
#define GOR_ERR_REPLAY_EXHAUSTED 4

typedef struct {
    int32_t kind;
    int32_t d;
    int32_t k;             /* mixture components / curve knots */
    const double *mu;      /* [k][d]  vMF: kappa_k * direction_k (distributions.py:126-127) */
    const double *lognorm; /* [k]     log(2 pi) + log(i0(|mu_k|))   (distributions.py:157) */
    const double *logw;    /* [k]     log of the normalised weights (distributions.py:213-220) */
    const double *A;       /* [d][d]  Bingham precision (distributions.py:70) */
    const double *b;       /* [d] or NULL: BinghamFisher linear term (distributions.py:106-114) */
    const double *knots;   /* [k][d]  SlerpCurve knots (spherical_curve.py:37,79) */
    double kappa;          /* curve concentration (distributions.py:265) */
    /* GOR_CPD (registration.py:65-293): k = number of source points */
    const double *src;     /* [k][3]  source.positions (pointcloud.py:213-231) */
    const double *src_w;   /* [k]     source.weights */
    const double *tgt;     /* [n_target][target_dim] target.positions */
    const double *tgt_w;   /* [n_target] target.weights */
    int32_t n_target, target_dim; /* 3: PointCloud source (3D-3D); 2: RotationProjection source (3D-2D, pointcloud.py:273-293) */
    int32_t k_nn;          /* neighbours per target point (registration.py:103) */
    int32_t outlier;       /* 1: CoherentPointDrift (outlier column, :186-293); 0: GaussianMixtureModel (:62-118) */
    double sigma, beta, omega, log_volume; /* log_volume = sum(log(ptp(target.positions, 0))) (registration.py:207-213) */
} gor_target;

/* ------------------------------------------------------------------ sphere.py */

static double gor_dot(const double *a, const double *b, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}

/* sphere.py:10-18 : x / (||x|| + 1e-100) */
void gor_radial_projection(const double *x, int d, double *out)
{
    double nrm = sqrt(gor_dot(x, x, d)) + 1e-100;
    for (int i = 0; i < d; ++i) out[i] = x[i] / nrm;
}

/* sphere.py:21-26 : x - (x . n) n  with n = radial_projection(y) */
void gor_orthogonal_projection(const double *x, const double *y, int d, double *out)
{
    double *n = (double *)malloc(sizeof(double) * (size_t)d);
    gor_radial_projection(y, d, n);
    double c = gor_dot(x, n, d);
    for (int i = 0; i < d; ++i) out[i] = x[i] - c * n[i];
    free(n);
}

/* sphere.py:29-33 */
void gor_spherical_projection(const double *x, const double *v, int d, double *out)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)d);
    gor_orthogonal_projection(x, v, d, t);
    gor_radial_projection(t, d, out);
    free(t);
}

static double gor_clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* sphere.py:64-68 */
double gor_distance(const double *x, const double *y, int d)
{
    return acos(gor_clip(gor_dot(x, y, d), -1.0, 1.0));
}

/* ------------------------------------------------------------------ spherical_curve.py */

/* spherical_curve.py:10-32 */
double gor_distance_slerp(const double *x, const double *a, const double *b, int d, double *y)
{
    double theta = gor_distance(a, b, d);
    double ax = gor_dot(a, x, d), bx = gor_dot(b, x, d);
    double t = atan2(bx - ax * cos(theta), ax * sin(theta));
    t = gor_clip(t, 0.0, theta);
    double sa = sin(theta - t), sb = sin(t), den = sin(theta) + 1e-10;
    for (int i = 0; i < d; ++i) y[i] = (sa * a[i] + sb * b[i]) / den;
    return gor_distance(x, y, d);
}

/* spherical_curve.py:95-102 : first minimum wins (np.argmin) */
void gor_find_nearest(const double *knots, int n_knots, int d, const double *x, double *out)
{
    double *y = (double *)malloc(sizeof(double) * (size_t)d);
    double best = INFINITY;
    int have = 0;
    for (int s = 0; s + 1 < n_knots; ++s) {
        double dist = gor_distance_slerp(x, knots + (size_t)s * d, knots + (size_t)(s + 1) * d, d, y);
        if (!have || dist < best) {
            best = dist;
            have = 1;
            memcpy(out, y, sizeof(double) * (size_t)d);
        }
    }
    free(y);
}

/* ------------------------------------------------------------------ distributions.py */

/* scipy 1.15.3 special.logsumexp restated (see header) */
double gor_logsumexp(const double *a, int n)
{
    double amax = a[0];
    for (int i = 1; i < n; ++i)
        if (a[i] > amax) amax = a[i];
    if (!isfinite(amax)) return amax; /* all -inf, or +inf / nan propagate */
    int m = 0;
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        if (a[i] == amax)
            ++m;
        else
            s += exp(a[i] - amax);
    }
    return log1p(s / (double)m) + log((double)m) + amax;
}

/* ------------------------------------------------------------------ registration.py */

/* R = Rotation.from_quat(q).as_matrix() (pointcloud.py:101-115): scalar last, q normalised */
static void gor_quat2matrix(const double *x, double R[3][3])
{
    double nq = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    double qx = x[0] / nq, qy = x[1] / nq, qz = x[2] / nq, qw = x[3] / nq;
    double x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, w2 = qw * qw;
    double xy = qx * qy, zw = qz * qw, xz = qx * qz, yw = qy * qw, yz = qy * qz, xw = qx * qw;
    double M[3][3] = {{x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)},
                      {2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)},
                      {2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2}};
    memcpy(R, M, sizeof(M));
}

/* Registration.log_prob (registration.py:47-53) of CoherentPointDrift (:215-250) / GaussianMixtureModel (:103-118) at the
 * quaternion x; if grad_q != NULL also Registration.gradient (:55-60) = J^T vec(beta * _grad_R) with _grad_R of :120-160 /
 * :252-293 and the Jacobian of pointcloud.py:135-204 */
static double gor_cpd(const gor_target *t, const double *x, double *grad_q)
{
    double R[3][3];
    gor_quat2matrix(x, R);
    int ns = t->k, dt = t->target_dim, kn = t->k_nn;
    /* transform_positions: positions @ R.T (3D, pointcloud.py:252-264) or positions @ R[:-1].T (projection, :280-293) */
    double *p = (double *)calloc((size_t)ns * 3, sizeof(double));
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < dt; ++j)
            p[3 * i + j] = R[j][0] * t->src[3 * i] + R[j][1] * t->src[3 * i + 1] + R[j][2] * t->src[3 * i + 2];
    double *d2 = (double *)malloc(sizeof(double) * (size_t)ns);
    int *idx = (int *)malloc(sizeof(int) * (size_t)ns);
    double *terms = (double *)malloc(sizeof(double) * (size_t)(kn + 1));
    double s2 = t->sigma * t->sigma;
    /* registration.py:215-219 (CPD) / :110-111 (GMM) */
    double log_const = t->outlier ? log(1.0 - t->omega) - 0.5 * dt * log(2.0 * 3.141592653589793 * s2)
                                  : -(0.5 * dt * log(2.0 * 3.141592653589793 * s2));
    double log_out = log(t->omega + 1e-308) - t->log_volume; /* registration.py:236 */
    double total = 0.0;
    double gR[3][3] = {{0.0}};
    for (int l = 0; l < t->n_target; ++l) {
        for (int i = 0; i < ns; ++i) {
            double acc = 0.0;
            for (int j = 0; j < dt; ++j) {
                double df = t->tgt[(size_t)l * dt + j] - p[3 * i + j];
                acc += df * df;
            }
            d2[i] = acc;
            idx[i] = i;
        }
        /* KDTree(trans_src).query(target, k): the k nearest, ascending (registration.py:95-101) */
        for (int a = 0; a < kn; ++a) {
            int best = a;
            for (int i = a + 1; i < ns; ++i)
                if (d2[idx[i]] < d2[idx[best]]) best = i;
            int tmp = idx[a];
            idx[a] = idx[best];
            idx[best] = tmp;
        }
        for (int a = 0; a < kn; ++a) {
            double dist = sqrt(d2[idx[a]]); /* the tree returns distances; the score squares them again */
            terms[a] = log(t->src_w[idx[a]]) + (-0.5 * dist * dist / s2) + log_const;
        }
        int nt = kn;
        if (t->outlier) terms[nt++] = log_out;
        double lse = gor_logsumexp(terms, nt);
        total += lse * t->tgt_w[l]; /* registration.py:118, 250 */
        if (grad_q) {
            for (int a = 0; a < kn; ++a) {
                double lg = terms[a] - lse;             /* registration.py:137-139, 262-264: clip(log gamma, -20, 0) */
                lg = lg < -20.0 ? -20.0 : (lg > 0.0 ? 0.0 : lg);
                double coeff = t->tgt_w[l] * exp(lg) / s2;
                for (int j = 0; j < 3; ++j) {            /* d_lk = y_l (padded) - x_k_trans (padded), :151-157 */
                    double yl = j < dt ? t->tgt[(size_t)l * dt + j] : 0.0;
                    double dj = yl - p[3 * idx[a] + j];
                    for (int i = 0; i < 3; ++i) gR[j][i] += coeff * dj * t->src[3 * idx[a] + i]; /* einsum "lk,lkj,lki->ji" */
                }
            }
        }
    }
    free(p);
    free(d2);
    free(idx);
    free(terms);
    if (grad_q) {
        /* _grad_R returns -grad_R, jacobian_rotation_matrix returns -J: grad_q = beta * J^T vec(grad_R) with
         * J = d (M / r) / dq, M the unnormalised rotation entries, r = |q|^2 + 1e-300 (pointcloud.py:135-204) */
        double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
        double r = qw * qw + qx * qx + qy * qy + qz * qz + 1e-300;
        double M[3][3] = {{qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qw * qy + qx * qz)},
                          {2 * (qw * qz + qx * qy), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qw * qx)},
                          {2 * (qx * qz - qw * qy), 2 * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz}};
        /* dM/dq_c, c = x, y, z, w */
        double dM[4][3][3] = {
            {{2 * qx, 2 * qy, 2 * qz}, {2 * qy, -2 * qx, -2 * qw}, {2 * qz, 2 * qw, -2 * qx}},
            {{-2 * qy, 2 * qx, 2 * qw}, {2 * qx, 2 * qy, 2 * qz}, {-2 * qw, 2 * qz, -2 * qy}},
            {{-2 * qz, -2 * qw, 2 * qx}, {2 * qw, -2 * qz, 2 * qy}, {2 * qx, 2 * qy, 2 * qz}},
            {{2 * qw, -2 * qz, 2 * qy}, {2 * qz, 2 * qw, -2 * qx}, {-2 * qy, 2 * qx, 2 * qw}}};
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) acc += (dM[c][j][i] / r - 2.0 * x[c] * M[j][i] / (r * r)) * gR[j][i];
            grad_q[c] = t->beta * acc;
        }
    }
    return t->beta * total;
}

double gor_logprob(const gor_target *t, const double *x)
{
    int d = t->d;
    if (t->kind == GOR_VMF_MIXTURE) {
        /* distributions.py:218-221 on top of :156-157 */
        double *p = (double *)malloc(sizeof(double) * (size_t)t->k);
        for (int k = 0; k < t->k; ++k) {
            double lp = gor_dot(x, t->mu + (size_t)k * d, d) - t->lognorm[k];
            p[k] = lp + t->logw[k];
        }
        double r = gor_logsumexp(p, t->k);
        free(p);
        return r;
    }
    if (t->kind == GOR_BINGHAM) {
        /* distributions.py:84-86 : sum((x @ A) * x) */
        double s = 0.0;
        for (int j = 0; j < d; ++j) {
            double xa = 0.0;
            for (int i = 0; i < d; ++i) xa += x[i] * t->A[(size_t)i * d + j];
            s += xa * x[j];
        }
        if (t->b) s += gor_dot(x, t->b, d); /* distributions.py:113-114 */
        return s;
    }
    if (t->kind == GOR_CPD) return gor_cpd(t, x, NULL);
    if (t->kind == GOR_CURVE_VMF) {
        /* distributions.py:272-275 */
        double *y = (double *)malloc(sizeof(double) * (size_t)d);
        gor_find_nearest(t->knots, t->k, d, x, y);
        double r = t->kappa * gor_dot(x, y, d);
        free(y);
        return r;
    }
    return NAN;
}

void gor_logprob_batch(const gor_target *t, const double *X, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = gor_logprob(t, X + (size_t)i * t->d);
}

/* ------------------------------------------------------------------ RNG stream (DESIGN.md §3 "Random streams") */

static void gor_philox_round(uint32_t c[4], const uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void gor_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        gor_philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(c));
}

/* 53-bit uniform in [0,1) from two words, numpy's construction */
static double gor_u53(uint32_t a, uint32_t b)
{
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* block `blk` of (seed, chain, step): four 32-bit words */
void gor_stream_words(uint64_t seed, uint64_t chain, uint64_t step, uint32_t blk, uint32_t w[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = blk;
    ctr[1] = (uint32_t)step;
    ctr[2] = (uint32_t)chain;
    ctr[3] = (uint32_t)((chain >> 32) & 0xFFFFu) | ((uint32_t)((step >> 32) & 0xFFFFu) << 16);
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    gor_philox4x32_10(ctr, key, w);
}

/* ... read as two 53-bit uniforms */
void gor_stream_block(uint64_t seed, uint64_t chain, uint64_t step, uint32_t blk, double u[2])
{
    uint32_t w[4];
    gor_stream_words(seed, chain, step, blk, w);
    u[0] = gor_u53(w[0], w[1]);
    u[1] = gor_u53(w[2], w[3]);
}

/* ... or as two Box-Muller pairs from four 32-bit uniforms (radius word in (0,1], angle word in [0,1)) */
/* In SINGLE precision, operation for operation fm::box_muller_f32 of geosss_amd/csrc/gsss_math.h (IEEE add, multiply, fma,
 * square root and exact bit operations only, so host and device agree to the bit): the normals only give the tangent its
 * direction (mcmc.py:387, sphere.py:29-33), 24 bits are plenty.  Polynomials: the classic single-precision minimax sets for
 * log(1 + x) and sin / cos on [-pi/4, pi/4] (Cephes). */
static uint32_t gor_f32_bits(float x)
{
    uint32_t b;
    memcpy(&b, &x, sizeof(b));
    return b;
}
static float gor_bits_f32(uint32_t b)
{
    float x;
    memcpy(&x, &b, sizeof(x));
    return x;
}
static void gor_box_muller32(uint32_t wr, uint32_t wa, double *z0, double *z1)
{
    const float vf = (float)((wr >> 8) + 1u);
    const uint32_t vb = gor_f32_bits(vf);
    int e = (int)(vb >> 23) - 127 - 24;
    float m = gor_bits_f32((vb & 0x007FFFFFu) | 0x3F800000u);
    const int up = m > 1.41421356f;
    m = up ? 0.5f * m : m;
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
    float ln = fmaf(x * z, p, fe * -2.12194440e-4f);
    ln = fmaf(-0.5f, z, ln);
    ln = (x + ln) + fe * 0.693359375f;
    const float t = -2.0f * ln;
    const float r = sqrtf(t > 0.0f ? t : 0.0f);
    const uint32_t k = wa >> 29;
    const float a = (float)(wa & 0x1FFFFFFFu) * 1.46291807926715968e-9f;
    const float y = (k & 1u) ? a - 0.785398163397448f : a;
    const float yy = y * y;
    float sp = fmaf(-1.9515295891e-4f, yy, 8.3321608736e-3f);
    sp = fmaf(sp, yy, -1.6666654611e-1f);
    const float sn = fmaf(y * yy, sp, y);
    float cp = fmaf(2.443315711809948e-5f, yy, -1.388731625493765e-3f);
    cp = fmaf(cp, yy, 4.166664568298827e-2f);
    const float cs = fmaf(yy * yy, cp, fmaf(-0.5f, yy, 1.0f));
    const uint32_t q = ((k + 1u) >> 1) & 3u;
    const float c0 = (q & 1u) ? sn : cs, s0 = (q & 1u) ? cs : sn;
    const float c = (q == 1u || q == 2u) ? -c0 : c0;
    const float sg = (q >= 2u) ? -s0 : s0;
    *z0 = (double)(r * c);
    *z1 = (double)(r * sg);
}
/* exposed for tests/test_math.py: the oracle's pair against the host build of gsss_math.h, bit for bit */
void gor_box_muller32_fill(const uint32_t *wr, const uint32_t *wa, int64_t n, double *z0, double *z1)
{
    for (int64_t i = 0; i < n; ++i) gor_box_muller32(wr[i], wa[i], z0 + i, z1 + i);
}

/* ---- numpy's PCG64 (XSL-RR 128/64, setseq) and the distributions built on it ---- */
typedef struct {
    unsigned __int128 state, inc;
} gor_pcg64;

static uint64_t gor_pcg64_next64(gor_pcg64 *g)
{
    const unsigned __int128 mult = ((unsigned __int128)2549297995355413924ULL << 64) | 4865540595714422341ULL;
    g->state = g->state * mult + g->inc; /* step, then output the NEW state */
    uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

static double gor_npy_double(gor_pcg64 *g) { return (double)(gor_pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0); }

/* random_standard_normal of numpy's distributions.c */
static double gor_npy_standard_normal(gor_pcg64 *g)
{
    for (;;) {
        uint64_t r = gor_pcg64_next64(g);
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 0x1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * NPY_ZIG_WI[idx];
        if (sign & 0x1) x = -x;
        if (rabs < NPY_ZIG_KI[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -NPY_ZIG_NOR_INV_R * log1p(-gor_npy_double(g));
                double yy = -log1p(-gor_npy_double(g));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 0x1) ? -(NPY_ZIG_NOR_R + xx) : NPY_ZIG_NOR_R + xx;
            }
        } else {
            if (((NPY_ZIG_FI[idx - 1] - NPY_ZIG_FI[idx]) * gor_npy_double(g) + NPY_ZIG_FI[idx]) < exp(-0.5 * x * x))
                return x;
        }
    }
}

/* test hooks: fill arrays from a PCG64 state given as (state_hi, state_lo, inc_hi, inc_lo) */
static void gor_pcg_load(gor_pcg64 *g, const uint64_t *w)
{
    g->state = ((unsigned __int128)w[0] << 64) | w[1];
    g->inc = ((unsigned __int128)w[2] << 64) | w[3];
}
static void gor_pcg_store(const gor_pcg64 *g, uint64_t *w)
{
    w[0] = (uint64_t)(g->state >> 64);
    w[1] = (uint64_t)g->state;
    w[2] = (uint64_t)(g->inc >> 64);
    w[3] = (uint64_t)g->inc;
}
void gor_npy_fill(uint64_t *pcg, int64_t n_normal, double *normals, int64_t n_uniform, double *uniforms)
{
    gor_pcg64 g;
    gor_pcg_load(&g, pcg);
    for (int64_t i = 0; i < n_normal; ++i) normals[i] = gor_npy_standard_normal(&g);
    for (int64_t i = 0; i < n_uniform; ++i) uniforms[i] = gor_npy_double(&g);
    gor_pcg_store(&g, pcg);
}

typedef struct {
    /* numpy stream */
    gor_pcg64 *pcg;
    /* replay */
    const double *replay;
    int64_t replay_len, cursor;
    int exhausted;
    /* philox */
    uint64_t seed, chain, step;
    int d;
    int64_t try_idx;
    uint32_t cached[4]; /* the words of the try block in use */
} gor_draws;

static double gor_take(gor_draws *g)
{
    if (g->cursor >= g->replay_len) {
        g->exhausted = 1;
        return 0.5;
    }
    return g->replay[g->cursor++];
}

/* the d standard normals of a step (mcmc.py:387) */
static void gor_draw_normals(gor_draws *g, double *z)
{
    int d = g->d;
    if (g->pcg) {
        for (int i = 0; i < d; ++i) z[i] = gor_npy_standard_normal(g->pcg);
        return;
    }
    if (g->replay) {
        for (int i = 0; i < d; ++i) z[i] = gor_take(g);
        return;
    }
    for (int j = 0; 4 * j < d; ++j) { /* block 1+j carries normals 4j .. 4j+3 */
        uint32_t w[4];
        double zz[4];
        gor_stream_words(g->seed, g->chain, g->step, (uint32_t)(1 + j), w);
        gor_box_muller32(w[0], w[1], &zz[0], &zz[1]);
        gor_box_muller32(w[2], w[3], &zz[2], &zz[3]);
        for (int i = 0; i < 4 && 4 * j + i < d; ++i) z[4 * j + i] = zz[i];
    }
}

/* Library stream (Philox) on S^2, d = 3: the unit tangent u at x is drawn directly -- an angle phi in the tangent plane
 * from word 3 of block 0 -- instead of three normals projected and normalised (sphere.py:29-33 has the same law: the
 * direction of the projected normal vector is uniform on the tangent circle).  n = x / |x|, (b1, b2) the branch-free
 * orthonormal basis of Duff et al. 2017, u = cos(phi) b1 + sin(phi) b2: the expressions of tangent3 in gsss_device.h.
 * The replayed and the numpy streams keep the reference's normals. */
static void gor_tangent3(const double *x, uint32_t w0, double *u)
{
    double n[3];
    gor_radial_projection(x, 3, n);
    const double phi = 2.0 * 3.141592653589793 * ((double)w0 * 0x1.0p-32);
    const double sn = sin(phi), cs = cos(phi);
    const double s = copysign(1.0, n[2]);
    const double a = -1.0 / (s + n[2]);
    const double b = n[0] * n[1] * a;
    const double b10 = 1.0 + s * n[0] * n[0] * a, b11 = s * b, b12 = -s * n[0];
    const double b20 = b, b21 = s + n[1] * n[1] * a, b22 = -n[1];
    u[0] = fma(sn, b20, cs * b10);
    u[1] = fma(sn, b21, cs * b11);
    u[2] = fma(sn, b22, cs * b12);
}

/* the threshold uniform (mcmc.py:389) and theta0 uniform (mcmc.py:391) */
static void gor_draw_step_uniforms(gor_draws *g, double *u_thr, double *u_theta0, int need_theta0)
{
    if (g->pcg) {
        *u_thr = gor_npy_double(g->pcg);
        *u_theta0 = need_theta0 ? gor_npy_double(g->pcg) : 0.0;
        return;
    }
    if (g->replay) {
        *u_thr = gor_take(g);
        *u_theta0 = need_theta0 ? gor_take(g) : 0.0;
        return;
    }
    double u[2];
    gor_stream_block(g->seed, g->chain, g->step, 0u, u);
    *u_thr = u[0];
    *u_theta0 = u[1];
}

/* the uniform of try number `g->try_idx` (mcmc.py:395).  Library stream (philox-v3, round 5): ONE 32-bit word per try -- try t
 * is word t % 4 of block 1 + nb + t / 4, u = w / 2^32, so theta = lo + (hi - lo) u keeps 32 bits of resolution RELATIVE to the
 * bracket however far it has shrunk (rounds 1-4, philox-v2: two 53-bit uniforms a block; half the blocks now -- the try
 * loop of the lane kernels is to a third Philox arithmetic) */
static double gor_draw_try(gor_draws *g)
{
    if (g->pcg) return gor_npy_double(g->pcg);
    if (g->replay) return gor_take(g);
    int64_t t = g->try_idx++;
    if ((t & 3) == 0) {
        uint32_t nb = (uint32_t)((g->d + 3) / 4);
        gor_stream_words(g->seed, g->chain, g->step, 1u + nb + (uint32_t)(t >> 2), g->cached);
    }
    return (double)g->cached[t & 3] * 0x1.0p-32;
}

/* ------------------------------------------------------------------ mcmc.py */

/* one transition.  Returns the error bits; *tries = number of log_prob(y) evaluations. */
static int gor_step(const gor_target *t, double *x, gor_draws *g, int sampler, int64_t max_tries, int64_t *tries,
                    double *trace_thr, double *scratch)
{
    int d = t->d;
    double *z = scratch, *u = scratch + d, *y = scratch + 2 * d;
    const double two_pi = 2.0 * 3.141592653589793; /* 2 * np.pi */

    g->try_idx = 0;
    double u_thr, u_th0;
    if (d == 3 && !g->pcg && !g->replay) {
        /* library stream on S^2: block 0 carries the whole set-up of the step -- U_threshold (53 bits: words 0, 1),
         * U_theta0 (32 bits: word 2), the angle of the tangent direction (word 3, gor_tangent3); block 1 is not drawn */
        uint32_t w[4];
        gor_stream_words(g->seed, g->chain, g->step, 0u, w);
        u_thr = (double)(((uint64_t)(w[0] >> 5) << 26) | (uint64_t)(w[1] >> 6)) * 0x1.0p-53;
        u_th0 = (double)w[2] * 0x1.0p-32;
        gor_tangent3(x, w[3], u);
    } else {
        gor_draw_normals(g, z);                /* mcmc.py:387 */
        gor_spherical_projection(z, x, d, u);  /* mcmc.py:387 */
        gor_draw_step_uniforms(g, &u_thr, &u_th0, sampler == GOR_SHRINK);
    }
    double px = gor_logprob(t, x);
    double threshold = px + log(u_thr);    /* mcmc.py:389 */
    if (trace_thr) *trace_thr = threshold;
    if (!(px > -INFINITY) || isnan(px)) {
        *tries = 0;
        return GOR_ERR_NONFINITE;
    }
    double lo, hi;
    if (sampler == GOR_SHRINK) {
        double theta0 = 0.0 + (two_pi - 0.0) * u_th0; /* mcmc.py:391 */
        lo = theta0 - two_pi;                            /* mcmc.py:392 */
        hi = theta0;
    } else {
        lo = 0.0;                                        /* mcmc.py:367 */
        hi = two_pi;
    }
    int64_t n = 0;
    for (;;) {
        if (n >= max_tries) {
            *tries = n;
            return GOR_ERR_MAX_TRIES;
        }
        double theta = lo + (hi - lo) * gor_draw_try(g); /* mcmc.py:395 */
        double c = cos(theta), s = sin(theta);
        for (int i = 0; i < d; ++i) y[i] = c * x[i] + s * u[i]; /* mcmc.py:396 */
        ++n;
        if (gor_logprob(t, y) > threshold) {             /* mcmc.py:397 */
            memcpy(x, y, sizeof(double) * (size_t)d);
            *tries = n;
            return g->exhausted ? GOR_ERR_REPLAY_EXHAUSTED : 0;
        }
        if (sampler == GOR_SHRINK) {                     /* mcmc.py:400 */
            if (theta < 0.0)
                lo = theta;
            else
                hi = theta;
        }
        if (g->exhausted) {
            *tries = n;
            return GOR_ERR_REPLAY_EXHAUSTED;
        }
    }
}

/*
 * Advance n_chains independent chains by n_steps transitions each.
 *   state      [n_chains][d] in/out (row per chain)
 *   samples    [n_chains][n_keep][d] or NULL; a state is kept after every `thin`-th step
 *   n_reject, n_tries [n_chains] are ADDED to; err [n_chains] is OR-ed into
 *   replay     [n_chains][replay_stride] recorded draws in consumption order, or NULL
 *   thr_trace  [n_chains][n_steps] or NULL
 *   pcg_state  [n_chains][4] or NULL: numpy PCG64 (state_hi, state_lo, inc_hi, inc_lo) per chain,
 *              in/out -- draws then come from numpy's own stream (default_rng semantics)
 * chain ids are chain_offset + i, step ids step_offset + s (the RNG counter), so any
 * partition of chains / steps over calls or devices gives the same numbers.
 */
int gor_run(const gor_target *t, double *state, int64_t n_chains, int64_t n_steps, int64_t thin, uint64_t seed,
            uint64_t chain_offset, uint64_t step_offset, int sampler, int64_t max_tries, double *samples,
            int64_t *n_reject, int64_t *n_tries, int32_t *err, const double *replay, int64_t replay_stride,
            double *thr_trace, int n_threads, uint64_t *pcg_state)
{
    int d = t->d;
    if (thin < 1) thin = 1;
    int64_t n_keep = n_steps / thin;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t c = 0; c < n_chains; ++c) {
        double *scratch = (double *)malloc(sizeof(double) * (size_t)(3 * d));
        double *x = state + (size_t)c * d;
        gor_draws g;
        memset(&g, 0, sizeof(g));
        g.d = d;
        g.seed = seed;
        g.chain = chain_offset + (uint64_t)c;
        if (replay) {
            g.replay = replay + (size_t)c * replay_stride;
            g.replay_len = replay_stride;
        }
        gor_pcg64 pcg;
        if (pcg_state) {
            gor_pcg_load(&pcg, pcg_state + 4 * c);
            g.pcg = &pcg;
        }
        for (int64_t s = 0; s < n_steps; ++s) {
            g.step = step_offset + (uint64_t)s;
            int64_t tries = 0;
            int e = gor_step(t, x, &g, sampler, max_tries, &tries, thr_trace ? thr_trace + c * n_steps + s : NULL,
                             scratch);
            if (n_tries) n_tries[c] += tries;
            if (n_reject) n_reject[c] += e ? tries : tries - 1;
            if (err && e) err[c] |= e;
            if (samples && (s + 1) % thin == 0) {
                int64_t row = (s + 1) / thin - 1;
                memcpy(samples + ((size_t)c * n_keep + row) * d, x, sizeof(double) * (size_t)d);
            }
            if (e) break;
        }
        if (pcg_state) gor_pcg_store(&pcg, pcg_state + 4 * c);
        free(scratch);
    }
    return 0;
}

/* ------------------------------------------------------------------ RWMH and spherical HMC (mcmc.py:80-332) */

#define GOR_RWMH 2
#define GOR_HMC 3
#define GOR_INDEP 4 /* IndependenceSampler, mcmc.py:179-182 */
#define GOR_MIX 5   /* MixtureRWMHIndependenceSampler, mcmc.py:185-234 */

/* Distribution.gradient of the three target families.  BinghamFisher inherits Bingham.gradient in the
 * reference (distributions.py:106-114 defines log_prob only), i.e. 2 A x without b: restated as is. */
void gor_gradient(const gor_target *t, const double *x, double *g)
{
    int d = t->d;
    if (t->kind == GOR_VMF_MIXTURE) {
        /* distributions.py:223-227 : sum_k exp(p_k) mu_k / exp(logsumexp(p)) */
        double *p = (double *)malloc(sizeof(double) * (size_t)t->k);
        for (int k = 0; k < t->k; ++k) p[k] = gor_dot(x, t->mu + (size_t)k * d, d) - t->lognorm[k] + t->logw[k];
        double den = exp(gor_logsumexp(p, t->k));
        for (int i = 0; i < d; ++i) {
            double acc = 0.0;
            for (int k = 0; k < t->k; ++k) acc += exp(p[k]) * t->mu[(size_t)k * d + i];
            g[i] = acc / den;
        }
        free(p);
    } else if (t->kind == GOR_CPD) {
        (void)gor_cpd(t, x, g);
    } else if (t->kind == GOR_BINGHAM) {
        for (int i = 0; i < d; ++i) g[i] = 2.0 * gor_dot(t->A + (size_t)i * d, x, d); /* distributions.py:88-89 : 2 * A @ x */
    } else {
        gor_find_nearest(t->knots, t->k, d, x, g);                                    /* distributions.py:277-278 */
        for (int i = 0; i < d; ++i) g[i] *= t->kappa;
    }
}

/* numpy's random_standard_gamma for shape > 1 (Marsaglia-Tsang, numpy/random/src/distributions/distributions.c):
 * what Generator.gamma(d/2) in MetropolisHastings.propose (mcmc.py:143) draws from. */
static double gor_npy_standard_gamma(gor_pcg64 *g, double shape)
{
    double b = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * b);
    for (;;) {
        double X, V;
        do {
            X = gor_npy_standard_normal(g);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        double U = gor_npy_double(g);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
        if (log(U) < 0.5 * X * X + b * (1.0 - V + log(V))) return b * V;
    }
}
void gor_npy_gamma_fill(uint64_t *pcg, double shape, int64_t n, double *out)
{
    gor_pcg64 g;
    gor_pcg_load(&g, pcg);
    for (int64_t i = 0; i < n; ++i) out[i] = gor_npy_standard_gamma(&g, shape);
    gor_pcg_store(&g, pcg);
}

/* r = sqrt(2 * gamma(d / 2)) of mcmc.py:143.  numpy stream: numpy's own gamma; replay: the recorded gamma
 * variate; Philox stream: the norm of d further standard normals (blocks 1 + nq ...), a chi_d variate like r */
static double gor_draw_chi(gor_draws *g)
{
    int d = g->d;
    if (g->pcg) return sqrt(2.0 * gor_npy_standard_gamma(g->pcg, 0.5 * (double)d));
    if (g->replay) return sqrt(2.0 * gor_take(g));
    uint32_t nq = (uint32_t)((d + 3) / 4);
    double ss = 0.0;
    for (int j = 0; 4 * j < d; ++j) {
        uint32_t w[4];
        double zz[4];
        gor_stream_words(g->seed, g->chain, g->step, 1u + nq + (uint32_t)j, w);
        gor_box_muller32(w[0], w[1], &zz[0], &zz[1]);
        gor_box_muller32(w[2], w[3], &zz[2], &zz[3]);
        for (int i = 0; i < 4 && 4 * j + i < d; ++i) ss += zz[i] * zz[i];
    }
    return sqrt(ss);
}

static double gor_draw_accept(gor_draws *g)
{
    if (g->pcg) return gor_npy_double(g->pcg);
    if (g->replay) return gor_take(g);
    double u[2];
    gor_stream_block(g->seed, g->chain, g->step, 0u, u);
    return u[0];
}

/* the uniform that picks the kernel of MixtureRWMHIndependenceSampler (mcmc.py:213); Philox stream: the other half of block 0 */
static double gor_draw_mix(gor_draws *g)
{
    if (g->pcg) return gor_npy_double(g->pcg);
    if (g->replay) return gor_take(g);
    double u[2];
    gor_stream_block(g->seed, g->chain, g->step, 0u, u);
    return u[1];
}

/* project(x, v) of mcmc.py:231-235 : x - v (v . x) */
static void gor_project(double *x, const double *v, int d)
{
    double c = gor_dot(v, x, d);
    for (int i = 0; i < d; ++i) x[i] -= v[i] * c;
}

/*
 * n_steps transitions of MetropolisHastings (sampler = GOR_RWMH, mcmc.py:138-167) or SphericalHMC (GOR_HMC,
 * mcmc.py:270-318) for n_chains independent chains.
 *   state    [n_chains][d] in/out;  momenta [n_chains][d] in/out or NULL (HMC: the v half of the reference's state)
 *   stepsize [n_chains] in/out: AdaptiveStepsize (mcmc.py:108-115) multiplies it by 1.02 / 0.98 after each of the
 *            first `adapt_steps` steps of this call
 *   n_accept [n_chains] ADDED to;  accept_trace / stepsize_trace [n_chains][n_steps] or NULL
 * Draws per step in the reference's order: RWMH gamma(d/2), d normals, 1 uniform; HMC d normals, 1 uniform.
 * GOR_INDEP (mcmc.py:179-182): d normals, 1 uniform; the inherited stepsize adaptation runs and is never used.
 * GOR_MIX (mcmc.py:185-234): 1 uniform (RWMH iff < mix_alpha), the chosen proposal's draws, 1 uniform; only RWMH proposals
 * adapt the stepsize and advance the burn-in counter: adapt_left [n_chains] in/out counts the RWMH proposals that still
 * adapt (adapt_steps unused); n_rwmh [n_chains] or NULL is ADDED to; proposal_trace [n_chains][n_steps] or NULL gets 1 for RWMH.
 */
int gor_mh_run(const gor_target *t, double *state, double *momenta, int64_t n_chains, int64_t n_steps, int64_t thin,
               uint64_t seed, uint64_t chain_offset, uint64_t step_offset, int sampler, double *stepsize,
               int64_t adapt_steps, int n_leapfrog, double *samples, int64_t *n_accept, int32_t *err,
               const double *replay, int64_t replay_stride, int n_threads, uint64_t *pcg_state, uint8_t *accept_trace,
               double *stepsize_trace, double mix_alpha, int64_t *adapt_left, int64_t *n_rwmh, uint8_t *proposal_trace)
{
    int d = t->d;
    if (thin < 1) thin = 1;
    int64_t n_keep = n_steps / thin;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t c = 0; c < n_chains; ++c) {
        double *buf = (double *)malloc(sizeof(double) * (size_t)(6 * d));
        double *z = buf, *y = buf + d, *xx = buf + 2 * d, *vv = buf + 3 * d, *gr = buf + 4 * d, *v = buf + 5 * d;
        double *x = state + (size_t)c * d;
        double eps = stepsize[c];
        gor_draws g;
        memset(&g, 0, sizeof(g));
        g.d = d;
        g.seed = seed;
        g.chain = chain_offset + (uint64_t)c;
        if (replay) {
            g.replay = replay + (size_t)c * replay_stride;
            g.replay_len = replay_stride;
        }
        gor_pcg64 pcg;
        if (pcg_state) {
            gor_pcg_load(&pcg, pcg_state + 4 * c);
            g.pcg = &pcg;
        }
        if (momenta)
            memcpy(v, momenta + (size_t)c * d, sizeof(double) * (size_t)d);
        else
            memset(v, 0, sizeof(double) * (size_t)d);
        for (int64_t s = 0; s < n_steps; ++s) {
            g.step = step_offset + (uint64_t)s;
            int accepted, use_rwmh = 1;
            if (sampler == GOR_RWMH || sampler == GOR_INDEP || sampler == GOR_MIX) {
                if (sampler == GOR_MIX) use_rwmh = gor_draw_mix(&g) < mix_alpha; /* mcmc.py:213 */
                if (sampler == GOR_INDEP) use_rwmh = 0;
                if (use_rwmh) {
                    double r = gor_draw_chi(&g);                   /* mcmc.py:143 */
                    gor_draw_normals(&g, z);                       /* mcmc.py:144 */
                    for (int i = 0; i < d; ++i) y[i] = r * x[i] + eps * z[i];
                } else {
                    gor_draw_normals(&g, y);                       /* mcmc.py:181 */
                }
                gor_radial_projection(y, d, y);                    /* mcmc.py:145, :182 */
                double prob = gor_logprob(t, y) - gor_logprob(t, x); /* mcmc.py:152 */
                accepted = log(gor_draw_accept(&g)) < prob;        /* mcmc.py:153 */
                if (accepted) memcpy(x, y, sizeof(double) * (size_t)d);
            } else {
                gor_draw_normals(&g, z);                           /* mcmc.py:278 : v = project(normal, x) */
                memcpy(v, z, sizeof(double) * (size_t)d);
                gor_project(v, x, d);
                double h0 = 0.5 * gor_dot(v, v, d) - gor_logprob(t, x); /* mcmc.py:285-286 */
                memcpy(xx, x, sizeof(double) * (size_t)d);
                memcpy(vv, v, sizeof(double) * (size_t)d);
                gor_gradient(t, xx, gr);
                gor_project(gr, xx, d);
                for (int i = 0; i < d; ++i) vv[i] += 0.5 * eps * gr[i]; /* mcmc.py:301 */
                for (int l = 0; l < n_leapfrog; ++l) {
                    double norm = sqrt(gor_dot(vv, vv, d));
                    double cs = cos(eps * norm), sn = sin(eps * norm);
                    for (int i = 0; i < d; ++i) {
                        double yi = xx[i];
                        xx[i] = yi * cs + (vv[i] / norm) * sn;     /* mcmc.py:307 */
                        vv[i] = vv[i] * cs - (yi * norm) * sn;     /* mcmc.py:308 */
                    }
                    gor_gradient(t, xx, gr);
                    gor_project(gr, xx, d);
                    double f = (l < n_leapfrog - 1) ? eps : 0.5 * eps; /* mcmc.py:310-313 */
                    for (int i = 0; i < d; ++i) vv[i] += f * gr[i];
                }
                gor_radial_projection(xx, d, xx);                  /* mcmc.py:315 */
                double h1 = 0.5 * gor_dot(vv, vv, d) - gor_logprob(t, xx);
                accepted = log(gor_draw_accept(&g)) < h0 - h1;     /* mcmc.py:318-319 */
                if (accepted) {
                    memcpy(x, xx, sizeof(double) * (size_t)d);
                    memcpy(v, vv, sizeof(double) * (size_t)d);
                }
            }
            if (n_accept) n_accept[c] += accepted;
            if (sampler == GOR_MIX) {                              /* mcmc.py:226-228 */
                if (use_rwmh) {
                    if (n_rwmh) n_rwmh[c] += 1;
                    if (adapt_left && adapt_left[c] > 0) {         /* mcmc.py:113-115: the counter advances on RWMH proposals */
                        eps *= accepted ? 1.02 : 0.98;
                        adapt_left[c] -= 1;
                    }
                }
                if (proposal_trace) proposal_trace[c * n_steps + s] = (uint8_t)use_rwmh;
            } else if (s < adapt_steps) {
                eps *= accepted ? 1.02 : 0.98;                     /* mcmc.py:113-115 */
            }
            if (accept_trace) accept_trace[c * n_steps + s] = (uint8_t)accepted;
            if (stepsize_trace) stepsize_trace[c * n_steps + s] = eps;
            if (samples && (s + 1) % thin == 0)
                memcpy(samples + ((size_t)c * n_keep + ((s + 1) / thin - 1)) * d, x, sizeof(double) * (size_t)d);
            if (g.exhausted) {
                if (err) err[c] |= GOR_ERR_REPLAY_EXHAUSTED;
                break;
            }
        }
        stepsize[c] = eps;
        if (momenta) memcpy(momenta + (size_t)c * d, v, sizeof(double) * (size_t)d);
        if (pcg_state) gor_pcg_store(&pcg, pcg_state + 4 * c);
        free(buf);
    }
    return 0;
}

/* initial states: the device twin of sphere.sample_sphere (sphere.py:39-50) draws its
 * normals from stream blocks of step id 2^48-1 (reserved), see DESIGN.md */
void gor_sample_sphere(uint64_t seed, uint64_t chain_offset, int64_t n, int d, double *out)
{
    gor_draws g;
    for (int64_t c = 0; c < n; ++c) {
        memset(&g, 0, sizeof(g));
        g.d = d;
        g.seed = seed;
        g.chain = chain_offset + (uint64_t)c;
        g.step = 0xFFFFFFFFFFFFull;
        double *z = out + (size_t)c * d;
        gor_draw_normals(&g, z);
        gor_radial_projection(z, d, z);
    }
}

int gor_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

/*
 * gsss.h -- C ABI of libgsss_hip.so: many-chain geodesic slice sampling on the sphere,
 * hand-written HIP for gfx950 (MI355X).
 *
 * The reference (microscopic-image-analysis/geosss) is pure Python and has no FFI for
 * this path; its boundary is the duck-typed protocol
 *     Sampler(distribution, initial_state, seed).sample(n_samples, burnin)
 *     distribution.log_prob(x)
 * (geosss/mcmc.py:28-77, 340-401; geosss/distributions.py:16-25).  The entry points
 * below are what a ctypes binding of that protocol needs; each names the reference
 * code it replaces.  INTEGRATION.md shows the binding a geosss maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative GSSS_E_* code and never throws;
 *    gsss_last_error() gives the message of the calling thread's last failure.
 *  - pointers named *_dev are DEVICE pointers on the target's device; the caller owns
 *    every buffer, the library keeps none of them past the call (asynchronous work is
 *    ordered on `stream`, so keep buffers alive until the stream has drained).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *    enqueue work; nothing synchronises except gsss_stream_synchronize/gsss_memcpy_*.
 *  - all floating point is IEEE double, as in the reference.
 *  - chain states are stored component-major ("SoA"): state_dev[j * n_chains + c] is
 *    component j of chain c, so that one wavefront reads 64 consecutive doubles.
 */
#ifndef GSSS_H
#define GSSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSSS_ABI_VERSION 10

/* target families (geosss/distributions.py) */
#define GSSS_VMF_MIXTURE 1 /* MixtureModel of VonMisesFisher  :117-160, :209-227 */
#define GSSS_BINGHAM 2     /* Bingham                         :36-103            */
#define GSSS_CURVE_VMF 3   /* CurvedVonMisesFisher+SlerpCurve :261-278, spherical_curve.py:10-32,74-102 */
#define GSSS_CPD 4         /* registration.py: CoherentPointDrift :186-293 / GaussianMixtureModel :62-118 on unit quaternions (d = 4) */

/* samplers (geosss/mcmc.py) */
#define GSSS_SHRINK 0 /* ShrinkageSphericalSliceSampler.__next__  :382-401 */
#define GSSS_REJECT 1 /* RejectionSphericalSliceSampler.__next__  :357-374 */
#define GSSS_RWMH 2   /* MetropolisHastings.__next__ (random-walk MH)  :138-167, AdaptiveStepsize :80-115 */
#define GSSS_HMC 3    /* SphericalHMC.__next__ (leapfrog on the sphere) :270-319 */
#define GSSS_INDEP 4  /* IndependenceSampler: proposal = a uniform point of the sphere          :179-182 */
#define GSSS_MIX 5    /* MixtureRWMHIndependenceSampler: RWMH with probability alpha, else the independence proposal :185-234 */

/* arithmetic of the proposal evaluation */
#define GSSS_MODE_EXACT 0 /* y = cos*x + sin*u formed, log_prob(y) evaluated from y op by op as the reference does */
#define GSSS_MODE_FAST 1  /* log_prob restricted to the great circle: O(1) per try after O(d) per step (same value to ~1e-14) */

/* gsss_run_args.variant in GSSS_MODE_FAST: 0 = library's choice (tries screened in single precision with a rigorous error
 * margin, double precision where the margin does not decide: same chains); this value forces the all-double kernels */
#define GSSS_VARIANT_FAST_DOUBLE 100
/* verification: the library's choice of kernel with the verdicts of its single-precision screen ignored where that kernel
 * can do so (the group-speculative curve-vMF kernel and the lane kernels of d = 11 .. 16: every try is decided in double
 * precision by the same arithmetic, so the run must reproduce variant 0 bit for bit); other kernels run as with variant 0 */
#define GSSS_VARIANT_FAST_VERIFY 101

/* return codes */
#define GSSS_OK 0
#define GSSS_E_INVALID (-1)     /* bad argument */
#define GSSS_E_UNSUPPORTED (-2) /* shape outside what the kernels cover */
#define GSSS_E_HIP (-3)         /* a HIP runtime call failed */
#define GSSS_E_NO_DEVICE (-4)   /* no usable gfx950 device */

/* per-chain error bits written to gsss_run_args.err_dev */
#define GSSS_CHAIN_MAX_TRIES 1        /* shrink/reject loop hit max_tries; state left at x */
#define GSSS_CHAIN_NONFINITE 2        /* log_prob(state) was -inf/NaN (reference would spin forever, mcmc.py:394) */
#define GSSS_CHAIN_REPLAY_EXHAUSTED 4 /* replay stream shorter than the draws consumed */
#define GSSS_CHAIN_COUNTER_SATURATED 8 /* fast mode counts the proposals of ONE launch in 32 bits: more than 2^32-1 of them
                                          in a single gsss_run call leave n_tries/n_reject too small (split the call) */

/* gsss_run_args.stats_flags */
#define GSSS_STATS_NO_SECOND_MOMENT 1 /* stats_dev has no sum x_i x_j rows (T = 0 below): they grow as d^2 -- 20 100 rows at d = 200 */

typedef struct gsss_target gsss_target; /* opaque; owns a small device parameter block */

/*
 * Host-side description of a target.  All pointers are HOST pointers, copied at create.
 *   VMF_MIXTURE: k components on S^{d-1}.  mu[k][d] = kappa_k * direction_k exactly as
 *       VonMisesFisher stores it (distributions.py:126-127);  logc[k] = log(w_k) - log(2 pi)
 *       - log(i0(|mu_k|)) = the x-independent part of distributions.py:157 and :220.
 *   BINGHAM:     A[d][d] symmetric (distributions.py:67-70); if mu is non-NULL it is the vector b[d] of a
 *       BinghamFisher target, log_prob = x^T A x + x.b (distributions.py:106-114).
 *   CURVE_VMF:   knots[k][d] unit vectors, kappa (distributions.py:263-265).
 *   CPD:         see the fields below; GSSS_MODE_EXACT, lane-per-chain kernels; samplers GSSS_SHRINK / GSSS_REJECT / GSSS_RWMH.
 */
typedef struct gsss_target_desc {
    int32_t kind;
    int32_t d;
    int32_t k;
    int32_t reserved;
    const double *mu;
    const double *logc;
    const double *A;
    const double *knots;
    double kappa;
    /* GSSS_CPD: rigid registration of a 3-D source cloud onto a 3-D target (PointCloud source, pointcloud.py:206-270) or onto a
     * 2-D target after projection (RotationProjection source, :273-293).  The state is a unit quaternion (x, y, z, w), d = 4;
     * log_prob(q) = beta * sum_l w_l logsumexp_{i in kNN(l)} [ log w_i - |y_l - R(q) x_i|^2 / (2 sigma^2) + const ] with the
     * outlier term of CoherentPointDrift when `outlier` (registration.py:47-53, 103-118, 215-250).  k = number of source points. */
    const double *source;   /* [k][3] */
    const double *source_w; /* [k] */
    const double *target;   /* [n_target][target_dim] */
    const double *target_w; /* [n_target] */
    int32_t n_target;
    int32_t target_dim;     /* 3 or 2 */
    int32_t k_nn;           /* neighbours per target point, <= 24 and <= k */
    int32_t outlier;        /* 1: CoherentPointDrift, 0: GaussianMixtureModel */
    double sigma;
    double beta;
    double omega;
    double log_volume;      /* sum_j log(ptp(target[:, j]))  (registration.py:207-213) */
} gsss_target_desc;

/*
 * One launch: advance n_chains chains by n_steps transitions each.
 * Replaces the loop `while len(samples) < n: samples.append(next(self))` of Sampler.sample
 * (mcmc.py:66-69) for all chains at once.
 *
 * Random numbers: counter-based Philox4x32-10 keyed by `seed`; the draws of chain c at
 * step s are a pure function of (seed, chain_offset + c, step_offset + s), so any split of
 * the chains over devices or of the steps over calls reproduces the same numbers
 * (DESIGN.md §3 "Random streams").  On this stream a step draws d normals for the tangent direction as the reference does
 * (mcmc.py:387; Box-Muller pairs formed in single precision: they only set a direction) -- except on S^2 (d = 3), where the uniformly distributed unit tangent is drawn directly as one angle in
 * the tangent plane and one block carries the whole set-up of a step (same law; the two sources below keep the reference's
 * normals and draw order).  The uniform of a try (mcmc.py:395) is ONE 32-bit word of the stream, u = w / 2^32 -- try t is word t % 4
 * of block 1 + ceil(d / 4) + t / 4 ("philox-v3"; theta = lo + (hi - lo) u keeps 32 bits of resolution relative to the bracket
 * however far it has shrunk) -- the threshold uniform (mcmc.py:389) has 53 bits.
 * If replay_dev is non-NULL the draws are read from it instead:
 * per chain `replay_stride` doubles in the order the reference consumes them
 * (d normals, u_threshold, [u_theta0,] u_try, u_try, ... ; next step ...) -- this is how the
 * parity tests reproduce reference chains bit for bit.  A third source, rng_state_dev, is numpy's
 * own PCG64 + ziggurat stream restated on the device: `seed -> chain` exactly as in the reference.
 */
typedef struct gsss_run_args {
    double *state_dev;         /* [d][n_chains] in/out                                             */
    double *samples_dev;       /* NULL or the state after every thin-th step, layout per samples_chain_rows    */
    int64_t *n_reject_dev;     /* [n_chains] or NULL; rejections are ADDED (RejectionSphericalSliceSampler.n_reject) */
    int64_t *n_tries_dev;      /* [n_chains] or NULL; log_prob(y) evaluations are ADDED            */
    int32_t *err_dev;          /* [n_chains] or NULL; GSSS_CHAIN_* bits are OR-ed in               */
    const double *replay_dev;  /* [n_chains][replay_stride] or NULL                                */
    int64_t replay_stride;
    int64_t n_chains;
    int64_t n_steps;
    int64_t thin;              /* >= 1 */
    uint64_t seed;
    uint64_t chain_offset;     /* global id of chain 0 of this call (< 2^48) */
    uint64_t step_offset;      /* global id of step 0 of this call  (< 2^48 - 1) */
    int32_t sampler;           /* GSSS_SHRINK | GSSS_REJECT | GSSS_RWMH | GSSS_HMC | GSSS_INDEP | GSSS_MIX */
    int32_t mode;              /* GSSS_MODE_EXACT | GSSS_MODE_FAST */
    int32_t max_tries;         /* > 0: give up a step after this many proposals */
    int32_t variant;           /* 0 = library's choice; otherwise a kernel variant id (gsss_variant_name) */
    uint64_t *rng_state_dev;   /* [n_chains][4] or NULL.  Non-NULL selects NUMPY'S OWN STREAM instead of Philox: per chain the
                                  PCG64 words (state_hi, state_lo, inc_hi, inc_lo) of np.random.default_rng(seed).bit_generator,
                                  read at entry and written back at exit, so a chain consumes exactly the numbers the reference's
                                  sampler.rng would (mcmc.py:45, 387-395).  GSSS_MODE_EXACT, or GSSS_MODE_FAST for the lane-per-chain shapes
                                  (gsss_variant_name "fast-lane": one wavefront per chain when spread, one lane per chain when packed);
                                  `seed` and the offsets are then unused */
    int64_t samples_chain_rows; /* 0: samples_dev is [n_steps/thin][d][n_chains] (component-major, like state_dev).
                                   R > 0: samples_dev points into a [n_chains][R][d] array -- the reference's (chains, draws,
                                   dims) order -- and this call writes rows 0 .. n_steps/thin-1 of every chain's run of R rows
                                   (offset the pointer by row0*d doubles to continue a run across calls) */
    int32_t placement;         /* lane-per-chain kernels: 0 = library's choice (spread for small ensembles: <= 3072 chains in fast mode, 1536 for curve targets, 2048 in exact mode), 1 = packed
                                  (64 chains per wavefront: throughput), 2 = spread (one chain per wavefront: chains of a
                                  small ensemble do not wait for each other's shrink loops; same numbers either way) */
    int32_t stats_lags;        /* L >= 0: lags of the running autocovariance sums (see stats_dev) */
    double *stats_dev;         /* NULL or [gsss_stats_rows(d, stats_modes, stats_lags, stats_flags)][n_chains]: running statistics of the RETAINED
                                  series (every thin-th state, whether or not samples_dev is given), ADDED to across calls, so that
                                  moments, geodesic step, hopping frequency, mode occupancy and the autocorrelation / IAT / ESS of one
                                  projection need no stored draws (geosss/utils.py:96-134, sphere.py:64-68, scripts/bingham.py:23-25,
                                  scripts/vMF_diagnostics.py:335-342).  Zero it before the first call.  Rows, with T = d (d + 1) / 2 (0 with
                                  GSSS_STATS_NO_SECOND_MOMENT), K = stats_modes, L = stats_lags:
                                    0                count n of retained draws
                                    1 .. d           the last retained draw
                                    .. + d           sum of the draws
                                    .. + T           sum of x_i x_j, i <= j, row-major upper triangle
                                    .. + 1           sum over consecutive draws of arccos(clip(x_t . x_{t-1}, -1, 1))
                                    .. + 1           number of consecutive draws with sign(x_t . h) != sign(x_{t-1} . h)
                                    .. + K           number of draws whose first-largest x . mode_k is k
                                    .. + 2           sum p, sum p^2 of the projection p_t = x_t . w
                                    .. + L           sum_t p_t p_{t-l}, l = 1 .. L
                                    .. + L           ring of the last L values of p (slot t mod L)
                                    .. + L           the first L values of p
                                  Every slice-sampler kernel family accumulates them (lane, lane-group and cooperative layouts;
                                  the group layouts form the second moments through cross-lane reads: d <= 64 with them,
                                  any d with GSSS_STATS_NO_SECOND_MOMENT); GSSS_E_UNSUPPORTED for registration targets */
    const double *stats_dirs_dev; /* [2 + stats_modes][d]: w, h, then the mode directions; required with stats_dev */
    int32_t stats_modes;       /* K >= 0 */
    int32_t n_leapfrog;        /* GSSS_HMC: leapfrog steps per proposal (SphericalHMC(n_steps=10), mcmc.py:243) */
    /* GSSS_RWMH / GSSS_HMC (the baselines of the paper; GSSS_MODE_EXACT, any layout; n_tries / n_reject / stats unused): */
    double *stepsize_dev;      /* [n_chains] in/out: proposal scale / leapfrog stepsize of every chain (mcmc.py:97, 133, 241) */
    int64_t *n_accept_dev;     /* [n_chains] or NULL; accepted proposals are ADDED (MetropolisHastings.n_accept) */
    double *momenta_dev;       /* GSSS_HMC: NULL or [d][n_chains] in/out, the momentum half of the reference's state (mcmc.py:262) */
    int64_t adapt_steps;       /* the first adapt_steps steps of this call multiply the stepsize by 1.02 after an accepted and by
                                  0.98 after a rejected proposal (AdaptiveStepsize.adapt_stepsize during burn-in, mcmc.py:108-115).
                                  Draws per step, in the reference's order: RWMH gamma(d/2), d normals, one uniform; HMC d normals,
                                  one uniform (replay_dev holds the gamma variate itself; the Philox stream uses the norm of d
                                  further normals, the same chi_d law; rng_state_dev restates numpy's gamma).
                                  GSSS_INDEP: d normals, one uniform (the stepsize is adapted like RWMH's, as the reference's class
                                  inherits it, and never used).  GSSS_MIX: one uniform (RWMH iff it is < mixing_probability), then the
                                  draws of the chosen proposal, then the accept uniform; on the Philox stream the two uniforms are the
                                  two halves of block 0 (accept, mix) */
    /* GSSS_MIX (MixtureRWMHIndependenceSampler, mcmc.py:185-234): */
    double mixing_probability; /* alpha: probability of the RWMH kernel */
    int64_t *adapt_left_dev;   /* [n_chains] in/out: RWMH proposals that still adapt the stepsize.  The reference's burn-in counter
                                  only advances on RWMH proposals (mcmc.py:108-115 called from :226-228), so the adaptation of a chain
                                  ends after `burnin` RWMH proposals, not after `burnin` steps; adapt_steps is unused */
    int64_t *n_rwmh_dev;       /* [n_chains] or NULL; RWMH proposals are ADDED (rwmh_counter; indep_counter = steps - that) */
    /* ABI 9 */
    double *momenta_samples_dev; /* GSSS_HMC: NULL or the momentum half of the state after every thin-th step, laid out like
                                    samples_dev (same samples_chain_rows): SphericalHMC.sample(return_momenta=True), mcmc.py:321-332 */
    double *stepsize_trace_dev;  /* GSSS_RWMH / GSSS_INDEP / GSSS_MIX: NULL or [n_steps][n_chains]; entry [s][c] = the stepsize of chain c
                                    after step s if that step made a RWMH proposal, else NaN
                                    (MixtureRWMHIndependenceSampler.rwmh_stepsize_vals, mcmc.py:201, 228) */
    int32_t stats_flags;         /* GSSS_STATS_* bits; changes the row layout of stats_dev (gsss_stats_rows takes the same flags) */
    int32_t reserved0;
} gsss_run_args;

int gsss_abi_version(void);
const char *gsss_last_error(void);

/* sha256 (hex) over the kernel sources this library was BUILT from (the .h, .hip and .inc files of geosss_amd/csrc and this header, by name:
 * geosss_amd/build.py source_digest), "unknown" for a build that did not pass it.  Profiles are bound to the binary through it:
 * bench.py quotes the committed rocprofv3 counters only for the digest of the loaded library.  Nothing in the reference. */
const char *gsss_source_digest(void);

/* number of visible gfx950 devices (0 if none); does not fail */
int gsss_device_count(void);

/* distributions.py ctor twins: copy the parameters to `device` */
int gsss_target_create(const gsss_target_desc *desc, int device, gsss_target **out);
int gsss_target_destroy(gsss_target *t);
int gsss_target_dim(const gsss_target *t);

/* Distribution.log_prob for n points (distributions.py:84-86, :156-157, :218-221, :272-275).
 * x_dev is ROW-major [n][d] (numpy's natural layout for pdf.log_prob(samples)). */
int gsss_logprob(const gsss_target *t, const double *x_dev, int64_t n, double *out_dev, void *stream);

/* Distribution.gradient for n points (distributions.py:88-89 Bingham 2 A x -- also what BinghamFisher inherits --, :159-160
 * vMF mu, :223-227 mixture, :277-278 curve kappa * nearest; registration.py:55-60 the pose score's gradient with respect to the
 * quaternion): the functions the spherical HMC kernel evaluates (GSSS_HMC), for rows of x_dev [n][d] -> grad_dev [n][d]. */
int gsss_gradient(const gsss_target *t, const double *x_dev, int64_t n, double *grad_dev, void *stream);

/* The sampler (see gsss_run_args). */
int gsss_run(const gsss_target *t, const gsss_run_args *args, void *stream);

/* What the calling thread's last gsss_run launched, for logs and benches: the grid of the sampler kernel and the slice length if
 * the launch was SLICED (kernels whose chunks of chains do not fit the chip at once run one workgroup per (chunk, slice of steps)
 * and hand the chunk's state from slice to slice through HBM: (16 d + 32) / slice_steps more bytes per chain-step than an unsliced
 * launch; same results bit for bit; GSSS_SLICE_STEPS in the environment sets the length, 0 turns it off; a launch is cut into at
 * most 64 slices per chunk -- long launches get longer slices, n_steps = 10^6 -> 15 680 steps --, so the chain of hand-overs a
 * workgroup may wait on stays short for any n_steps < 2^31).  slice_steps 0:
 * unsliced; grid 0: a kernel family that never slices.  sliced_fraction: the share of the chains that ran sliced (the lane
 * kernels slice only a small last round of workgroups).  Any pointer may be NULL.  There is nothing in the reference this
 * replaces. */
int gsss_last_launch(int64_t *grid_out, int32_t *slice_steps_out, double *sliced_fraction_out);

/* Number of rows of gsss_run_args.stats_dev for dimension d, K modes, L lags and the GSSS_STATS_* flags (< 0: bad argument). */
int64_t gsss_stats_rows(int32_t d, int32_t n_modes, int32_t n_lags, int32_t flags);

/* 1 if gsss_run accepts `mode` for this target's shape (fast mode is built for the shapes listed in
 * geosss_amd/csrc/gsss_fast_*.hip), else 0. */
int gsss_mode_supported(const gsss_target *t, int32_t mode);

/* Names the kernel variant gsss_run would use / the variants available, for logs and benches. */
const char *gsss_variant_name(const gsss_target *t, int32_t mode, int32_t variant);
/* The kernel instantiation gsss_run launches for this target (what a rocprofv3 kernel trace shows, template arguments
 * abbreviated), e.g. "fast_kernel<3, FastVmf<3, 3>>"; placement as in gsss_run_args (2 = spread).  "" if unsupported.
 * The string lives in thread-local storage until the thread's next call. */
const char *gsss_kernel_name(const gsss_target *t, int32_t mode, int32_t variant, int32_t placement);

/* sphere.sample_sphere twin (sphere.py:39-50): n uniform points on S^{d-1}, component-major,
 * from the reserved step id of the RNG stream. */
int gsss_sample_sphere(uint64_t seed, uint64_t chain_offset, int64_t n, int32_t d, double *state_dev, int device,
                       void *stream);

/* Verification of the library stream's set-up on S^2 (d = 3).  The reference draws z ~ N(0, I_3) and takes
 * u = sphere.spherical_projection(z, x) (mcmc.py:387, sphere.py:29-33): a uniformly distributed unit tangent at x.  The Philox
 * stream draws that tangent directly from ONE 32-bit word w: u = cos(phi) b1 + sin(phi) b2, phi = 2 pi w / 2^32, in a fixed
 * orthonormal basis (b1, b2) of the tangent plane at n = x / |x|.  This entry point evaluates exactly what the sampler kernels
 * evaluate there, for n points: x_dev [n][3] states (any norm), w_dev [n] angle words -> out_dev [n][12] = n, b1, b2, u.
 * table_driven != 0: the sincos of the throughput kernels (screened / fast / one-wavefront), 0: the exact kernels'.
 * tests/test_hip_tangent.py holds it to the reference's own tangents (tests/golden/tangent_kat.npz). */
int gsss_tangent_s2(const double *x_dev, const uint32_t *w_dev, int64_t n, int32_t table_driven, double *out_dev, int device,
                    void *stream);

/* Verification of the single-precision screen.  In fast mode four tries out of five are decided on the hardware's
 * single-precision sin / cos / 2^x / log2 / sqrt with an error margin; the decision protected is
 * mcmc.py:397 `if p(y) > threshold`, and it is the double-precision one only if each instruction's worst-case error is inside the
 * constant the margin is built from.  gsss_screen_constants: out[0..5) = the constants compiled into the library --
 * kSinCosErr32 (|v_sin/cos_f32(fl32(theta / 2 pi)) - sin/cos(theta)|, |theta| <= 2 pi, argument rounding included), kExp2Err32
 * (relative, normal results), kLog2Err32 (log2 of a double's mantissa, its rounding to single included), kSqrtRelErr32, 2^-24.
 * gsss_f32_error_sweep: evaluates EVERY float of an instruction's argument range on `device` against double precision
 * (synchronous; ~1 s) and writes four maxima to out_host:
 *   which 0  sin / cos of t revolutions, every |t| <= 1:   [0] sin, [1] cos at the float; [2], [3] with 2 pi x half an ulp of t added
 *                                                           (any theta whose revolutions round to t) -- to hold against kSinCosErr32
 *   which 1  2^x, every finite float:                       [0] relative error on normal results (kExp2Err32), [1] absolute error where
 *                                                           the result is below the normals, [2] overflows not returned as +inf (0)
 *   which 2  log2 x, every float of [0.5, 2]:               [0] at the float, [1] on [0.5, 1] with half an ulp / (x ln 2) added (kLog2Err32)
 *   which 3  sqrt x, every positive normal float:           [0] relative error (kSqrtRelErr32)
 * n_swept_out (may be NULL): how many floats were evaluated.  tests/test_hip_screen_bounds.py.  Nothing in the reference. */
int gsss_screen_constants(double *out, int32_t n);
int gsss_f32_error_sweep(int32_t which, double *out_host, uint64_t *n_swept_out, int device, void *stream);

/* Layout changes between numpy's row-major arrays and the component-major device layout.
 *   gsss_rows_to_components: in [n][d]            -> out [d][n]
 *   gsss_components_to_rows: in [d][n]            -> out [n][d]
 *   gsss_samples_to_chains:  in [n_keep][d][n]    -> out [n][n_keep][d]   (chains, draws, dims) */
int gsss_rows_to_components(const double *in_dev, double *out_dev, int64_t n, int32_t d, int device, void *stream);
int gsss_components_to_rows(const double *in_dev, double *out_dev, int64_t n, int32_t d, int device, void *stream);
int gsss_samples_to_chains(const double *in_dev, double *out_dev, int64_t n, int64_t n_keep, int32_t d, int device,
                           void *stream);

/* Minimal memory / stream helpers for callers without another HIP front-end (ctypes-only hosts). */
int gsss_malloc(void **out_dev, size_t bytes, int device);
int gsss_free(void *p_dev, int device);
int gsss_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, int device, void *stream);
int gsss_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, int device, void *stream);
int gsss_memset(void *dst_dev, int value, size_t bytes, int device, void *stream);
/* The return path of Sampler.sample (mcmc.py:55-77 returns an ndarray; its consumers read it on the host, scripts/curve_vMF.py:119-120)
 * at PCIe speed: page-locked host memory the device copies into directly (a copy into pageable memory is staged by the driver at
 * a sixth of the link's rate), and a device-to-host copy that is only ENQUEUED on `stream`, so that the copy of one block of chains
 * overlaps the kernel of the next (geosss_amd/mcmc.py sample).  The buffer must stay alive until the stream has drained. */
int gsss_malloc_host(void **out_host, size_t bytes, int device);
int gsss_free_host(void *p_host);
/* ... or the caller's own pages, page-locked where they lie (hipHostRegister): locking pages that were TOUCHED before costs 4 ms for
 * 2.4 GB, locking untouched ones 105 ms -- the page faults, taken one by one inside the call (gsss_malloc_host: 107 ms); a host with
 * several cores touches them in parallel first (11 ms on 16 threads; tools/microbench_pinning.py, geosss_amd/_pinned.py). */
int gsss_host_register(void *p_host, size_t bytes, int device);
int gsss_host_unregister(void *p_host);
int gsss_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, int device, void *stream);
int gsss_stream_synchronize(int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSSS_H */

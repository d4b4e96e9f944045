// gsss_curvespec.h -- curve-vMF targets at 4 <= d <= 256 (launch_fast_curve says from where): L lanes per chain, L speculative tries per batch,
// every try screened in single precision.
//
// The lane-per-chain kernels hold x, u and 2 NK coefficients of a chain in one lane: from d = 9 on that is more than
// 256 vector registers (one wavefront per SIMD, and at the 10^5-chain ensembles of the reference's curve experiments not
// even one wavefront for every SIMD).  The cooperative kernel of gsss_fast.h spreads a chain over 16 lanes but runs the
// shrinkage loop one try after the other on nine of them.  Here a chain owns a GROUP of L = 4 (d <= 64), 8 (d <= 128) or 16
// lanes, each lane 4 Q of its components (gsss_fast_curvespec.hip has the measured choice of L and Q per dimension):
//
//  1. The O(d) work of a step -- normals, projection, the dots with the knots, the state update -- is spread over the
//     group (CoopVec<L, 4 Q>: lane g holds the component quads g, g + L, ...; all sums are DPP reductions, after which
//     every lane of the group holds the same bits).
//  2. The bracket sequence of a step does not depend on any log-density (mcmc.py:395, 400: while tries are rejected,
//     theta_t and the shrunken bracket follow from the uniforms alone), so the group replays the recurrence for L tries,
//     lane t evaluates try t -- the whole curve, all NK - 1 segments, in SINGLE precision on the hardware transcendentals
//     with the rigorous margin of gsss_screen.h (Curve32) -- and a ballot picks the first try that is not certainly
//     rejected.  Certainly accepted: the step ends there.  Undecided: that lane alone takes the double-precision decision
//     (from the step's double-precision coefficients, parked in a few LDS words of the group), and the scan resumes.  The
//     accepted try is the one the sequential loop stops at.  7.2 tries per step (kappa = 800) are 1.05 batches at L = 16.
//  3. One or two Philox rounds per step: lane g draws the block of its normals; lanes without components draw block 0
//     (threshold and theta_0 uniforms) and the blocks of the first tries; the uniforms travel through a few LDS words.
//  4. a_i.x follows the recurrence a_i.x' = c a_i.x + s a_i.u between refreshes for L = 16 (as in coopfast_kernel);
//     the L = 4 kernel forms it from x at every step, so its chains do not depend on how the steps are split over
//     launches (bitwise, like the lane kernels it stands in for).
//
// Stream layout, arithmetic of the projection and of the restricted level: as in gsss_fast.h / gsss_spec64.h, so the
// chains agree with the other fast kernels to rounding and with the oracle on the Philox stream in every integer output.
// Semantics followed: geosss/mcmc.py:357-401, sphere.py:10-33, spherical_curve.py:10-32, 95-102, distributions.py:272-275.
#pragma once
#include "gsss_screen.h"

namespace gsss {

// Wavefronts per SIMD asked of the compiler (d <= 64 with <= 10 knots / larger shapes): the most the kernels reach WITHOUT spilling registers.
// Measured on MI355X (ms per 10^8 chain-steps at d = 10 / 50): 2 wavefronts 43.7 / 81.1, 3 wavefronts 34.7 / 66.1, 4 wavefronts
// with 28-38 registers spilled 29.7 / 60.8 -- but then the spill traffic reaches HBM (294 MB of scratch for 25 000 wavefronts
// do not fit the L2): 3.9 GB / 33 GB per launch against 0.17 / 0.48 GB of algorithmic bytes.  Three it is.
#ifndef GSSS_CS_WAVES
#define GSSS_CS_WAVES 3
#endif
#ifndef GSSS_CS_WAVES_BIG
#define GSSS_CS_WAVES_BIG 2
#endif
// Two or three component quads per lane, <= 10 knots: THREE wavefronts per SIMD with registers in scratch beat two without, while
// the resident wavefronts' scratch stays in or near the L2 (384 wavefronts per XCD x 64 lanes x the bytes below; 4 MB of L2 per XCD,
// the memory-side cache behind it).  Measured at 10^5 chains x 1000 steps, ms per launch (profiles/r04_ab_q2_three_waves.log):
//   <4, 2, 10>  d = 17 .. 32     128 B a lane   d = 24: 26.45 -> 21.65 (+22 %; with the tangent parked in LDS too, 52 B: 21.8 -- not taken)
//   <4, 3, 10>  d = 33 .. 48     224 B, tangent parked in LDS (264 B without: 25.8)   d = 40: 29.88 -> 25.46 (+17 %)
//   <8, 3, 10>  d = 65 .. 96     148 B, parked (244 B: 43.7)                            d = 80: 47.33 -> 40.73 (+16 %)
//   <16, 3, 10> d = 129 .. 192   200 B, parked (252 B: 79.2)                            d = 160: 86.49 -> 73.27 (+18 %)
// Four quads lose: <4, 4, 10> 320 B (parked) d = 50: 32.46 -> 37.41; <8, 4, 10> 328 B d = 120: 54.45 -> 64.0; <16, 4, 10> 308 B
// d = 200: 94.96 -> 110.6 -- they stay at two wavefronts, nothing spilled, the tangent in registers.
// The 17-knot builds (curves of 11 .. 17 knots) likewise, one and two quads per lane (164 .. 208 B of scratch): 16 knots, 10^5 chains
// x 1000 steps, <4,1,17> d = 10 34.5 -> 28.3 ms, <4,2,17> d = 24 36.7 -> 30.5, <16,2,17> d = 100 91.8 -> 77.0 (+19 .. 22 %); three quads
// (<4,3,17>, 300 B) lose 6 % and stay at two.
#ifndef GSSS_CS_WAVES_K17
#define GSSS_CS_WAVES_K17 3
#endif
#ifndef GSSS_CS_WAVES_Q2
#define GSSS_CS_WAVES_Q2 3
#endif
#ifndef GSSS_CS_WAVES_Q3
#define GSSS_CS_WAVES_Q3 3
#endif
#ifndef GSSS_CS_PARK_FROM_Q
#define GSSS_CS_PARK_FROM_Q 99  // (A/B builds: component quads per lane from which every 10-knot build parks the tangent in LDS while the tries run)
#endif
#ifndef GSSS_CS_PACKED_BELOW
#define GSSS_CS_PACKED_BELOW 64  // lanes per chain below which the screen evaluates two segments per packed instruction (Curve32<.., PACKED>):
                                 // every group size (sixteen lanes: packed with preloaded constants lost 1 %, packed without them and
                                 // with the copy of the loop for full curves gains 1.4 %: d = 200 96.24 -> 94.9 ms)
#endif
#ifndef GSSS_CS_PRELOAD_Q
#define GSSS_CS_PRELOAD_Q 2  // component quads per lane from which the try evaluation reads all segment constants in one go -- in the
                             // sixteen-lane builds only: the four- and eight-lane ones evaluate segments in pairs and take the copy of the loop
                             // for full curves instead, which preloading would make spill (round 4: d = 24 27.79 -> 26.58 ms, d = 50 33.90 -> 32.62)
#endif
#ifndef GSSS_CS_KNOT_PIPE
#define GSSS_CS_KNOT_PIPE 1  // knot rows read from LDS one row ahead of their products (more than four components per lane)
#endif
#ifndef GSSS_CS_KNOT_PIPE_Q3
#define GSSS_CS_KNOT_PIPE_Q3 0  // (A/B: 1 = the pipeline in the three-quad builds at three wavefronts per SIMD too, as before round 5)
#endif
#ifndef GSSS_CS_KNOT_BARRIER
#define GSSS_CS_KNOT_BARRIER 2  // knots between scheduling barriers in the knot-dot loop
#endif

template <int L, int NK>
__host__ __device__ constexpr int curvespec_scratch_doubles()
{
    return 2 + 4 * L + 2 * NK;  // U_threshold, U_theta0 | a ring of 8 L try words (4 L doubles of a replayed stream) | the step's double-precision coefficients
}
// The tangent u rests in LDS while the tries run where the registers would not hold it beside the try loop's: the 17-knot
// builds with more than four components per lane.  (Round 3: the 10-knot builds kept parking it at two wavefronts per SIMD, where
// 256 registers are to be had -- without, <4, 4, 10> takes 254 and <16, 4, 10> 246, nothing spilled: d = 50 35.3 -> 34.9 ms,
// d = 200 99.4 -> 97.4 ms per 10^8 chain-steps.)
// (HEAVY: the replay and statistics builds carry more state and keep parking it.)
template <int Q, int NK, bool HEAVY>
__host__ __device__ constexpr bool curvespec_parks_u()
{
    return Q >= 2 && (NK > 10 || HEAVY || Q >= GSSS_CS_PARK_FROM_Q || (Q == 3 && GSSS_CS_WAVES_Q3 >= 3));
}
// (R) the tail slot of the tangent rests in LDS beside the quads' -- but for sixteen-lane groups: with its 2 KB the <16, 3, 10, +1>
// workgroup needs 55.8 KB of LDS, TWO workgroups per CU under a kernel built for three wavefronts per SIMD (the first measurement of
// that build: 95.1 -> 106.7 ms at d = 200); without, 53.7 KB: three
template <int L, int R>
__host__ __device__ constexpr bool curvespec_parks_tail()
{
    return R == 1 && L < 16;  // (two tail slots stay in registers everywhere: parked, <4, 3, 10, +2> is at the edge of three workgroups per CU)
}
template <int L, int Q, int NK, bool STATS>
__host__ __device__ constexpr int curvespec_waves()
{
    if (STATS) return GSSS_CS_WAVES_BIG;
    if (NK > 10) return Q <= 2 ? GSSS_CS_WAVES_K17 : GSSS_CS_WAVES_BIG;
    return Q == 1 ? GSSS_CS_WAVES : (Q == 2 ? GSSS_CS_WAVES_Q2 : (Q == 3 ? GSSS_CS_WAVES_Q3 : GSSS_CS_WAVES_BIG));
}
template <int L, int Q, int NK, bool HEAVY, int R = 0>
__host__ __device__ constexpr size_t curvespec_lds_doubles()
{
    return (size_t)NK * (4 * Q * L + R * L) + 4 * (size_t)(NK - 1) + 2 * (size_t)(NK - 1) +
           (size_t)curvespec_scratch_doubles<L, NK>() * (kBlock / L) + kTabLds + 2 +
           (curvespec_parks_u<Q, NK, HEAVY>() ? (size_t)(4 * Q + (curvespec_parks_tail<L, R>() ? R : 0)) * kBlock : 0);
}

// The all-double decision of one try (rare): threshold from x as the level of theta = 0 (mcmc.py:389, 397), FastCurve::level
// operation for operation, from the step's double-precision coefficients parked in the group's LDS words: coef[2 i] = a_i.x,
// coef[2 i + 1] = a_i.u.  (Inlined on purpose: as a called function it takes the tables and segment constants through generic
// pointers, and every table read of the kernel becomes a flat_load.)
// Taken by the GROUP (round 3; round 2: by the one lane of the try, segment after segment): lane g evaluates segments g, g + L, ... at theta = 0 and at theta, a DPP
// butterfly over the group picks the first segment of maximal clipped value for each -- (value, index) is totally ordered
// (larger value first, then smaller index), so the winner is the one the sequential scan keeps, and the comparison at the
// end is formed from the same bits.  An undecided try costs 3 rolled iterations (NK = 10, L = 4) instead of 9 with the other
// lanes of the wavefront idle: 0.03 undecided tries per chain-step at kappa = 800 were ~15 % of the d = 10 kernel.
// Every lane of the group calls it (same theta); every lane returns the same answer.
template <int L, int NK>
__device__ __forceinline__ bool curvespec_decide_group(const FastCurve<1, NK> &scl, const double *coef, const fm::Tables &tab, double theta,
                                                    double u_thr, int g)
{
    double sn, cs;
    fm::sincos_tab(theta, tab, sn, cs);
    double best0 = -INFINITY, dot0 = 0.0, best1 = -INFINITY, dot1 = 0.0;
    int idx0 = NK, idx1 = NK;
#pragma unroll 1
    for (int gg = g; gg + 1 < NK; gg += L) {
        const double cxa = coef[2 * gg], cua = coef[2 * gg + 1], cxb = coef[2 * gg + 2], cub = coef[2 * gg + 3];
        const double ay0 = fma(1.0, cxa, 0.0 * cua), by0 = fma(1.0, cxb, 0.0 * cub);
        const double ay1 = fma(cs, cxa, sn * cua), by1 = fma(cs, cxb, sn * cub);
        double xc, xy;
        scl.segment_value(gg, ay0, by0, xc, xy);
        if (xc > best0) {
            best0 = xc;
            dot0 = xy;
            idx0 = gg;
        }
        scl.segment_value(gg, ay1, by1, xc, xy);
        if (xc > best1) {
            best1 = xc;
            dot1 = xy;
            idx1 = gg;
        }
    }
    auto merge = [](double &xc, double &xy, int &idx, double oxc, double oxy, int oidx) {
        const bool take = oxc > xc || (oxc == xc && oidx < idx);
        xc = take ? oxc : xc;
        xy = take ? oxy : xy;
        idx = take ? oidx : idx;
    };
#define GSSS_CS_STAGE(CTRL)                                                                             \
    do {                                                                                                \
        const double o0 = dpp_move<CTRL>(best0), p0 = dpp_move<CTRL>(dot0), o1 = dpp_move<CTRL>(best1), \
                     p1 = dpp_move<CTRL>(dot1);                                                         \
        const int i0 = dpp_move<CTRL>(idx0), i1 = dpp_move<CTRL>(idx1);                                 \
        merge(best0, dot0, idx0, o0, p0, i0);                                                           \
        merge(best1, dot1, idx1, o1, p1, i1);                                                           \
    } while (0)
    GSSS_CS_STAGE(kDppXor1);
    if (L >= 4) GSSS_CS_STAGE(kDppXor2);
    if (L >= 8) GSSS_CS_STAGE(kDppHalfMirror);
    if (L >= 16) GSSS_CS_STAGE(kDppMirror);
#undef GSSS_CS_STAGE
    return scl.kappa * dot1 > scl.kappa * dot0 + fm::log_fast(u_thr);
}

// R = 1, 2 (round 5): one or two more components per lane behind its Q quads -- the "uneven" layouts for dimensions that miss a
// whole number of quads per lane by a few components (gsss_fast_curvespec.hip lists the shapes and what each gains): d = 50 as
// <4, 3, +1> and d = 200 as <16, 3, +1> (BASELINE cfg4), d = 24 as <4, 1, +2>.  The next quad cost those shapes the register class
// below (four quads: the third wavefront per SIMD -- 254 / 246 registers, 15 % of the SIMD cycles idle with nobody to switch to:
// VERDICT r4).  Components 4 Q L + R g + r (r < R) are lane g's tail slots: scalars beside the register arrays of the quads; rows in
// LDS are [4 Q L main | R L tail] doubles in component order; their normals come from Philox block 1 + Q L + R g / 4, word pair
// (R g / 2) & 1 (the block of their quad, as in every other kernel: the stream does not know the layout).
template <int L, int Q, int NK, bool REPLAY, bool STATS = false, int R = 0>
__global__ void __launch_bounds__(kBlock, (curvespec_waves<L, Q, NK, STATS>())) curvespec_kernel(TargetBlock tb, RunBlock a)
{
    static_assert(R == 0 || ((R == 1 || R == 2) && !STATS), "tail slots: one or two components a lane, plain and replay builds");
    constexpr int RS = R > 0 ? R : 1;            // (array extents)
    using V = CoopVec<L, 4 * Q>;
    using Scalar = FastCurve<1, NK>;  // its segment(): the double-precision restricted level
    constexpr int DMAIN = V::DPAD;               // components in the lanes' quads
    constexpr int DPAD = DMAIN + R * L;          // a knot row in LDS
    constexpr int N = V::N;
    constexpr int kRing = 4 * L;      // doubles
    constexpr int kRing32 = 8 * L;    // the same words as 32-bit try words (philox-v3: a try's uniform is one word of the stream)
    constexpr int kScratch = curvespec_scratch_doubles<L, NK>();
    // a_i.x by recurrence between refreshes (every kCoefRefresh global steps and at the first step of a launch or slice) for
    // sixteen-lane groups; L = 4 / 8 form it from x at every step, so that a run split over launches gives the same bits.
    // Round 4 measured the recurrence for them too (10 dots and group sums less a step: 80 of the 1878 vector instructions of a
    // wavefront-step at d = 10, 200 of 2736 at d = 50).  With `refresh` a run-time condition inside this one loop the compiler forms
    // the dots anyway and selects: no gain (20.79 -> 20.89 ms at d = 10).  With the loop compiled once per value of `refresh`:
    // 20.63 -> 20.17 ms (d = 10), 29.01 -> 27.94 (d = 24), 37.80 -> 36.73 (d = 50) -- but that restructuring ALONE cost the
    // two-wavefront builds 8-11 % (d = 50 34.83 -> 37.79 ms, d = 200 97.7 -> 108.7: another schedule of the same operations, 250
    // instead of 254 registers, more exposed LDS waits), so the net was a loss (profiles/r04_ab_recurrence.log).  Left as it was.
    constexpr bool kRecur = L >= 16;
    extern __shared__ __attribute__((aligned(16))) double lds[];

    const int d = tb.d, k = tb.k;
    // LDS: knots [NK][DPAD] | segments [NK-1][4] | the same in single precision | per group: scratch | tables
    // L = 16: the knot rows are staged slot-major, slot i of lane g at word i L + g, so that the 16 lanes of a group read
    // consecutive words (the four groups of a wavefront read the same ones: a broadcast).  In component order a lane's quads lie
    // 32 bytes apart, the compiler reads them with ds_read2_b64, and lanes g and g + 8 of every 64-bit access meet in the same
    // banks: measured at d = 200, 13 conflict cycles per LDS instruction and 151 ms per launch against 109 ms with this layout.
    // (L = 4 and 8 span at most 256 bytes per group and have no conflicts; the cooperative kernels of the other targets read
    // their rows with ds_read_b128, eight lanes a cycle, which the 32-byte stride suits: slot-major made them slower.)
    constexpr bool kSlotMajor = L >= 16;
    if (kSlotMajor) {
        for (int j = threadIdx.x; j < NK * DPAD; j += kBlock) {
            const int r = j / DPAD, w = j - r * DPAD;
            // (round 3: slot PAIRS -- slots 2 p, 2 p + 1 of lane g at words 2 (p L + g), 2 (p L + g) + 1: the same conflict-free
            // broadcast pattern with 16-byte reads, half the LDS instructions)
            const int c = w < DMAIN ? V::comp((w / 2) % L, 2 * ((w / 2) / L) + (w & 1)) : w;  // (tail slots: lane g's at word DMAIN + g)
            lds[j] = (r < k && c < d) ? tb.blob[(size_t)r * d + c] : 0.0;
        }
    } else {
        lds_fill(lds, k, DPAD, tb.blob, d);
        for (int i = threadIdx.x + k * DPAD; i < NK * DPAD; i += kBlock) lds[i] = 0.0;
    }
    double *sg = lds + (size_t)NK * DPAD;
    for (int i = threadIdx.x; i < NK - 1; i += kBlock) {  // blob: theta, cos, sin, sin + 1e-10
        const bool real = i < k - 1;
        sg[4 * i + 0] = real ? tb.blob[(size_t)k * d + 4 * i + 1] : 1.0;
        sg[4 * i + 1] = real ? tb.blob[(size_t)k * d + 4 * i + 2] : 0.0;
        sg[4 * i + 2] = real ? 1.0 / tb.blob[(size_t)k * d + 4 * i + 3] : 0.0;
        sg[4 * i + 3] = 0.0;
    }
    __syncthreads();
    Curve32<NK, (L < GSSS_CS_PACKED_BELOW)> c32;
    c32.stage(reinterpret_cast<float4 *>(sg + 4 * (NK - 1)), sg, k - 1, tb.kappa);
    double *scr = sg + 6 * (size_t)(NK - 1) + (size_t)kScratch * (threadIdx.x / L);
    double *ring = scr + 2;
    uint32_t *ring32 = reinterpret_cast<uint32_t *>(ring);
    double *coef = ring + kRing;  // [NK][2]: a_i.x, a_i.u of the step (for the double-precision decisions), 16-byte aligned
    const fm::Tables tab = stage_tables(sg + 6 * (size_t)(NK - 1) + (size_t)kScratch * (kBlock / L));
    // Q >= 2 (d > 64): u is only needed again when the chain moves; its 8 Q registers are lent to the try loop meanwhile
    // (slots 2 p, 2 p + 1 of thread t at [p][t][2]: 16-byte accesses, conflict-free)
    constexpr bool kParkU = curvespec_parks_u<Q, NK, REPLAY || STATS>();
    constexpr bool kParkT = kParkU && curvespec_parks_tail<L, R>();
    double2 *upark = reinterpret_cast<double2 *>(sg + 6 * (size_t)(NK - 1) + (size_t)kScratch * (kBlock / L) + kTabLds + 2) + threadIdx.x;
    double *upark_t = sg + 6 * (size_t)(NK - 1) + (size_t)kScratch * (kBlock / L) + kTabLds + 2 + (size_t)N * kBlock + threadIdx.x;  // (R: tail slot r at [r * kBlock])
    Scalar sc;
    sc.knots = lds;
    sc.seg = sg;
    sc.kappa = tb.kappa;
    sc.nseg = k - 1;
    __syncthreads();

    const int lane = threadIdx.x % 64;
    const int g = lane % L;              // lane of the group
    const int base = lane - g;           // first lane of the group in the wavefront
    const int64_t n = a.n_chains;
    const bool shrink = a.sampler == GSSS_SHRINK;
    const int nq = (d + 3) >> 2;         // Philox blocks of the normals
    const uint32_t try_base = 1u + (uint32_t)nq;
    const int max_tries = a.max_tries;

    // This workgroup's work: the whole launch of the chunk of kBlock / L chains it was launched for -- or, sliced (a.sched,
    // SliceSched in gsss_device.h), the (chunk, step slice) of the ticket it draws.
    __shared__ uint32_t sched_word[4];
    const bool sliced = a.sched != nullptr;
    uint32_t chunk = blockIdx.x;
    int32_t s_begin = 0, len = (int32_t)a.n_steps;  // (fast mode: < 2^31 steps per launch)
    bool timed_out = false;
    if (sliced && !SliceSched::take(a, kBlock / L, sched_word, chunk, s_begin, len, timed_out)) return;  // (one ticket per workgroup: never)
    const uint64_t step0 = a.step_offset + (uint64_t)s_begin;  // global id of the item's first step
    const int64_t c_raw = (int64_t)chunk * (kBlock / L) + threadIdx.x / L;
    const bool active = c_raw < n;
    const int64_t c = active ? c_raw : n - 1;

    double x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int cc = V::comp(g, i);
        x[i] = (cc < d) ? a.state[(size_t)cc * n + c] : 0.0;
    }
    const int ct = DMAIN + R * g;        // (R) the first component of this lane's tail slots: DMAIN + R g + r
    double xt[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) xt[r] = (r < R && ct + r < d) ? a.state[(size_t)(ct + r) * n + c] : 0.0;

    PhiloxDraws<V> dr;
    dr.init(a, c, d);
    const double *rp = REPLAY ? a.replay + (size_t)c * a.replay_stride : nullptr;
    int64_t cursor = 0;
    int err = 0;
    // (kRecur: a_i . x is carried from step to step in the group's LDS words coef[2 i], advanced by lane 0)
    float lvl_c = 0.0f, e_c = 0.0f;  // single-precision level of the accepted point and its error bound, carried to the next step
    int64_t n_try = 0;
    // (a chain that is alive has made every step so far: retained rows follow from the slice's first step)
    int32_t steps_done = 0, until_keep = (int32_t)a.thin - s_begin % (int32_t)a.thin, row_out = s_begin / (int32_t)a.thin;
    bool alive = active && len > 0;
    if (sliced) {
        if (SliceSched::dead(a, kBlock / L)[c] != 0) alive = false;  // stopped with an error flag in an earlier slice of this launch
        if (timed_out && alive) {              // cannot happen (SliceSched::take); never compute from a state that is not there
            err |= GSSS_CHAIN_MAX_TRIES | GSSS_CHAIN_COUNTER_SATURATED;
            alive = false;
        }
    }

    auto wave_sync = [&]() {  // LDS words written by some lanes of the wavefront, read by others
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // one Philox round of "extra" blocks: extra index e = 0 is block 0, e >= 1 the tries' block e - 1 -- the words of tries
    // 4 (e - 1) .. 4 (e - 1) + 3 (philox-v3), stored as they are (one 16-byte store; a round of L blocks is 4 L tries: with the L a batch
    // may still hold, 5 L of the ring's 8 L words)
    auto publish_extra = [&](int e, const uint32_t (&w)[4]) {
        if (e == 0) {
            scr[0] = u53(w[0], w[1]);
            scr[1] = u53(w[2], w[3]);
        } else if (e > 0) {
            *reinterpret_cast<uint4 *>(ring32 + ((4 * (e - 1)) & (kRing32 - 1))) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };

    for (int32_t s = 0; s < len; ++s) {
        if (!__any(alive)) break;
        // The target's constants never change, so the compiler would hoist their LDS loads out of this loop and keep them
        // in registers for the whole launch: the base is made opaque once per step.
        unsigned opaque = 0u;
        asm volatile("" : "+s"(opaque));
        // (an offset, so that the pointer stays an LDS pointer: ds_read, not flat_load)
        const double *knots = lds + opaque;
        Scalar scl = sc;
        scl.seg = knots + (size_t)NK * DPAD;
        Curve32<NK, (L < GSSS_CS_PACKED_BELOW)> c32s = c32;
        c32s.seg32 = reinterpret_cast<const float4 *>(knots + (size_t)NK * DPAD + 4 * (NK - 1));
        // ---------------- draws of the step
        double u[N], u_thr, u_th0;
        double ut[RS];  // (R) the tangent's tail slots
#pragma unroll
        for (int r = 0; r < RS; ++r) ut[r] = 0.0;
        int pref = 0;  // tries whose uniforms are in the ring: [.., pref)
        if (REPLAY) {
            const bool ok = cursor + d <= a.replay_stride;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int cc = V::comp(g, i);
                u[i] = (cc < d) ? (ok ? rp[cursor + cc] : 0.5) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) ut[r] = (ct + r < d) ? (ok ? rp[cursor + ct + r] : 0.5) : 0.0;
            if (ok)
                cursor += d;
            else {
                cursor = a.replay_stride;
                if (alive) err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            }
            auto take = [&]() -> double {
                if (cursor >= a.replay_stride) {
                    if (alive) err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
                    return 0.5;
                }
                return rp[cursor++];
            };
            u_thr = take();
            u_th0 = shrink ? take() : 0.0;
        } else {
            dr.begin_step(step0 + (uint64_t)s);
#pragma unroll
            for (int iq = 0; iq < Q; ++iq) {
                const int quad = g + L * iq;
                const int e = quad - nq;  // lanes past the normals draw block 0 and the first tries' blocks
                uint32_t w[4];
                dr.words(e < 0 ? 1u + (uint32_t)quad : (e == 0 ? 0u : try_base + (uint32_t)(e - 1)), w);
                double z0, z1, z2, z3;
                box_muller32(w[0], w[1], tab, z0, z1);
                box_muller32(w[2], w[3], tab, z2, z3);
                const int c0 = 4 * quad;
                u[4 * iq + 0] = (c0 < d) ? z0 : 0.0;
                u[4 * iq + 1] = (c0 + 1 < d) ? z1 : 0.0;
                u[4 * iq + 2] = (c0 + 2 < d) ? z2 : 0.0;
                u[4 * iq + 3] = (c0 + 3 < d) ? z3 : 0.0;
                publish_extra(e, w);
                if (Q > 1) {  // one round at a time, finished: left alone the compiler sinks the last operations of every
                              // round's Box-Muller pairs to their first use and keeps ~18 intermediates per round alive
                    asm volatile("" : "+v"(u[4 * iq + 0]), "+v"(u[4 * iq + 1]), "+v"(u[4 * iq + 2]), "+v"(u[4 * iq + 3]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            pref = 4 * (L * Q - nq - 1);
            if (R) {
                // the tail round: a lane that holds tail components draws the block of their quad -- components DMAIN + R g + r sit in
                // word pair (R g / 2) & 1 of block 1 + Q L + R g / 4: ONE Box-Muller pair, of which the lane keeps one half (R = 1) or
                // both (R = 2); lanes past the tail draw block 0 and the first tries' blocks, as the lanes past the normals do in the
                // even layouts
                const int T = d - DMAIN;              // 1 .. R L tail components (the host picks this build for no other d)
                const int tl = (T + R - 1) / R;       // lanes that hold some
                const int e = g - tl;
                uint32_t w[4];
                dr.words(e < 0 ? 1u + (uint32_t)(Q * L + ((R * g) >> 2)) : (e == 0 ? 0u : try_base + (uint32_t)(e - 1)), w);
                const bool second_pair = (((R * g) >> 1) & 1) != 0;
                double za, zb;
                box_muller32(second_pair ? w[2] : w[0], second_pair ? w[3] : w[1], tab, za, zb);
                if (R == 1) {
                    ut[0] = (e < 0 && ct < d) ? ((g & 1) ? zb : za) : 0.0;
                } else {
                    ut[0] = (e < 0 && ct < d) ? za : 0.0;
                    ut[RS - 1] = (e < 0 && ct + 1 < d) ? zb : 0.0;
                }
                publish_extra(e, w);
                pref = 4 * (L - tl - 1);
            }
            if (pref < 0) {  // every lane holds normals: block 0 and the first tries take a round of their own
                uint32_t w[4];
                dr.words(g == 0 ? 0u : try_base + (uint32_t)(g - 1), w);
                publish_extra(g, w);
                pref = 4 * (L - 1);
            }
            u_thr = u_th0 = 0.0;  // (published to the group's LDS words: read behind the wave_sync that follows the knot dots)
        }
        // ---------------- u = spherical_projection(z, x)   (sphere.py:29-33)
        auto gdot = [&](const double (&p)[N], const double (&pt)[RS], const double (&q)[N], const double (&qt)[RS]) -> double {  // (R = 0: vdot<V>, its bits)
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) acc = fma(p[i], q[i], acc);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = fma(pt[r], qt[r], acc);
            return group_sum<L>(acc);
        };
        const double xx = gdot(x, xt, x, xt);
        const double rnx = inv_norm(xx);
        const double cz = gdot(u, ut, x, xt) * rnx;
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = fma(-cz * rnx, x[i], u[i]);  // w = z - (z . n) n
#pragma unroll
        for (int r = 0; r < R; ++r) ut[r] = fma(-cz * rnx, xt[r], ut[r]);
        // ---------------- a_r . u = (a_r . w) / |w|, a_r . x; single-precision pack; the doubles parked for decide()
        const bool refresh = !kRecur || s == 0 || ((step0 + (uint64_t)s) % kCoefRefresh) == 0;
        float q[Curve32<NK, (L < GSSS_CS_PACKED_BELOW)>::kFloats];
        {
            double pw = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) pw = fma(u[i], u[i], pw);
#pragma unroll
            for (int r = 0; r < R; ++r) pw = fma(ut[r], ut[r], pw);
            const double rnw = inv_norm(group_sum<L>(pw));
            // Round 4: the knot rows software-pipelined -- row r + 1 is read from LDS while row r is multiplied (N more registers, the
            // same products in the same order, the same bits).  The two-wavefront builds wait on LDS with nobody to switch to (vector
            // pipes 0.83 busy): d = 24 28.60 -> 28.20 ms, d = 50 34.87 -> 33.96, d = 200 97.8 -> 95.96 per 10^8 chain-steps
            // (profiles/r04_ab_knot_pipeline.log; scheduling barriers every 1 / 3 / 5 knots or none: 0 .. +3 %, left at 2).
            // Round 5: NOT in the three-quad builds at three wavefronts per SIMD -- there a stalled wavefront has two others to
            // switch to, and the row held ahead costs 48 .. 88 B of scratch a lane: at d = 200 the resident wavefronts' scratch falls
            // from 5.9 to 3.7 MB per XCD, under the 4 MB of its L2 (34.2 -> 8.3 GB of HBM traffic per launch, 80.3 -> 74.1 ms); d = 50
            // 4.19 -> 2.25 GB, 26.0 -> 25.4 ms; d = 40 / 100 / 160 +0.5 .. 1.3 %, d = 80 -0.8 % (profiles/r05_ab_knot_pipe_q3.log).
            // Two quads (three wavefronts, d = 24: -1 % without) and four quads (two wavefronts) keep it.
            constexpr bool kPipe = GSSS_CS_KNOT_PIPE && Q >= 2 && !(Q == 3 && curvespec_waves<L, Q, NK, STATS>() >= 3 && !REPLAY && !GSSS_CS_KNOT_PIPE_Q3);
            if constexpr (kPipe) {
                // (sixteen lanes x sixteen components: half a row ahead -- a whole one spills 12 bytes a lane at two wavefronts per SIMD)
                constexpr int kAhead = (L >= 16 && Q >= 4) ? N / 4 : N / 2;
                auto load_part = [&](int r, double2 (&dst)[N / 2], int from, int to) {
                    const double *row = knots + (size_t)r * DPAD;
#pragma unroll
                    for (int h = 0; h < N / 2; ++h)
                        if (h >= from && h < to)
                            dst[h] = kSlotMajor ? *reinterpret_cast<const double2 *>(row + 2 * (h * L + g))
                                                : *reinterpret_cast<const double2 *>(row + V::comp(g, 2 * h));
                };
                double2 cur_row[N / 2], nxt_row[N / 2];
                load_part(0, cur_row, 0, N / 2);
                double cur_t[RS], nxt_t[RS];  // (R) the rows' tail slots, one row ahead like the rest
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    cur_t[r] = r < R ? knots[ct + r] : 0.0;
                    nxt_t[r] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < NK; ++r) {
                    if (r + 1 < NK) load_part(r + 1, nxt_row, 0, kAhead);
                    if (R && r + 1 < NK) {
#pragma unroll
                        for (int t = 0; t < R; ++t) nxt_t[t] = knots[(size_t)(r + 1) * DPAD + ct + t];
                    }
                    double pu = 0.0, px = 0.0;
#pragma unroll
                    for (int i = 0; i < N; i += 2) {
                        const double2 kv = cur_row[i / 2];
                        pu = fma(kv.x, u[i], pu);
                        if (refresh) px = fma(kv.x, x[i], px);
                        pu = fma(kv.y, u[i + 1], pu);
                        if (refresh) px = fma(kv.y, x[i + 1], px);
                    }
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        pu = fma(cur_t[t], ut[t], pu);
                        if (refresh) px = fma(cur_t[t], xt[t], px);
                        cur_t[t] = nxt_t[t];
                    }
                    const double au = group_sum<L>(pu) * rnw;
                    const double axr = refresh ? group_sum<L>(px) : coef[2 * r];
                    q[r] = (float)axr;
                    q[NK + r] = (float)au;
                    if (refresh)
                        *reinterpret_cast<double2 *>(coef + 2 * r) = make_double2(axr, au);
                    else
                        coef[2 * r + 1] = au;
#pragma unroll
                    for (int h = 0; h < kAhead; ++h) cur_row[h] = nxt_row[h];
                    __builtin_amdgcn_sched_barrier(0);  // one row (or half of one) ahead, no further
                    if (kAhead < N / 2 && r + 1 < NK) load_part(r + 1, cur_row, kAhead, N / 2);
                }
            } else {
#pragma unroll
                for (int r = 0; r < NK; ++r) {
                    const double *row = knots + (size_t)r * DPAD;
                    double pu = 0.0, px = 0.0;
                    if constexpr (kSlotMajor) {
#pragma unroll
                        for (int i = 0; i < N; i += 2) {
                            const double2 kv = *reinterpret_cast<const double2 *>(row + 2 * ((i / 2) * L + g));
                            pu = fma(kv.x, u[i], pu);
                            if (refresh) px = fma(kv.x, x[i], px);
                            pu = fma(kv.y, u[i + 1], pu);
                            if (refresh) px = fma(kv.y, x[i + 1], px);
                        }
                    } else {
                        // component order: a lane's quads are 32 contiguous bytes, read as two 16-byte halves (ds_read_b128: 4 LDS
                        // cycles per wavefront; left to itself the compiler, which cannot see the alignment behind the opaque
                        // offset, reads them with ds_read2_b64: 8 cycles for the same bytes)
#pragma unroll
                        for (int i = 0; i < N; i += 2) {
                            const double2 kv = *reinterpret_cast<const double2 *>(row + V::comp(g, i));
                            pu = fma(kv.x, u[i], pu);
                            if (refresh) px = fma(kv.x, x[i], px);
                            pu = fma(kv.y, u[i + 1], pu);
                            if (refresh) px = fma(kv.y, x[i + 1], px);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < R; ++t) {  // the row's tail slots
                        const double kt = row[ct + t];
                        pu = fma(kt, ut[t], pu);
                        if (refresh) px = fma(kt, xt[t], px);
                    }
                    const double au = group_sum<L>(pu) * rnw;
                    const double axr = refresh ? group_sum<L>(px) : coef[2 * r];
                    q[r] = (float)axr;
                    q[NK + r] = (float)au;
                    // parked for decide() and the recurrence.  EVERY lane of the group stores the pair: they hold the same bits after
                    // the group sums, same address, same value -- an LDS instruction instead of the 4 selects per knot that routed
                    // the pair to one owner lane (round 2: 40 of the step's vector instructions)
                    if (refresh)
                        *reinterpret_cast<double2 *>(coef + 2 * r) = make_double2(axr, au);  // (one 16-byte store)
                    else
                        coef[2 * r + 1] = au;
                    // two knots at a time: left alone the scheduler runs all NK reduction chains side by side (4 NK registers)
                    if (GSSS_CS_KNOT_BARRIER > 0 && r % GSSS_CS_KNOT_BARRIER == GSSS_CS_KNOT_BARRIER - 1) __builtin_amdgcn_sched_barrier(0);  // (five knots at a time: measured, no change)
                }
            }
#pragma unroll
            for (int i = 0; i < N; i += 2) {
                u[i] *= rnw;
                u[i + 1] *= rnw;
                if (kParkU) upark[(size_t)(i / 2) * kBlock] = make_double2(u[i], u[i + 1]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ut[r] *= rnw;
                if (kParkT) upark_t[(size_t)r * kBlock] = ut[r];
            }
        }
        wave_sync();  // (one synchronisation for the step's uniforms, the parked coefficients and u)
        if (!REPLAY) {
            u_thr = scr[0];
            u_th0 = scr[1];
        }
        // threshold and margin (Curve32).  The level of x in single precision: the value the accepted try of the previous
        // step was screened with (it is within that step's evaluation error of the double-precision level of y(theta_T), which
        // IS the level of x now: the same operations on the same coefficients under the recurrence, and within ~1e-15 of
        // it where a_i.x is formed from x again) -- or the screen's own evaluation at theta = 0 after a refresh.
        float e_eval;
        bool finite;
        if ((refresh && kRecur) || s == 0) {
            finite = c32s.finish32(u_thr, q);
            e_eval = c32s.eval_error(q);
        } else {
            finite = c32s.finish32_carried(u_thr, q, lvl_c, e_c + 1.0e-12f, e_eval);
        }
        e_c = e_eval;
        if (a.screen == 2) q[2 * NK + 1] = INFINITY;  // verification: every try is left to the double-precision decision
        if (alive && (!finite || !(xx < INFINITY))) {  // (a NaN state: the clipped level of the curve swallows it, so x.x is looked at)
            err |= GSSS_CHAIN_NONFINITE;
            alive = false;
        }
        if (REPLAY && (err & GSSS_CHAIN_REPLAY_EXHAUSTED)) alive = false;
        double lo, hi;
        if (shrink) {
            hi = kTwoPi * u_th0;  // mcmc.py:391-392
            lo = hi - kTwoPi;
        } else {
            lo = 0.0;             // mcmc.py:367
            hi = kTwoPi;
        }
        auto decide = [&](double theta) -> bool { return curvespec_decide_group<L, NK>(scl, coef, tab, theta, u_thr, g); };

        // ---------------- batches of L speculative tries
        bool done = !alive, accepted = false;
        double th_acc = 0.0;
        for (int t_base = 0; __any(!done); t_base += L) {
            if (!done && t_base >= max_tries) {
                n_try += max_tries;
                err |= GSSS_CHAIN_MAX_TRIES;
                done = true;
            }
            int valid = L;  // replay: draws available for this batch
            if (REPLAY) {
                valid = 0;
#pragma unroll
                for (int qq = 0; qq < L; ++qq) {
                    const bool have = cursor + qq < a.replay_stride;
                    ring[qq] = have ? rp[cursor + qq] : 0.5;
                    valid += have ? 1 : 0;
                }
                wave_sync();
            } else {
                while (t_base + L > pref) {  // (wave-uniform) another round: lane g draws the tries' block pref / 4 + g (four tries a block)
                    uint32_t w[4];
                    dr.words(try_base + (uint32_t)(pref >> 2) + (uint32_t)g, w);
                    publish_extra(1 + (pref >> 2) + g, w);
                    pref += 4 * L;
                    wave_sync();
                }
            }
            double my_theta = 0.0;
            if (shrink) {  // (two loops: the sampler is the same for the whole launch, a select per end and try otherwise)
#pragma unroll
                for (int qq = 0; qq < L; ++qq) {
                    const double u_try = REPLAY ? ring[qq] : try_uniform(ring32[(t_base + qq) & (kRing32 - 1)]);
                    const double theta = fma(hi - lo, u_try, lo);  // mcmc.py:395
                    if (g == qq) my_theta = theta;
                    const bool neg = theta < 0.0;               // mcmc.py:400, assuming try qq is rejected
                    lo = neg ? theta : lo;
                    hi = neg ? hi : theta;
                }
            } else {
#pragma unroll
                for (int qq = 0; qq < L; ++qq) {
                    const double u_try = REPLAY ? ring[qq] : try_uniform(ring32[(t_base + qq) & (kRing32 - 1)]);
                    const double theta = fma(hi - lo, u_try, lo);  // mcmc.py:367, 395: the bracket stays (0, 2 pi)
                    if (g == qq) my_theta = theta;
                }
            }
            if (REPLAY) wave_sync();  // the ring is rewritten by the next batch
            float s32, c32f;
            sincos_rev32(my_theta, s32, c32f);
            const bool mine = !done && g < valid && t_base + g < max_tries;
            // (Q >= 2: two wavefronts per SIMD and registers to spare -- the segments' constants are read from LDS in one go, one
            // round trip per evaluation instead of one per segment: 38.3 -> 37.0 ms at d = 50, 105.4 -> 103.8 at d = 200)
            const float my_b = c32s.template best32<(Q >= GSSS_CS_PRELOAD_Q && L >= GSSS_CS_PACKED_BELOW)>(q, c32f, s32);
            const float gap = my_b - q[2 * NK];
            int verdict = mine ? (gap < -q[2 * NK + 1] ? -1 : (gap > q[2 * NK + 1] ? 1 : 0)) : -1;
            // first try of the group that is not certainly rejected; an undecided one is decided in double precision by its lane
            // (the whole group takes that decision: curvespec_decide_group)
            int T = L;
            for (;;) {
                const unsigned long long open = __ballot(verdict >= 0);
                const unsigned gm = (unsigned)(open >> base) & ((1u << L) - 1u);
                T = gm ? __builtin_ctz(gm) : L;
                const bool need = g == T && verdict == 0;
                const unsigned long long needs = __ballot(need);
                if (needs == 0ull) break;
                if (((unsigned)(needs >> base) & ((1u << L) - 1u)) != 0u) {  // this group's first open try is undecided
                    const bool acc = decide(__shfl(my_theta, base + T));
                    if (need) verdict = acc ? 1 : -1;
                }
            }
            const double th_T = __shfl(my_theta, base + (T < L ? T : 0));
            const float b_T = __shfl(my_b, base + (T < L ? T : 0));
            if (!done) {
                if (T < L) {
                    accepted = true;
                    done = true;
                    th_acc = th_T;
                    lvl_c = b_T;
                    n_try += t_base + T + 1;
                    if (REPLAY) cursor += T + 1;
                } else {
                    if (REPLAY) {
                        cursor += valid;
                        if (valid < L) {
                            n_try += t_base + valid;
                            err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
                            done = true;
                        }
                    }
                    if (!done && max_tries - t_base <= L) {
                        n_try += max_tries;
                        err |= GSSS_CHAIN_MAX_TRIES;
                        done = true;
                    }
                }
            }
        }
        if (alive && !accepted) alive = false;  // (an error flag is set)
        // ---------------- move (mcmc.py:396)
        double sn, cs;
        fm::sincos_tab(th_acc, tab, sn, cs);
        if (alive) {
            if constexpr (kParkU) {
#pragma unroll
                for (int i = 0; i < N; i += 2) {
                    const double2 up = upark[(size_t)(i / 2) * kBlock];
                    x[i] = fma(sn, up.x, cs * x[i]);
                    x[i + 1] = fma(sn, up.y, cs * x[i + 1]);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) xt[r] = fma(sn, kParkT ? upark_t[(size_t)r * kBlock] : ut[r], cs * xt[r]);
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) x[i] = fma(sn, u[i], cs * x[i]);
#pragma unroll
                for (int r = 0; r < R; ++r) xt[r] = fma(sn, ut[r], cs * xt[r]);
            }
            if (kRecur && g == 0) {
#pragma unroll
                for (int r = 0; r < NK; ++r) coef[2 * r] = fma(cs, coef[2 * r], sn * coef[2 * r + 1]);  // a . x' = c a.x + s a.u
            }
            ++steps_done;
            if ((a.samples != nullptr || STATS) && --until_keep == 0) {
                until_keep = (int32_t)a.thin;
                if (a.samples != nullptr) {
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const int cc = V::comp(g, i);
                        if (cc < d) a.samples[sample_index(a, row_out, cc, d, c)] = x[i];
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (ct + r < d) a.samples[sample_index(a, row_out, ct + r, d, c)] = xt[r];
                }
                if constexpr (STATS) stats_update_group<V>(a, c, g, d, x);  // (`alive` is the same in every lane of a group)
                ++row_out;
            }
        }
        if (kRecur) wave_sync();  // coef is rewritten by the next step
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int cc = V::comp(g, i);
            if (cc < d) a.state[(size_t)cc * n + c] = x[i];
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ct + r < d) a.state[(size_t)(ct + r) * n + c] = xt[r];
        if (g == 0) {
            if (a.n_reject) a.n_reject[c] += n_try - steps_done;
            if (a.n_tries) a.n_tries[c] += n_try;
            if (a.err && err) a.err[c] |= err;
            if (a.sched != nullptr && err) SliceSched::dead(a, kBlock / L)[c] = 1;  // (every error flag of this kernel stops the chain)
        }
    }
    if (a.sched != nullptr) SliceSched::publish(a, sched_word);
}

template <int L, int Q, int NK, int R = 0>
int do_curvespec(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    if (rb.rng_state != nullptr) {
        set_error("in fast mode the numpy stream is served by the one-wavefront-per-chain kernel only; use GSSS_MODE_EXACT");
        return GSSS_E_UNSUPPORTED;
    }
    if (tb.d > (4 * Q + R) * L || (R && tb.d <= 4 * Q * L) || tb.k > NK || tb.k < 2) {
        set_error("curvespec kernel <%d, %d, %d, +%d> cannot hold d=%d, %d knots", L, Q, NK, R, tb.d, tb.k);
        return GSSS_E_UNSUPPORTED;
    }
    const size_t lds = (replay || rb.stats != nullptr ? curvespec_lds_doubles<L, Q, NK, true, R>() : curvespec_lds_doubles<L, Q, NK, false, R>()) * sizeof(double);
    auto kern = replay ? curvespec_kernel<L, Q, NK, true, false, R> : curvespec_kernel<L, Q, NK, false, false, R>;
    if (rb.stats != nullptr) {  // running statistics: a build of its own (the plain kernel carries none of it)
        if (replay) {
            set_error("running statistics are not accumulated from a replayed stream by the group kernels");
            return GSSS_E_UNSUPPORTED;
        }
        if constexpr (R == 0) {
            kern = curvespec_kernel<L, Q, NK, false, true>;
        } else {
            set_error("the uneven group layouts carry no statistics build");  // (launch_curvespec never asks)
            return GSSS_E_UNSUPPORTED;
        }
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int64_t per_block = kBlock / L;
    const int64_t n_chunks = (rb.n_chains + per_block - 1) / per_block;
    // more chunks than the chip holds at once: a grid of the resident workgroups takes (chunk, step slice) tickets (SliceSched)
    const SlicePlan plan = plan_slices(kern, lds, rb, n_chunks, !replay, st);
    RunBlock rbl = rb;
    rbl.sched = plan.ws;
    rbl.slice_steps = plan.slice_steps;
    hipLaunchKernelGGL(kern, dim3((unsigned)plan.grid), dim3(kBlock), lds, st, tb, rbl);
    hipError_t e = hipGetLastError();
    if (plan.ws) (void)hipFreeAsync(plan.ws, st);
    if (e != hipSuccess) {
        set_error("curvespec kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

}  // namespace gsss

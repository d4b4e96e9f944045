// gsss_spec64.h -- curve-vMF targets at 64 < d <= 256: one chain per wavefront, speculative tries
// across the four 16-lane rows.
//
// The cooperative kernel of gsss_fast.h spreads the O(d) work of a step over the 64 lanes but runs the
// shrinkage loop on nine of them (one lane per segment of the curve), one try after the other.  Here:
//
//  1. The bracket sequence of a step does not depend on any log-density (mcmc.py:395, 400: while tries
//     are rejected, theta_t and the shrunken bracket follow from the uniforms alone).  Every lane replays
//     the recurrence for four tries, row r of the wavefront keeps theta_r, its first lanes evaluate the
//     segments of the curve for it, and a ballot picks the first accepted row -- the try the
//     sequential loop stops at, hence the same chain bit for bit.  7.1 tries per step (kappa = 800)
//     become 2.2 iterations.
//  2. One Philox round per step: lanes 0 .. ceil(d/4)-1 draw the blocks of the normals, the next eight
//     lanes the blocks of the first 16 tries, the next one block 0 (U_threshold, U_theta0).
//  3. The 1 + k group sums of a step (|w|^2 and the k knot dots with w) are reduced together:
//     two half-exchange stages (v_permlane32_swap / v_permlane16_swap: 1.5 instructions per value)
//     leave a quarter of the values in each row, one DPP row reduction finishes them
//     (3 k + 12 instead of 25 k instructions).
//  4. The level of the accepted point is carried to the next step (it IS the next step's level of x:
//     the same FMA on the same operands), so only log U is taken per step.
//
// Semantics followed: geosss/mcmc.py:357-401, sphere.py:10-33, spherical_curve.py:10-32, 95-102,
// distributions.py:272-275 (see FastCurve in gsss_fast.h for the trigonometry-free segment algebra).
#pragma once
#include "gsss_fast.h"

namespace gsss {

// half exchanges (gfx950): a' = [a_lo | b_lo], b' = [a_hi | b_hi] over the wave's halves (32) or over
// each half's two rows (16)
__device__ __forceinline__ void swap32(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap16(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// Sums over the 64 lanes of N values at once.  On return row r (lanes 16 r .. 16 r + 15) holds in
// out[j] the total of value r N/4 + j, the same bits in each of its 16 lanes.
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], double (&out)[N / 4])
{
    static_assert(N % 4 == 0, "values come in fours");
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {  // lanes < 32 keep value j, lanes >= 32 value j + N/2
        swap32(v[j], v[j + N / 2]);
        v[j] += v[j + N / 2];
    }
#pragma unroll
    for (int j = 0; j < N / 4; ++j) {  // even rows keep value j, odd rows value j + N/4
        swap16(v[j], v[j + N / 4]);
        v[j] += v[j + N / 4];
    }
#pragma unroll
    for (int j = 0; j < N / 4; ++j) out[j] = group_sum<16>(v[j]);
}

__device__ __forceinline__ double row_max16(double v)
{
    v = fmax(v, dpp_move<kDppXor1>(v));
    v = fmax(v, dpp_move<kDppXor2>(v));
    v = fmax(v, dpp_move<kDppHalfMirror>(v));
    v = fmax(v, dpp_move<kDppMirror>(v));
    return v;
}

constexpr int kSpecRows = 4;        // tries evaluated per iteration
constexpr int kSpecPrefetched = 16; // try uniforms drawn with the normals

// NV = number of values of the big reduction (>= k + 1, multiple of 4); k <= NV - 1 knots
template <int NV, bool REPLAY, bool STATS = false>
__global__ void __launch_bounds__(kBlock) curve64_kernel(TargetBlock tb, RunBlock a)
{
    using V = CoopVec<64, 4>;
    constexpr int DPAD = V::DPAD;  // 256
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = tb.d, k = tb.k, nseg = tb.k - 1;
    // LDS: knots [k][DPAD] | segment constants [nseg][4] | per wave: 16 try uniforms, U_thr, U_th0, pad, NV exchange slots
    lds_fill(lds, k, DPAD, tb.blob, d);
    double *sg = lds + (size_t)k * DPAD;
    for (int i = threadIdx.x; i < nseg; i += kBlock) {  // blob: theta, cos, sin, sin + 1e-10
        sg[4 * i + 0] = tb.blob[(size_t)k * d + 4 * i + 1];
        sg[4 * i + 1] = tb.blob[(size_t)k * d + 4 * i + 2];
        sg[4 * i + 2] = 1.0 / tb.blob[(size_t)k * d + 4 * i + 3];
        sg[4 * i + 3] = 0.0;
    }
    constexpr int kScr = kSpecPrefetched + 4 + NV;
    double *scr = sg + 4 * (size_t)nseg + (size_t)kScr * (threadIdx.x / 64);
    double *xch = scr + kSpecPrefetched + 4;
    const fm::Tables tab = stage_tables(sg + 4 * (size_t)nseg + (size_t)kScr * (kBlock / 64));
    __syncthreads();

    const int lane = threadIdx.x % 64;
    const int row = lane >> 4, col = lane & 15;
    const int64_t n = a.n_chains;
    const int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    if (c >= n) return;  // whole wavefront
    const bool shrink = a.sampler == GSSS_SHRINK;
    const int nq = (d + 3) >> 2;                     // Philox blocks of the normals = lanes that hold components
    const bool folded = nq + kSpecPrefetched / 2 + 1 <= 64;  // room for the tries' blocks and block 0 in the same round
    const uint32_t try_base = 1u + (uint32_t)nq;
    const double kappa = tb.kappa;

    // this lane's segment (rows replicate the assignment: lane col of every row evaluates segment col)
    const bool has_seg = col < nseg;
    const int seg = has_seg ? col : 0;
    const double ct = sg[4 * seg], st = sg[4 * seg + 1], rden = sg[4 * seg + 2];
    double ax0 = 0.0, ax1 = 0.0, au0 = 0.0, au1 = 0.0;  // a_seg . x, a_{seg+1} . x, same for u

    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = 4 * lane + i;
        x[i] = (cc < d) ? a.state[(size_t)cc * n + c] : 0.0;
    }
    const double *krow = lds + 4 * lane;  // component quad of this lane in knot row r: krow[r * DPAD + i]

    PhiloxDraws<V> dr;
    dr.init(a, c, d);
    const double *rp = REPLAY ? a.replay + (size_t)c * a.replay_stride : nullptr;
    int64_t cursor = 0;
    int err = 0;
    double lvl = 0.0;
    int64_t n_try = 0, steps_done = 0, until_keep = a.thin, row_out = 0;

    // kappa * (y . nearest point of the curve) for y(theta) of THIS row; row-uniform result
    auto level_row = [&](double cs, double sn) -> double {
        const double ay = fma(cs, ax0, sn * au0), by = fma(cs, ax1, sn * au1);
        const double A = ay * st;
        const double B = fma(-ay, ct, by);
        const double h2 = fma(A, A, B * B);
        const double rh = h2 > 0.0 ? rsqrt(h2) : 0.0;
        const double inner = fma(fma(st, A, -ct * B), ay, B * by) * rh;
        const bool at_a = B < 0.0 || (B == 0.0 && A >= 0.0);
        const bool at_b = A * rh < ct;
        const double num = at_a ? st * ay : (at_b ? st * by : inner);
        const double xy = num * rden;
        const double xc = has_seg ? fmin(fmax(xy, -1.0), 1.0) : -INFINITY;
        const double mx = row_max16(xc);
        // nearest segment = first one of maximal clipped y.near (np.argmin of the distances, spherical_curve.py:97-102).
        // Unless the maximum sits on the clip, the winner's y.near IS the maximum.
        if (__builtin_expect(__any(mx >= 1.0 || mx <= -1.0 || !(mx == mx)), 0)) {
            double bxc = xc, bxy = xy;
            int idx = col;
            auto step = [&](double oxc, double oxy, int oidx) {
                const bool take = oxc > bxc || (oxc == bxc && oidx < idx);
                bxc = take ? oxc : bxc;
                bxy = take ? oxy : bxy;
                idx = take ? oidx : idx;
            };
            step(dpp_move<kDppXor1>(bxc), dpp_move<kDppXor1>(bxy), dpp_move<kDppXor1>(idx));
            step(dpp_move<kDppXor2>(bxc), dpp_move<kDppXor2>(bxy), dpp_move<kDppXor2>(idx));
            step(dpp_move<kDppHalfMirror>(bxc), dpp_move<kDppHalfMirror>(bxy), dpp_move<kDppHalfMirror>(idx));
            step(dpp_move<kDppMirror>(bxc), dpp_move<kDppMirror>(bxy), dpp_move<kDppMirror>(idx));
            return kappa * bxy;
        }
        return kappa * mx;
    };

    // a_r . v for all knots r (+ v . v as value 0), reduced together; afterwards this lane's two
    // coefficients (knots seg and seg + 1) scaled by `scale(vv)`; returns v . v
    auto knot_dots = [&](const double (&v)[4], double &c0, double &c1, bool normalise) -> double {
        double part[NV];
        part[0] = fma(v[0], v[0], fma(v[1], v[1], fma(v[2], v[2], v[3] * v[3])));
#pragma unroll
        for (int r = 0; r < NV - 1; ++r) {
            if (r < k) {
                const double *kr = krow + (size_t)r * DPAD;
                part[r + 1] = fma(kr[0], v[0], fma(kr[1], v[1], fma(kr[2], v[2], kr[3] * v[3])));
            } else {
                part[r + 1] = 0.0;
            }
        }
        double red[NV / 4];
        wave_reduce_scatter<NV>(part, red);
        const double vv = lane_broadcast(red[0], 0);
        const double scale = normalise ? inv_norm(vv) : 1.0;
        // value row * NV/4 + j sits in red[j] of the lanes of `row`: lane col == j publishes it
#pragma unroll
        for (int j = 0; j < NV / 4; ++j)
            if (col == j) xch[row * (NV / 4) + j] = red[j] * scale;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        c0 = xch[1 + seg];
        c1 = xch[2 + seg];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return vv;
    };

    for (int64_t s = 0; s < a.n_steps && !err; ++s) {
        // ---------------- draws of the step
        double u[4], u_thr, u_th0;
        if (REPLAY) {
            const bool ok = cursor + d <= a.replay_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = 4 * lane + i;
                u[i] = (cc < d) ? (ok ? rp[cursor + cc] : 0.5) : 0.0;
            }
            if (ok)
                cursor += d;
            else {
                cursor = a.replay_stride;
                err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            }
            auto take = [&]() -> double {
                if (cursor >= a.replay_stride) {
                    err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
                    return 0.5;
                }
                return rp[cursor++];
            };
            u_thr = take();
            u_th0 = shrink ? take() : 0.0;
        } else {
            dr.begin_step(a.step_offset + (uint64_t)s);
            uint32_t w[4];
            {
                // lanes past the normals: 0..7 the tries' blocks, the rest block 0.  philox-v3: try i is word i % 4 of block
                // try_base + i / 4 -- lane t publishes tries 2 t, 2 t + 1 as before: words 2 (t & 1), 2 (t & 1) + 1 of block try_base + t / 2
                const int t = lane - nq;
                const uint32_t blk = lane < nq ? 1u + (uint32_t)lane
                                               : ((folded && t < kSpecPrefetched / 2) ? try_base + (uint32_t)(t >> 1) : 0u);
                dr.words(blk, w);
            }
            const double p0 = u53(w[0], w[1]), p1 = u53(w[2], w[3]);
            if (folded) {
                const int t = lane - nq;
                if (t >= 0 && t < kSpecPrefetched / 2) {
                    scr[2 * t] = try_uniform((t & 1) ? w[2] : w[0]);
                    scr[2 * t + 1] = try_uniform((t & 1) ? w[3] : w[1]);
                }
                if (t == kSpecPrefetched / 2) {
                    scr[kSpecPrefetched] = p0;
                    scr[kSpecPrefetched + 1] = p1;
                }
            }
            {   // normals 4 lane .. 4 lane + 3 (block 1 + lane), zeros past d
                double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
                box_muller32(w[0], w[1], tab, z0, z1);
                box_muller32(w[2], w[3], tab, z2, z3);
                const int c0 = 4 * lane;
                u[0] = (lane < nq && c0 < d) ? z0 : 0.0;
                u[1] = (lane < nq && c0 + 1 < d) ? z1 : 0.0;
                u[2] = (lane < nq && c0 + 2 < d) ? z2 : 0.0;
                u[3] = (lane < nq && c0 + 3 < d) ? z3 : 0.0;
            }
            if (!folded) {  // d > 220: the tries' blocks and block 0 take a round of their own
                uint32_t w2[4];
                dr.words(lane < kSpecPrefetched / 2 ? try_base + (uint32_t)(lane >> 1) : 0u, w2);
                const double q0 = u53(w2[0], w2[1]), q1 = u53(w2[2], w2[3]);
                if (lane < kSpecPrefetched / 2) {
                    scr[2 * lane] = try_uniform((lane & 1) ? w2[2] : w2[0]);
                    scr[2 * lane + 1] = try_uniform((lane & 1) ? w2[3] : w2[1]);
                }
                if (lane == kSpecPrefetched / 2) {
                    scr[kSpecPrefetched] = q0;
                    scr[kSpecPrefetched + 1] = q1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            u_thr = scr[kSpecPrefetched];
            u_th0 = scr[kSpecPrefetched + 1];
        }
        // ---------------- u = spherical_projection(z, x)   (sphere.py:29-33)
        double rnx, cz;
        bool x_ok;  // a NaN / Inf state: flagged (the curve's clipped level would swallow it)
        {
            double pxx = fma(x[0], x[0], fma(x[1], x[1], fma(x[2], x[2], x[3] * x[3])));
            double pzx = fma(u[0], x[0], fma(u[1], x[1], fma(u[2], x[2], u[3] * x[3])));
            swap32(pxx, pzx);                      // lanes < 32: x.x partials, lanes >= 32: z.x partials
            const double t = group_sum<16>(pxx + pzx);
            const double xx = lane_broadcast(t, 0) + lane_broadcast(t, 16);
            const double zx = lane_broadcast(t, 32) + lane_broadcast(t, 48);
            x_ok = xx < INFINITY;
            rnx = inv_norm(xx);
            cz = zx * rnx;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = fma(-cz * rnx, x[i], u[i]);  // w = z - (z . n) n
        const bool first = s == 0;
        const bool refresh = first || ((a.step_offset + (uint64_t)s) % kCoefRefresh) == 0;
        if (refresh) (void)knot_dots(x, ax0, ax1, false);
        {
            const double ww = knot_dots(u, au0, au1, true);  // a_r . u = (a_r . w) / |w|
            const double rnw = inv_norm(ww);
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] *= rnw;
        }
        const double lvl0 = refresh ? level_row(1.0, 0.0) : lvl;
        if (!(lvl0 > -INFINITY && lvl0 < INFINITY) || !x_ok) {
            err |= GSSS_CHAIN_NONFINITE;
            break;
        }
        const double thr = lvl0 + fm::log_fast(u_thr);  // mcmc.py:389
        double lo, hi;
        if (shrink) {
            hi = kTwoPi * u_th0;  // mcmc.py:391-392
            lo = hi - kTwoPi;
        } else {
            lo = 0.0;             // mcmc.py:367
            hi = kTwoPi;
        }
        // ---------------- batches of four speculative tries
        int t_base = 0;
        bool accepted = false;
        double acs = 1.0, asn = 0.0;
        for (;;) {
            if (t_base >= a.max_tries) {
                n_try += a.max_tries;
                err |= GSSS_CHAIN_MAX_TRIES;
                break;
            }
            double ut[kSpecRows];
            int valid = kSpecRows;  // replay: draws available for this batch
            if (REPLAY) {
                valid = 0;
#pragma unroll
                for (int q = 0; q < kSpecRows; ++q) {
                    const bool have = cursor + q < a.replay_stride;
                    ut[q] = have ? rp[cursor + q] : 0.5;
                    valid += have ? 1 : 0;
                }
            } else if (t_base < kSpecPrefetched) {
#pragma unroll
                for (int q = 0; q < kSpecRows; ++q) ut[q] = scr[t_base + q];
            } else {  // rare: past the prefetched tries, one more block per batch of four (every lane forms the same one)
                static_assert(kSpecRows == 4, "a batch is one block of the stream");
                uint32_t w2[4];
                dr.words(try_base + (uint32_t)(t_base >> 2), w2);
#pragma unroll
                for (int q = 0; q < kSpecRows; ++q) ut[q] = try_uniform(w2[q]);
            }
            double my_theta = 0.0;
#pragma unroll
            for (int q = 0; q < kSpecRows; ++q) {
                const double theta = fma(hi - lo, ut[q], lo);  // mcmc.py:395
                if (row == q) my_theta = theta;
                if (shrink) {                                  // mcmc.py:400, assuming try q is rejected
                    if (theta < 0.0)
                        lo = theta;
                    else
                        hi = theta;
                }
            }
            double sn, cs;
            fm::sincos_tab(my_theta, tab, sn, cs);
            const double my_lvl = level_row(cs, sn);
            const bool ok = row < valid && t_base + row < a.max_tries && my_lvl > thr;  // mcmc.py:397
            const unsigned long long mask = __ballot(ok);
            if (mask != 0ull) {
                const int T = (int)__builtin_ctzll(mask) >> 4;  // first accepted row = where the sequential loop stops
                acs = lane_broadcast_dyn(cs, 16 * T);
                asn = lane_broadcast_dyn(sn, 16 * T);
                lvl = lane_broadcast_dyn(my_lvl, 16 * T);
                n_try += t_base + T + 1;
                if (REPLAY) cursor += T + 1;
                accepted = true;
                break;
            }
            if (REPLAY) {
                cursor += valid;
                if (valid < kSpecRows) {
                    n_try += t_base + valid;
                    err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
                    break;
                }
            }
            if (a.max_tries - t_base <= kSpecRows) {
                n_try += a.max_tries;
                err |= GSSS_CHAIN_MAX_TRIES;
                break;
            }
            t_base += kSpecRows;
        }
        if (!accepted) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = fma(asn, u[i], acs * x[i]);  // mcmc.py:396
        ax0 = fma(acs, ax0, asn * au0);                                  // a . x' = c a.x + s a.u
        ax1 = fma(acs, ax1, asn * au1);
        ++steps_done;
        if ((a.samples != nullptr || STATS) && --until_keep == 0) {
            until_keep = a.thin;
            if (a.samples != nullptr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cc = 4 * lane + i;
                    if (cc < d) a.samples[sample_index(a, row_out, cc, d, c)] = x[i];
                }
            }
            if constexpr (STATS) stats_update_group<V>(a, c, lane, d, x);
            ++row_out;
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = 4 * lane + i;
        if (cc < d) a.state[(size_t)cc * n + c] = x[i];
    }
    if (lane == 0) {
        if (a.n_reject) a.n_reject[c] += n_try - steps_done;
        if (a.n_tries) a.n_tries[c] += n_try;
        if (a.err && err) a.err[c] |= err;
    }
}

template <int NV>
int do_curve64(const TargetBlock &tb, const RunBlock &rb, bool replay, hipStream_t st)
{
    if (rb.rng_state != nullptr) {
        set_error("in fast mode the numpy stream is served for lane-per-chain shapes only; use GSSS_MODE_EXACT");
        return GSSS_E_UNSUPPORTED;
    }
    const size_t lds = ((size_t)tb.k * 256 + 4 * (size_t)(tb.k - 1) + (size_t)(kSpecPrefetched + 4 + NV) * (kBlock / 64) + kTabLds) *
                       sizeof(double);
    auto kern = replay ? curve64_kernel<NV, true> : curve64_kernel<NV, false>;
    if (rb.stats != nullptr) {  // running statistics: a build of its own
        if (replay) {
            set_error("running statistics are not accumulated from a replayed stream by the one-wavefront curve kernel");
            return GSSS_E_UNSUPPORTED;
        }
        kern = curve64_kernel<NV, false, true>;
    }
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
        set_error("curve64 kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

}  // namespace gsss

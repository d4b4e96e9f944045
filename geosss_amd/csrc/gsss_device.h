// gsss_device.h -- device-side building blocks of the many-chain geodesic slice sampler.
// gfx950 (CDNA4) only: 64-wide wavefronts, FP64 VALU, LDS-staged target parameters.
//
// Reference semantics followed (paths relative to the geosss repository):
//   proposal + shrinkage loop   geosss/mcmc.py:382-401   (rejection variant :357-374)
//   spherical_projection        geosss/sphere.py:10-33
//   vMF mixture log_prob        geosss/distributions.py:156-157, 218-221
//   Bingham log_prob            geosss/distributions.py:84-86
//   curve-vMF log_prob          geosss/distributions.py:272-275, spherical_curve.py:10-32, 95-102
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gsss.h"
#include "gsss_math.h"

namespace gsss {

constexpr int kBlock = 256;             // threads per workgroup = 4 wavefronts
constexpr double kTwoPi = 6.283185307179586;  // 2*np.pi

// ------------------------------------------------------------------------------------------
// Kernel-argument blocks (plain data, passed by value)
// ------------------------------------------------------------------------------------------
struct TargetBlock {
    const double *blob;  // device parameter blob, layout per kind (see gsss_capi.hip: build_blob)
    int32_t kind, d, k, dpad;
    double kappa;
    double scale;  // largest magnitude the log-density varies by along a circle (vMF: max kappa; Bingham: spectral spread;
                   // curve: kappa): the single-precision screen of the fast kernels is used while its error margin stays small
};

struct RunBlock {
    double *state;
    double *samples;
    int64_t *n_reject;
    int64_t *n_tries;
    int32_t *err;
    const double *replay;
    uint64_t *rng_state;  // numpy stream: [n_chains][4] PCG64 words (state_hi, state_lo, inc_hi, inc_lo)
    int64_t replay_stride;
    int64_t n_chains, n_steps, thin;
    uint64_t seed, chain_offset, step_offset;
    int32_t sampler, max_tries;
    int64_t keep_rows;  // > 0: samples are written chain-major [chain][keep_rows][d]; 0: [row][d][chain]
    int32_t spread;     // lane layouts: one chain per WAVEFRONT (small ensembles: no divergence between chains)
    int32_t one_per_lane;  // the lane kernels that park a second chain per lane: leave it out (mid-size ensembles: twice the
                           // workgroups while they all fit the chip at once; set by do_screened_run)
    int32_t stage_rows;    // lane kernels, one chain per lane, chain-major retained rows of 8 d bytes that are not whole 32-byte
                           // sectors: a lane holds rows back in LDS until their run ends on a sector boundary (gsss_screen.h)
    int32_t screen;     // fast mode: tries are screened in single precision where the target's kernel is built for it
                        // (2: verification -- the screen's verdicts are ignored by the kernels that can, see gsss.h)
    double *stats;             // NULL or [gsss_stats_rows][n_chains] running statistics of the retained series
    const double *stats_dirs;  // [2 + stats_modes][d]: projection w, hop direction h, mode directions
    int32_t stats_lags, stats_modes;
    int32_t stats_flags;       // GSSS_STATS_* (row layout of `stats`)
    int32_t stats_onchip;      // lane kernels, one chain per lane: 1 = a launch's statistics on chip (StatsLane: the accumulators and
                               // lag sums in registers, the ring of the last stats_lags projections in LDS); 0 = every draw
                               // read-modify-writes its rows in HBM (stats_update)
    // Slice scheduling (kernels that support it; NULL: one workgroup per chunk of chains runs the whole launch).  A launch whose
    // chunks do not fit the chip at once runs its last, partial round of workgroups on a nearly empty chip; sliced, the grid has
    // one workgroup per (chunk, step slice), every workgroup draws its item from a ticket counter, slice-major, and a chunk's
    // state travels from slice to slice through HBM (SliceSched below): the dispatcher refills a slot whenever a slice ends, and
    // the chip drains for one slice, not for a whole chunk.  Zeroed workspace: [0] ticket | [2 ..) progress[n_chunks] | dead[n_chains].
    uint32_t *sched;
    int32_t slice_steps;       // steps per slice; boundaries sit at multiples of it in GLOBAL step ids
    int32_t sched_first;       // chunks below this one are not sliced: their workgroups (blockIdx < sched_first) run the whole launch
                               // (the lane kernels slice only the last, partial round of their workgroups)
};

// ------------------------------------------------------------------------------------------
// Slice scheduler.  The workgroup that draws ticket t works on (slice t / n_chunks, chunk t % n_chunks); slice k of a chunk
// starts when slice k - 1 has been published.  Tickets, not blockIdx: the predecessor's ticket was drawn earlier, so its
// workgroup is resident or done whatever order the dispatcher starts workgroups in, and it waits for nothing drawn later -- the
// wait always ends.  In slice-major order the predecessor was drawn n_chunks tickets earlier, so with n_chunks > resident
// workgroups the wait is over before it begins; it is still a real acquire: the chunk's state, counters and statistics rows were
// written by another CU (MI355X_MICROARCH.md, inter-workgroup visibility: plain stores -> every wave's s_waitcnt vmcnt(0) ->
// barrier -> lane 0: agent release, s_waitcnt, relaxed flag store; consumer: relaxed poll -> agent acquire -> s_waitcnt vmcnt(0)
// -> barrier -> plain loads).  Slice boundaries are multiples of slice_steps in global step ids (the first slice of a launch
// runs up to the next one), a multiple of kCoefRefresh, so a kernel that refreshes carried quantities at those steps computes
// the same bits sliced or not.  A wait that does not end (cannot happen) gives up after ~4 s per predecessor and flags the chains.
// ------------------------------------------------------------------------------------------
struct SliceSched {
    // (no state of its own: everything follows from the launch arguments, so that nothing of it stays in registers while a
    // slice runs -- the kernels that use it have no scalar register to spare)
    __device__ static __forceinline__ uint32_t chunks(const RunBlock &a, int per_block) { return (uint32_t)((a.n_chains + per_block - 1) / per_block); }
    __device__ static __forceinline__ uint32_t *progress(const RunBlock &a) { return a.sched + 2; }
    __device__ static __forceinline__ int32_t *dead(const RunBlock &a, int per_block)
    {
        return reinterpret_cast<int32_t *>(a.sched + 2 + chunks(a, per_block));
    }
    // workgroup-uniform; false: no such item.  `word`: three LDS words of the workgroup (the ticket; chunk and slice rest there
    // until publish()).  s_begin, len: the slice's steps within the launch; timed_out: the predecessor never arrived.
    __device__ static __forceinline__ bool take(const RunBlock &a, int per_block, uint32_t *word, uint32_t &chunk, int32_t &s_begin,
                                                int32_t &len, bool &timed_out)
    {
        if (threadIdx.x == 0) *word = __hip_atomic_fetch_add(a.sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)*word);
        const uint32_t n_chunks = chunks(a, per_block) - (uint32_t)a.sched_first;  // the sliced ones
        const int32_t ss = a.slice_steps, n_steps = (int32_t)a.n_steps;
        const int32_t first_len = ss - (int32_t)(a.step_offset % (uint64_t)ss);
        const int32_t rest = n_steps - first_len;
        const uint32_t n_items = n_chunks * (uint32_t)(1 + (rest > 0 ? (rest + ss - 1) / ss : 0));
        if (t >= n_items) return false;
        const uint32_t slice = t / n_chunks;
        chunk = (uint32_t)a.sched_first + (t - slice * n_chunks);
        if (threadIdx.x == 0) {
            word[1] = chunk;
            word[2] = slice;
        }
        s_begin = slice == 0 ? 0 : first_len + (int32_t)(slice - 1) * ss;
        int32_t s_end = first_len + (int32_t)slice * ss;
        s_end = s_end < n_steps ? s_end : n_steps;
        len = s_end - s_begin;
        timed_out = false;
        if (slice > 0) {
            __syncthreads();
            if (threadIdx.x == 0) {
                // Slice k of a chunk can start when its k predecessors have run one after the other, so the give-up limit grows
                // with k (~4 s per predecessor at ~1 us a poll: three orders of magnitude above a slice's own time; the host keeps a
                // launch to <= kMaxSlices slices, so the longest legitimate wait is a few hundred slice times).  It only exists so
                // that a lost predecessor ends in error flags instead of a hung GPU; it is not a scheduling decision.
                uint32_t ok = 1;
                uint64_t spins = 0;
                // (capped at ~30 s: the legitimate wait is at most kMaxSlices slice times of milliseconds, and a slice that gave up
                // publishes its progress like any other, so its successors do not wait out limits of their own: ADVICE r4)
                const uint64_t limit = (uint64_t)(slice + 1u < 8u ? slice + 1u : 8u) << 22;
                while (__hip_atomic_load(progress(a) + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < slice) {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > limit) {
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *word = ok;
            }
            __syncthreads();
            timed_out = __builtin_amdgcn_readfirstlane((int)*word) == 0;
        }
        return true;
    }
    // after the chunk's stores: make them visible, then publish the slice.  kWrittenThrough: EVERY byte the next slice reads was
    // stored by hand_over() below (write-through), so no release is needed -- the release writes back ALL dirty lines of this
    // XCD's L2, among them the half-written lines of retained rows that lanes of the lane kernels fill at their own pace
    // (measured on the README workload, a third of the chains sliced: 2 x FETCH + WRITE 690 MB per launch with the release).
    template <bool kWrittenThrough = false>
    __device__ static __forceinline__ void publish(const RunBlock &a, const uint32_t *word)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!kWrittenThrough) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(progress(a) + word[1], word[2] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // a store of handed-over state that needs no release behind it: agent scope, i.e. written through this XCD's L2
    template <class T>
    __device__ static __forceinline__ void hand_over(T *p, T v)
    {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

constexpr size_t slice_sched_bytes(int64_t n_chunks, int64_t n_chains) { return 4 * (size_t)(2 + n_chunks + n_chains); }

// Host side: should this launch be sliced, and on how many workgroups?  Sliced when the chunks do not fit the chip at once (a
// last, partial round would otherwise run on a nearly empty chip: measured 15 % of the curve kernels' time at 10^5 chains) and
// there are at least two slices.  The workspace is allocated, zeroed and freed in stream order; a failed allocation falls back
// to one workgroup per chunk.  GSSS_SLICE_STEPS: steps per slice (a multiple of 64; 0 turns slicing off).
struct SlicePlan {
    uint32_t *ws = nullptr;
    int64_t grid = 0;
    int32_t slice_steps = 0;
};
// A chunk's slices run one after the other, each waiting for its predecessor's hand-over: a launch is cut into at most
// kMaxSlices of them (long launches through the C ABI get longer slices: n_steps = 10^6 -> 15 680 steps per slice), so the
// serial chain a workgroup can wait on stays short whatever the caller asks for.  A multiple of 64 (kCoefRefresh).
constexpr int kMaxSlices = 64;
inline int slice_length(int64_t n_steps)
{
    const char *env = getenv("GSSS_SLICE_STEPS");  // (read per launch: tests switch it)
    // (measured at 10^5 chains x 1000 steps, d = 10 / 50 / 200: 64 -> 24.1 / 40.0 / 107.3 ms, 128 -> 23.8 / 39.7 / 106.1)
    const int env_val = env ? atoi(env) : 128;
    if (env_val <= 0) return 0;
    int64_t steps = ((int64_t)env_val + 63) / 64 * 64;
    const int64_t floor_steps = ((n_steps + kMaxSlices - 1) / kMaxSlices + 63) / 64 * 64;
    if (steps < floor_steps) steps = floor_steps;
    return steps < 0x40000000ll ? (int)steps : 0;
}
// hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs of the current device, cached per (kernel, LDS bytes): the query costs
// tens of microseconds and next(sampler) loops launch once per step
int64_t resident_workgroups(const void *kern, size_t lds_bytes, int *per_cu_out = nullptr);  // gsss_capi.hip; 0: the query failed
void slice_fallback_note(const char *why);                                                     // GSSS_DEBUG_OCCUPANCY: say so on stderr
// what the calling thread's last gsss_run launched (gsss_last_launch): grid 0 = a kernel that does not plan slices
struct LaunchInfo {
    int64_t grid;
    int32_t slice_steps;
    double sliced_fraction;  // share of the chunks of chains that ran sliced (0 unsliced, 1 every chunk)
};
LaunchInfo &last_launch();  // gsss_capi.hip, thread local
template <class Kern>
inline SlicePlan plan_slices(Kern kern, size_t lds_bytes, const RunBlock &rb, int64_t n_chunks, bool allowed, hipStream_t st)
{
    SlicePlan p;
    p.grid = n_chunks;
    last_launch() = LaunchInfo{n_chunks, 0, 0.0};
    if (!allowed) return p;
    const int env_steps = slice_length(rb.n_steps);
    if (env_steps <= 0 || rb.n_steps < 2 * (int64_t)env_steps) return p;
    const int64_t resident = resident_workgroups(reinterpret_cast<const void *>(kern), lds_bytes);
    if (resident < 1) return p;
    const int64_t n_slices = 2 + rb.n_steps / env_steps;
    if (n_chunks <= resident || n_chunks * n_slices >= 0x7FFFFFFFll || rb.n_chains >= 0x7FFFFFFFll) return p;
    const size_t bytes = slice_sched_bytes(n_chunks, rb.n_chains);
    void *ws = nullptr;
    if (hipMallocAsync(&ws, bytes, st) != hipSuccess || ws == nullptr) {
        (void)hipGetLastError();
        slice_fallback_note("hipMallocAsync of the slice workspace failed: launching unsliced");
        return p;
    }
    if (hipMemsetAsync(ws, 0, bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFreeAsync(ws, st);
        return p;
    }
    const int64_t first_len = env_steps - (int64_t)(rb.step_offset % (uint64_t)env_steps), rest = rb.n_steps - first_len;
    p.ws = static_cast<uint32_t *>(ws);
    p.grid = n_chunks * (1 + (rest > 0 ? (rest + env_steps - 1) / env_steps : 0));  // = SliceSched::take's n_items
    p.slice_steps = env_steps;
    last_launch() = LaunchInfo{p.grid, p.slice_steps, 1.0};
    return p;
}

// The lane kernels (two chains per lane, a workgroup's unit of time is a lane's pair of chains) slice only their LAST, PARTIAL
// round of workgroups, and only when it is small: n_chunks = k resident + rem with rem <= 3/4 resident -- the first k resident
// workgroups run the whole launch as ever (blockIdx < first), the rem chunks behind them are cut into slices that fill the
// chip when the full rounds end.  (A slice of a lane kernel ends when its slowest lane does: ~10 % of lane tail at 128 steps,
// paid on the partial round only.  At 10^6 chains: K = 10 mixture and compact Bingham 2.54 rounds of 768 -> sliced; headline 1.53 rounds of 1280 -> sliced.)
template <class Kern>
inline SlicePlan plan_partial_round(Kern kern, size_t lds_bytes, const RunBlock &rb, int64_t n_chunks, bool allowed, hipStream_t st,
                                    int32_t &first_out)
{
    SlicePlan p;
    p.grid = n_chunks;
    first_out = 0;
    last_launch() = LaunchInfo{n_chunks, 0, 0.0};
    if (!allowed) return p;
    const int env_steps = slice_length(rb.n_steps);
    if (env_steps <= 0 || rb.n_steps < 2 * (int64_t)env_steps) return p;
    const int64_t resident = resident_workgroups(reinterpret_cast<const void *>(kern), lds_bytes);
    if (resident < 1) return p;
    const int64_t rem = n_chunks % resident, first = n_chunks - rem;
    if (n_chunks <= resident || rem == 0 || 4 * rem > 3 * resident || rb.n_chains >= 0x7FFFFFFFll) return p;
    const int64_t first_len = env_steps - (int64_t)(rb.step_offset % (uint64_t)env_steps), rest = rb.n_steps - first_len;
    const int64_t n_slices = 1 + (rest > 0 ? (rest + env_steps - 1) / env_steps : 0);
    if (rem * n_slices >= 0x7FFFFFFFll) return p;
    const size_t bytes = slice_sched_bytes(n_chunks, rb.n_chains);
    void *ws = nullptr;
    if (hipMallocAsync(&ws, bytes, st) != hipSuccess || ws == nullptr) {
        (void)hipGetLastError();
        slice_fallback_note("hipMallocAsync of the slice workspace failed: launching unsliced");
        return p;
    }
    if (hipMemsetAsync(ws, 0, bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFreeAsync(ws, st);
        return p;
    }
    p.ws = static_cast<uint32_t *>(ws);
    p.grid = first + rem * n_slices;
    p.slice_steps = env_steps;
    first_out = (int32_t)first;
    last_launch() = LaunchInfo{p.grid, p.slice_steps, (double)rem / (double)n_chunks};
    return p;
}

// address of component j of retained row `row` of chain c
__device__ __forceinline__ size_t sample_index(const RunBlock &a, int64_t row, int j, int d, int64_t c)
{
    return a.keep_rows > 0 ? ((size_t)c * a.keep_rows + row) * d + j : ((size_t)row * d + j) * a.n_chains + c;
}

// ------------------------------------------------------------------------------------------
// Running statistics of the retained series (every thin-th state), accumulated per chain in HBM while the chain
// is sampled, so that diagnostics need no stored draws (include/gsss.h: gsss_stats_rows, GSSS_STATS_*).
// Definitions follow the reference's post-hoc estimators on a stored series s_0 .. s_{n-1}:
//   sum s, sum s s^T                           (moments)
//   sum_t distance(s_t, s_{t-1})               sphere.py:64-68 on consecutive draws
//   #{t : sign(s_t.h) != sign(s_{t-1}.h)}      scripts/bingham.py:23-25 (hopping frequency)
//   #{t : argmax_k s_t.mode_k = k}             scripts/vMF_diagnostics.py:335-342 (mode occupancy)
//   p_t = s_t.w: sum p, sum p^2, sum_t p_t p_{t-l} (l = 1 .. L), first and last L values
//                                              -> utils.acf (utils.py:96-110) exactly, IAT / n_eff by :119-134
// Row r of chain c lives at stats[r * n_chains + c].  D components of the state in registers.
// ------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void stats_update(const RunBlock &a, int64_t c, const double (&x)[D])
{
    const size_t n = (size_t)a.n_chains;
    double *s = a.stats + c;
    const int K = a.stats_modes, L = a.stats_lags;
    const bool second = !(a.stats_flags & GSSS_STATS_NO_SECOND_MOMENT);
    const int T = second ? D * (D + 1) / 2 : 0;
    const int r_prev = 1, r_sum = 1 + D, r_xx = 1 + 2 * D, r_dist = r_xx + T, r_hop = r_dist + 1, r_mode = r_hop + 1;
    const int r_p = r_mode + K, r_lag = r_p + 2, r_ring = r_lag + L, r_head = r_ring + L;
    const double *w = a.stats_dirs, *h = a.stats_dirs + D, *modes = a.stats_dirs + 2 * D;
    const int64_t cnt = (int64_t)s[0];
    double p = 0.0, xh = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        p = fma(x[j], w[j], p);
        xh = fma(x[j], h[j], xh);
    }
    if (cnt > 0) {
        double dot = 0.0, ph = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double pj = s[(size_t)(r_prev + j) * n];
            dot = fma(pj, x[j], dot);
            ph = fma(pj, h[j], ph);
        }
        s[(size_t)r_dist * n] += acos(fmin(fmax(dot, -1.0), 1.0));
        const int sa = (xh > 0.0) - (xh < 0.0), sb = (ph > 0.0) - (ph < 0.0);  // np.sign
        if (sa != sb) s[(size_t)r_hop * n] += 1.0;
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        s[(size_t)(r_prev + i) * n] = x[i];
        s[(size_t)(r_sum + i) * n] += x[i];
        if (second) {
#pragma unroll
            for (int j = i; j < D; ++j) s[(size_t)(r_xx + t++) * n] += x[i] * x[j];
        }
    }
    if (K > 0) {
        int best = 0;
        double bv = -INFINITY;
        for (int k = 0; k < K; ++k) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) v = fma(x[j], modes[k * D + j], v);
            if (v > bv) {  // first maximum, like np.argmax
                bv = v;
                best = k;
            }
        }
        s[(size_t)(r_mode + best) * n] += 1.0;
    }
    s[(size_t)r_p * n] += p;
    s[(size_t)(r_p + 1) * n] += p * p;
    if (L > 0) {
        const int64_t lmax = cnt < L ? cnt : L;
        for (int64_t l = 1; l <= lmax; ++l) s[(size_t)(r_lag + l - 1) * n] += p * s[(size_t)(r_ring + (cnt - l) % L) * n];
        s[(size_t)(r_ring + cnt % L) * n] = p;
        if (cnt < L) s[(size_t)(r_head + cnt) * n] = p;
    }
    s[0] = (double)(cnt + 1);
}

// ------------------------------------------------------------------------------------------
// Round 5: the same statistics with a LAUNCH's working set on chip (lane kernels, one chain per lane).  stats_update above
// read-modify-writes every row a draw touches in HBM, the L lag sums in a loop of dependent round trips: with one wavefront per
// SIMD (a statistics build's occupancy) a launch that accumulates every state is 28 x slower than the plain kernel -- 0.78 of the
// HBM peak at 32 lags (6.2 TB/s, profiles/r05_stats_thin1_lags32_summary.md), latency-bound beyond (64 lags: 160 us a draw).
// Here a lane takes its chain's accumulators INTO REGISTERS when it takes the chain up -- the moments, the L <= kStatsMaxLags lag
// sums (128 of the 512 registers a one-wavefront build may use) -- and the ring of the last L projections into LDS (where a second
// chain would be parked), and writes them back when it lets go of the chain (end of the launch or slice).  A draw then costs L
// LDS reads and L multiply-adds and no HBM traffic but the mode count.
// THE SAME BITS as stats_update: every accumulator sees the same operations in the same order, so the rows in HBM are the
// ones stats_update leaves -- any mix of the two over launches, slices and kernel families continues the same series.
// ------------------------------------------------------------------------------------------
constexpr int kStatsMaxLags = 64;

template <int D>
struct StatsLane {
    static constexpr int kT = D * (D + 1) / 2;
    static constexpr bool kXXRegs = D <= 6;  // the second moments in registers (21 at d = 6); beyond, read-modify-written per draw as before
    int64_t cnt;
    int pos;            // ring slot of the NEXT draw: cnt % L (the slot its row in HBM has: draw t rests in ring row t % L)
    double prev[D], sum[D], xx[kXXRegs ? kT : 1];
    double dist, hop, psum, psq;
    double lag[kStatsMaxLags];
    double *ring;       // LDS, this lane's slot j at ring[j * kBlock]

    struct Rows {
        int r_prev, r_sum, r_xx, r_dist, r_hop, r_mode, r_p, r_lag, r_ring, r_head, K, L, T;
        bool second;
    };
    __device__ __forceinline__ Rows rows(const RunBlock &a) const
    {
        Rows r;
        r.K = a.stats_modes;
        r.L = a.stats_lags;
        r.second = !(a.stats_flags & GSSS_STATS_NO_SECOND_MOMENT);
        r.T = r.second ? kT : 0;
        r.r_prev = 1;
        r.r_sum = 1 + D;
        r.r_xx = 1 + 2 * D;
        r.r_dist = r.r_xx + r.T;
        r.r_hop = r.r_dist + 1;
        r.r_mode = r.r_hop + 1;
        r.r_p = r.r_mode + r.K;
        r.r_lag = r.r_p + 2;
        r.r_ring = r.r_lag + r.L;
        r.r_head = r.r_ring + r.L;
        return r;
    }

    __device__ __forceinline__ void load(const RunBlock &a, int64_t c, double *ring_lds)
    {
        const size_t n = (size_t)a.n_chains;
        const double *s = a.stats + c;
        const Rows r = rows(a);
        ring = ring_lds;
        cnt = (int64_t)s[0];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            prev[i] = s[(size_t)(r.r_prev + i) * n];
            sum[i] = s[(size_t)(r.r_sum + i) * n];
        }
        if (kXXRegs && r.second) {
#pragma unroll
            for (int t = 0; t < kT; ++t) xx[t] = s[(size_t)(r.r_xx + t) * n];
        }
        dist = s[(size_t)r.r_dist * n];
        hop = s[(size_t)r.r_hop * n];
        psum = s[(size_t)r.r_p * n];
        psq = s[(size_t)(r.r_p + 1) * n];
        pos = r.L > 0 ? (int)(cnt % (int64_t)r.L) : 0;
#pragma unroll
        for (int l = 0; l < kStatsMaxLags; ++l) {
            const bool in = l < r.L;
            lag[l] = in ? s[(size_t)(r.r_lag + (in ? l : 0)) * n] : 0.0;
            if (in) ring[(size_t)l * kBlock] = s[(size_t)(r.r_ring + l) * n];
        }
    }

    __device__ __forceinline__ void draw(const RunBlock &a, int64_t c, const double (&x)[D])
    {
        const size_t n = (size_t)a.n_chains;
        double *s = a.stats + c;
        const Rows r = rows(a);
        const double *w = a.stats_dirs, *h = a.stats_dirs + D, *modes = a.stats_dirs + 2 * D;
        double p = 0.0, xh = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            p = fma(x[j], w[j], p);
            xh = fma(x[j], h[j], xh);
        }
        if (cnt > 0) {
            double dot = 0.0, ph = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                dot = fma(prev[j], x[j], dot);
                ph = fma(prev[j], h[j], ph);
            }
            dist += acos(fmin(fmax(dot, -1.0), 1.0));
            const int sa = (xh > 0.0) - (xh < 0.0), sb = (ph > 0.0) - (ph < 0.0);  // np.sign
            if (sa != sb) hop += 1.0;
        }
        int t = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            prev[i] = x[i];
            sum[i] += x[i];
            if (r.second) {
#pragma unroll
                for (int j = i; j < D; ++j) {
                    if (kXXRegs)
                        xx[t] += x[i] * x[j];
                    else
                        s[(size_t)(r.r_xx + t) * n] += x[i] * x[j];
                    ++t;
                }
            }
        }
        if (r.K > 0) {
            int best = 0;
            double bv = -INFINITY;
            for (int k = 0; k < r.K; ++k) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) v = fma(x[j], modes[k * D + j], v);
                if (v > bv) {  // first maximum, like np.argmax
                    bv = v;
                    best = k;
                }
            }
            s[(size_t)(r.r_mode + best) * n] += 1.0;
        }
        psum += p;
        psq += p * p;
        if (r.L > 0) {
            const int lm = cnt < (int64_t)r.L ? (int)cnt : r.L;  // lags this draw has a partner for
            int si = pos;                                        // draw cnt - l rests in slot (pos - l) mod L
#pragma unroll
            for (int l = 1; l <= kStatsMaxLags; ++l) {
                si = si == 0 ? r.L - 1 : si - 1;
                if (l <= lm) lag[l - 1] += p * ring[(size_t)si * kBlock];
            }
            ring[(size_t)pos * kBlock] = p;
            if (cnt < r.L) s[(size_t)(r.r_head + cnt) * n] = p;
            pos = pos + 1 == r.L ? 0 : pos + 1;
        }
        ++cnt;
    }

    // everything back where stats_update keeps it
    __device__ __forceinline__ void store(const RunBlock &a, int64_t c)
    {
        const size_t n = (size_t)a.n_chains;
        double *s = a.stats + c;
        const Rows r = rows(a);
        s[0] = (double)cnt;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            s[(size_t)(r.r_prev + i) * n] = prev[i];
            s[(size_t)(r.r_sum + i) * n] = sum[i];
        }
        if (kXXRegs && r.second) {
#pragma unroll
            for (int t = 0; t < kT; ++t) s[(size_t)(r.r_xx + t) * n] = xx[t];
        }
        s[(size_t)r.r_dist * n] = dist;
        s[(size_t)r.r_hop * n] = hop;
        s[(size_t)r.r_p * n] = psum;
        s[(size_t)(r.r_p + 1) * n] = psq;
#pragma unroll
        for (int l = 0; l < kStatsMaxLags; ++l) {
            if (l < r.L) {
                s[(size_t)(r.r_lag + l) * n] = lag[l];
                s[(size_t)(r.r_ring + l) * n] = ring[(size_t)l * kBlock];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// RNG stream (DESIGN.md §3 "Random streams"): Philox4x32-10, counter = (block, step_lo, chain_lo,
// chain_hi16 | step_hi16 << 16), key = seed.  Integer part is bit-identical to the oracle.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4])
{
    // The key is wave-uniform.  Left alone, the compiler hoists the ten round keys k + r W out of the sampler's
    // loops as 20 long-lived scalar registers and then spills them into vector lanes (v_readlane on the vector
    // pipe per use).  The empty asm makes the key opaque here, so the round keys are re-derived by ten
    // scalar adds per call -- on the scalar unit, beside the vector work of the other wavefronts.
    k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);  // (a no-op when the key already sits in a scalar register)
    k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k1);
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96; the compiler emits two v_xor_b32 for this)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// The uniform of a try from ONE word of the stream (philox-v3): theta = lo + (hi - lo) w / 2^32 has 32 bits of resolution relative
// to the bracket, however far it has shrunk.  Exact: a 32-bit integer and a power of two.
__device__ __forceinline__ double try_uniform(uint32_t w) { return (double)w * 0x1.0p-32; }

__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    // ((a >> 5) 2^26 + (b >> 6)) / 2^53 as numpy forms it, without the 64-bit integer detour: both halves convert exactly,
    // the fma is exact (a 53-bit integer), the scaling a power of two -- the same double, six instructions instead of ten
    return fma((double)(a >> 5), 67108864.0, (double)(b >> 6)) * 0x1.0p-53;
}

// Box-Muller pair from two 32-bit words of a stream block: radius word -> (0,1], angle word -> [0,1)
// (single precision inside, fm::box_muller_f32: the normals only give the tangent its direction; the same bits as the oracle's)
__device__ __forceinline__ void box_muller32(uint32_t wr, uint32_t wa, double &z0, double &z1)
{
    float f0, f1;
    fm::box_muller_f32(wr, wa, f0, f1);
    z0 = (double)f0;
    z1 = (double)f1;
}

// the same pair with the table-driven log and sincos of gsss_math.h (throughput kernels; ~1 ulp from the above)
__device__ __forceinline__ void box_muller32(uint32_t wr, uint32_t wa, const fm::Tables &, double &z0, double &z1)
{
    box_muller32(wr, wa, z0, z1);
}

// Philox stream on S^2 (d = 3): the unit tangent u at x is drawn directly -- an angle phi in the tangent plane, ONE 32-bit
// word of block 1 -- instead of three normals projected and normalised (sphere.py:29-33, mcmc.py:387: the direction of the
// projected normal vector is uniform on the tangent circle, so the chain's law is the same; the replayed and the numpy
// streams keep the reference's normals).  n = x / |x|; (b1, b2) the branch-free orthonormal basis of the tangent plane of
// Duff et al., "Building an Orthonormal Basis, Revisited" (JCGT 2017); u = cos(phi) b1 + sin(phi) b2, unit and orthogonal
// to n to rounding.  The oracle forms the same expressions (gor_tangent3).
__device__ __forceinline__ void tangent3(double n0, double n1, double n2, double sn, double cs, double &u0, double &u1, double &u2)
{
    const double s = copysign(1.0, n2);
    const double a = -1.0 / (s + n2);
    const double b = n0 * n1 * a;
    const double b10 = 1.0 + s * n0 * n0 * a, b11 = s * b, b12 = -s * n0;
    const double b20 = b, b21 = s + n1 * n1 * a, b22 = -n1;
    u0 = fma(sn, b20, cs * b10);
    u1 = fma(sn, b21, cs * b11);
    u2 = fma(sn, b22, cs * b12);
}

// LDS doubles a kernel sets aside for fm::Tables (16-byte aligned inside the dynamic LDS block)
// (only the 64 points of the circle: the logarithm table of gsss_math.h served the double-precision Box-Muller pairs of round 1
// and has no user in the kernels since the pairs are formed in single precision -- 1.5 KB per workgroup that the compact
// Bingham kernel needs for its third workgroup per CU)
constexpr int kTabLds = 2 * 64 + 2;
__device__ __forceinline__ fm::Tables stage_tables(double *lds_after_params)
{
    // one double further if that is what 16-byte alignment takes.  (Pointer + integer: rounding the pointer's integer value
    // up and casting back makes it a generic pointer, and every table read of the kernel a flat_load instead of a ds_read.)
    double *buf = lds_after_params + ((reinterpret_cast<uintptr_t>(lds_after_params) >> 3) & 1u);
    for (int i = threadIdx.x; i < 64; i += kBlock) fm::table_entry(buf, i);
    return fm::Tables{buf, nullptr};
}

constexpr uint64_t kInitStep = 0xFFFFFFFFFFFFull;  // reserved step id: initial states (gsss_sample_sphere)

// ------------------------------------------------------------------------------------------
// Vector policies: how the d components of one chain map onto lanes.
//   LaneVec<D>   : one lane per chain, all D components in that lane's registers.
//   CoopVec<L,S> : L lanes per chain (L | 64), S register slots per lane, components dealt in
//                  quads (4q .. 4q+3), quad q = g + L*iq, so each lane owns whole RNG blocks.
// Slots beyond d hold zeros (and zero-padded parameters), so dots need no guards.
// ------------------------------------------------------------------------------------------
template <int D_>
struct LaneVec {
    static constexpr int L = 1;
    static constexpr int N = D_;
    static constexpr int DPAD = D_;
    static constexpr bool kExactDim = true;  // requires d == D_
    __device__ static __forceinline__ int comp(int /*g*/, int i) { return i; }
    __device__ static __forceinline__ double reduce(double v) { return v; }
};

// ------------------------------------------------------------------------------------------
// Cross-lane moves without LDS: DPP within a row of 16 lanes, scalar broadcasts across rows.
// (`__shfl_xor` lowers to ds_bpermute, two LDS round trips per double; the reductions of the
// cooperative kernels sit on the critical path of every step.)
// ------------------------------------------------------------------------------------------
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7-i within 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15-i within 16

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    // (bound_ctrl on: every lane of these permutations has a source, and the destination then needs no prior value --
    // with it off the compiler zeroes the destination pair before every move)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_move(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ double lane_broadcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// sum over the L_ lanes of a group, identical bits in every lane: after the two quad steps all four
// lanes of a quad agree, so the mirror steps add equal partners in either order (commutative)
template <int L_>
__device__ __forceinline__ double group_sum(double v)
{
    if (L_ >= 2) v += dpp_move<kDppXor1>(v);
    if (L_ >= 4) v += dpp_move<kDppXor2>(v);
    if (L_ >= 8) v += dpp_move<kDppHalfMirror>(v);
    if (L_ >= 16) v += dpp_move<kDppMirror>(v);
    if (L_ == 64) {  // the wave is one group: add the four row sums in a fixed order
        const double r0 = lane_broadcast(v, 0), r1 = lane_broadcast(v, 16);
        const double r2 = lane_broadcast(v, 32), r3 = lane_broadcast(v, 48);
        v = (r0 + r1) + (r2 + r3);
    }
    return v;
}

template <int L_, int S_>
struct CoopVec {
    static_assert(S_ % 4 == 0, "slots come in quads");
    static constexpr int L = L_;
    static constexpr int N = S_;
    static constexpr int DPAD = L_ * S_;
    static constexpr bool kExactDim = false;  // any d <= DPAD
    __device__ static __forceinline__ int comp(int g, int i) { return 4 * (g + L_ * (i >> 2)) + (i & 3); }
    // sum over the L lanes of a group: every lane ends with the same bits
    static_assert(L_ == 2 || L_ == 4 || L_ == 8 || L_ == 16 || L_ == 64, "group sizes with a DPP reduction");
    __device__ static __forceinline__ double reduce(double v) { return group_sum<L_>(v); }
};

template <class V>
__device__ __forceinline__ double vdot(const double (&a)[V::N], const double (&b)[V::N])
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < V::N; ++i) s = fma(a[i], b[i], s);
    return V::reduce(s);
}

// ------------------------------------------------------------------------------------------
// Running statistics for the lane-GROUP layouts (CoopVec: L lanes hold the components of one chain): the same rows and
// definitions as stats_update above.  Every lane of the group calls it with its own slots; the dots are group sums, lane 0
// of the group keeps the scalar rows, every lane the rows of its own components.  The second moments need every pair of
// components: the other lanes' slots are fetched lane by lane (ds_bpermute) -- a statistics build only, and only while
// GSSS_STATS_NO_SECOND_MOMENT is off (d (d + 1) / 2 rows: the host side leaves them out beyond d = 16).
// ------------------------------------------------------------------------------------------
template <class V>
__device__ __forceinline__ void stats_update_group(const RunBlock &a, int64_t c, int g, int d, const double (&x)[V::N])
{
    constexpr int N = V::N, LG = V::L;
    const size_t n = (size_t)a.n_chains;
    double *s = a.stats + c;
    const int K = a.stats_modes, L = a.stats_lags;
    const bool second = !(a.stats_flags & GSSS_STATS_NO_SECOND_MOMENT);
    const int T = second ? d * (d + 1) / 2 : 0;
    const int r_prev = 1, r_sum = 1 + d, r_xx = 1 + 2 * d, r_dist = r_xx + T, r_hop = r_dist + 1, r_mode = r_hop + 1;
    const int r_p = r_mode + K, r_lag = r_p + 2, r_ring = r_lag + L, r_head = r_ring + L;
    const double *w = a.stats_dirs, *h = a.stats_dirs + d, *modes = a.stats_dirs + 2 * d;
    // the draw count is read by the lane that writes it (lane 0 of the group, at the end of this function) and handed to the
    // others through the group sum: no lane reads a word another lane stores
    const int64_t cnt = (int64_t)V::reduce(g == 0 ? s[0] : 0.0);
    double p = 0.0, xh = 0.0, dot = 0.0, ph = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int cc = V::comp(g, i);
        if (cc < d) {
            p = fma(x[i], w[cc], p);
            xh = fma(x[i], h[cc], xh);
            if (cnt > 0) {
                const double pj = s[(size_t)(r_prev + cc) * n];
                dot = fma(pj, x[i], dot);
                ph = fma(pj, h[cc], ph);
            }
        }
    }
    p = V::reduce(p);
    xh = V::reduce(xh);
    dot = V::reduce(dot);
    ph = V::reduce(ph);
    if (cnt > 0 && g == 0) {
        s[(size_t)r_dist * n] += acos(fmin(fmax(dot, -1.0), 1.0));
        const int sa = (xh > 0.0) - (xh < 0.0), sb = (ph > 0.0) - (ph < 0.0);  // np.sign
        if (sa != sb) s[(size_t)r_hop * n] += 1.0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int cc = V::comp(g, i);
        if (cc < d) {
            s[(size_t)(r_prev + cc) * n] = x[i];
            s[(size_t)(r_sum + cc) * n] += x[i];
        }
    }
    if (second) {  // pair (ci <= cj) lives in row r_xx + ci d - ci (ci - 1) / 2 + (cj - ci)
        const int base = (int)(threadIdx.x % 64) - g;
        for (int jl = 0; jl < LG; ++jl) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double xj = __shfl(x[j], base + jl);
                const int cj = V::comp(jl, j);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int ci = V::comp(g, i);
                    if (cj < d && ci <= cj) s[(size_t)(r_xx + ci * d - ci * (ci - 1) / 2 + (cj - ci)) * n] += x[i] * xj;
                }
            }
        }
    }
    if (K > 0) {
        int best = 0;
        double bv = -INFINITY;
        for (int k = 0; k < K; ++k) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int cc = V::comp(g, i);
                if (cc < d) v = fma(x[i], modes[(size_t)k * d + cc], v);
            }
            v = V::reduce(v);
            if (v > bv) {  // first maximum, like np.argmax
                bv = v;
                best = k;
            }
        }
        if (g == 0) s[(size_t)(r_mode + best) * n] += 1.0;
    }
    if (g == 0) {
        s[(size_t)r_p * n] += p;
        s[(size_t)(r_p + 1) * n] += p * p;
        if (L > 0) {
            const int64_t lmax = cnt < L ? cnt : L;
            for (int64_t l = 1; l <= lmax; ++l) s[(size_t)(r_lag + l - 1) * n] += p * s[(size_t)(r_ring + (cnt - l) % L) * n];
            s[(size_t)(r_ring + cnt % L) * n] = p;
            if (cnt < L) s[(size_t)(r_head + cnt) * n] = p;
        }
        s[0] = (double)(cnt + 1);
    }
}

// ------------------------------------------------------------------------------------------
// Draw sources
// ------------------------------------------------------------------------------------------
template <class V, bool TAB = false>
struct PhiloxDraws {
    fm::Tables tab;  // TAB: Box-Muller on the table-driven log / sincos (set by the kernel)
    static constexpr bool kReplay = false;
    static constexpr int kLdsDoubles = 0;
    __device__ __forceinline__ void stage(double *) {}
    __device__ __forceinline__ void finish(const RunBlock &, int64_t, bool) const {}
    uint32_t k0, k1, c1, c2, c3, chain_hi;
    int d, t;
    uint32_t cw0, cw1, cw2, cw3;  // the words of the try block in use (next_try)
    bool exhausted;

    __device__ __forceinline__ void init(const RunBlock &a, int64_t chain_local, int d_)
    {
        const uint64_t chain = a.chain_offset + (uint64_t)chain_local;
        k0 = (uint32_t)a.seed;
        k1 = (uint32_t)(a.seed >> 32);
        c2 = (uint32_t)chain;
        chain_hi = (uint32_t)((chain >> 32) & 0xFFFFu);
        d = d_;
        exhausted = false;
    }
    __device__ __forceinline__ void begin_step(uint64_t step)
    {
        c1 = (uint32_t)step;
        c3 = chain_hi | ((uint32_t)((step >> 32) & 0xFFFFu) << 16);
        t = 0;
    }
    __device__ __forceinline__ void words(uint32_t blk, uint32_t (&w)[4]) const
    {
        philox4x32_10(blk, c1, c2, c3, k0, k1, w);
    }
    __device__ __forceinline__ void block(uint32_t blk, double &u0, double &u1) const
    {
        uint32_t w[4];
        words(blk, w);
        u0 = u53(w[0], w[1]);
        u1 = u53(w[2], w[3]);
    }
    // block 1+q carries the normals 4q .. 4q+3 (blk_off: a second set of d normals further down the stream)
    __device__ __forceinline__ void normals(double (&z)[V::N], int g, uint32_t blk_off = 0u) const
    {
#pragma unroll
        for (int iq = 0; iq < (V::N + 3) / 4; ++iq) {
            const int c0 = V::comp(g, 4 * iq);
            double zz[4] = {0.0, 0.0, 0.0, 0.0};
            if (c0 < d) {
                uint32_t w[4];
                words(1u + blk_off + (uint32_t)(c0 >> 2), w);
                if constexpr (TAB) {
                    box_muller32(w[0], w[1], tab, zz[0], zz[1]);
                    if (c0 + 2 < d) box_muller32(w[2], w[3], tab, zz[2], zz[3]);
                } else {
                    box_muller32(w[0], w[1], zz[0], zz[1]);
                    if (c0 + 2 < d) box_muller32(w[2], w[3], zz[2], zz[3]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * iq + i < V::N) z[4 * iq + i] = (c0 + i < d) ? zz[i] : 0.0;
        }
    }
    // S^2 (d = 3): block 0 carries the whole set-up of a step -- U_threshold (53 bits: words 0, 1), U_theta0 (32 bits: word 2;
    // the offset of the bracket, mcmc.py:391) and the angle of the tangent direction (word 3) -- and block 1 is not drawn
    static constexpr bool kTangent3 = true;
    __device__ __forceinline__ void step_s2(double &u_thr, double &u_theta0, uint32_t &w_phi) const
    {
        uint32_t w[4];
        words(0u, w);
        u_thr = u53(w[0], w[1]);
        u_theta0 = (double)w[2] * 0x1.0p-32;
        w_phi = w[3];
    }
    // the unit tangent at n = x / |x| for the angle word w_phi (tangent3); lanes that hold no component keep zeros
    __device__ __forceinline__ void tangent(const double (&nrm)[V::N], double (&u)[V::N], int g, uint32_t w_phi) const
    {
        if constexpr (V::N >= 3) {
            double sn, cs;
            if constexpr (TAB)
                fm::sincos_word_tab(w_phi, tab, sn, cs);
            else
                fm::sincos_2pi((double)w_phi * 0x1.0p-32, sn, cs);
            double t0, t1, t2;
            tangent3(nrm[0], nrm[1], nrm[2], sn, cs, t0, t1, t2);
            const bool mine = V::comp(g, 0) == 0;
#pragma unroll
            for (int i = 0; i < V::N; ++i) u[i] = 0.0;
            u[0] = mine ? t0 : 0.0;
            u[1] = mine ? t1 : 0.0;
            u[2] = mine ? t2 : 0.0;
        }
    }
    __device__ __forceinline__ void step_uniforms(double &u_thr, double &u_theta0, bool /*need_theta0*/) const
    {
        block(0u, u_thr, u_theta0);
    }
    // the uniform of try t (mcmc.py:395): ONE 32-bit word -- word t % 4 of block 1 + nb + t / 4 (philox-v3, round 5; try_uniform)
    __device__ __forceinline__ double next_try()
    {
        const int tt = t++;
        if ((tt & 3) == 0) {
            uint32_t w[4];
            words(1u + (uint32_t)((d + 3) >> 2) + (uint32_t)(tt >> 2), w);
            cw0 = w[0];
            cw1 = w[1];
            cw2 = w[2];
            cw3 = w[3];
        }
        const int h = tt & 3;
        return try_uniform(h == 0 ? cw0 : (h == 1 ? cw1 : (h == 2 ? cw2 : cw3)));
    }
    // RWMH (mcmc.py:143): r = sqrt(2 gamma(d/2)) is a chi_d variate -- here the norm of d further normals
    __device__ __forceinline__ double chi(int g) const
    {
        double z2[V::N];
        normals(z2, g, (uint32_t)((d + 3) >> 2));
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) s = fma(z2[i], z2[i], s);
        return sqrt(V::reduce(s));
    }
    __device__ __forceinline__ double accept_uniform() const
    {
        double u0, u1;
        block(0u, u0, u1);
        return u0;
    }
    // MixtureRWMHIndependenceSampler's choice of kernel (mcmc.py:213): the other half of block 0
    __device__ __forceinline__ double mix_uniform() const
    {
        double u0, u1;
        block(0u, u0, u1);
        return u1;
    }
};

template <class V>
struct ReplayDraws {
    static constexpr bool kReplay = true;
    static constexpr int kLdsDoubles = 0;
    __device__ __forceinline__ void stage(double *) {}
    __device__ __forceinline__ void finish(const RunBlock &, int64_t, bool) const {}
    const double *p;
    int64_t len, cur;
    int d;
    bool exhausted;

    __device__ __forceinline__ void init(const RunBlock &a, int64_t chain_local, int d_)
    {
        p = a.replay + chain_local * a.replay_stride;
        len = a.replay_stride;
        cur = 0;
        d = d_;
        exhausted = false;
    }
    __device__ __forceinline__ void begin_step(uint64_t) {}
    __device__ __forceinline__ double take()
    {
        if (cur >= len) {
            exhausted = true;
            return 0.5;
        }
        return p[cur++];
    }
    __device__ __forceinline__ void normals(double (&z)[V::N], int g)
    {
        const bool ok = cur + d <= len;
#pragma unroll
        for (int i = 0; i < V::N; ++i) {
            const int c = V::comp(g, i);
            z[i] = (c < d) ? (ok ? p[cur + c] : 0.5) : 0.0;
        }
        if (ok)
            cur += d;
        else {
            cur = len;
            exhausted = true;
        }
    }
    __device__ __forceinline__ void step_uniforms(double &u_thr, double &u_theta0, bool need_theta0)
    {
        u_thr = take();
        u_theta0 = need_theta0 ? take() : 0.0;
    }
    static constexpr bool kTangent3 = false;  // the reference's own draws: d normals
    __device__ __forceinline__ void step_s2(double &, double &, uint32_t &) const {}
    __device__ __forceinline__ void tangent(const double (&)[V::N], double (&)[V::N], int, uint32_t) const {}
    __device__ __forceinline__ double next_try() { return take(); }
    __device__ __forceinline__ double chi(int) { return sqrt(2.0 * take()); }  // the recorded gamma(d/2) variate
    __device__ __forceinline__ double accept_uniform() { return take(); }
    __device__ __forceinline__ double mix_uniform() { return take(); }
};

// numpy's own stream: PCG64 (XSL-RR 128/64) + Generator.random / uniform / standard_normal, so that
// a chain seeded like the reference (np.random.default_rng(seed), geosss/mcmc.py:45) consumes the very
// numbers the reference consumes (mcmc.py:387, 389, 391, 395).  Sequential per chain by nature: every
// lane of a cooperative group runs the identical generator and keeps the components it owns.
#define NPY_ZIG_NOR_R 3.6541528853610087963519472518
#define NPY_ZIG_NOR_INV_R 0.27366123732975827203338247596
__device__ const uint64_t kNpyKi[256] = {
#define NPY_ZIG_ONLY_KI
#include "numpy_ziggurat_tables.inc"
#undef NPY_ZIG_ONLY_KI
};
__device__ const double kNpyWi[256] = {
#define NPY_ZIG_ONLY_WI
#include "numpy_ziggurat_tables.inc"
#undef NPY_ZIG_ONLY_WI
};
__device__ const double kNpyFi[256] = {
#define NPY_ZIG_ONLY_FI
#include "numpy_ziggurat_tables.inc"
#undef NPY_ZIG_ONLY_FI
};

template <class V>
struct NumpyDraws {
    static constexpr bool kReplay = false;
    static constexpr int kLdsDoubles = 768;
    uint64_t sh, sl, ih, il;
    const double *wi, *fi;
    const uint64_t *ki;
    int d;
    bool exhausted;

    __device__ __forceinline__ void stage(double *lds)
    {
        uint64_t *k = reinterpret_cast<uint64_t *>(lds);
        for (int i = threadIdx.x; i < 256; i += kBlock) {
            k[i] = kNpyKi[i];
            lds[256 + i] = kNpyWi[i];
            lds[512 + i] = kNpyFi[i];
        }
        ki = k;
        wi = lds + 256;
        fi = lds + 512;
    }
    __device__ __forceinline__ void init(const RunBlock &a, int64_t chain_local, int d_)
    {
        const uint64_t *w = a.rng_state + 4 * chain_local;
        sh = w[0];
        sl = w[1];
        ih = w[2];
        il = w[3];
        d = d_;
        exhausted = false;
    }
    __device__ __forceinline__ void finish(const RunBlock &a, int64_t chain_local, bool store) const
    {
        if (!store) return;
        uint64_t *w = a.rng_state + 4 * chain_local;
        w[0] = sh;
        w[1] = sl;
    }
    __device__ __forceinline__ void begin_step(uint64_t) {}
    __device__ __forceinline__ uint64_t next64()
    {
        constexpr uint64_t MH = 2549297995355413924ull, ML = 4865540595714422341ull;
        // (sh:sl) = (sh:sl) * (MH:ML) + (ih:il)  mod 2^128
        const uint64_t lo = sl * ML;
        uint64_t hi = __umul64hi(sl, ML) + sl * MH + sh * ML;
        const uint64_t nlo = lo + il;
        hi += ih + (nlo < lo ? 1ull : 0ull);
        sl = nlo;
        sh = hi;
        const uint64_t x = sh ^ sl;
        const unsigned rot = (unsigned)(sh >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    __device__ __forceinline__ double next_double() { return (double)(next64() >> 11) * 0x1.0p-53; }
    __device__ double standard_normal()
    {
        for (;;) {
            uint64_t r = next64();
            const int idx = (int)(r & 0xff);
            r >>= 8;
            const bool neg = r & 1;
            const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
            double x = (double)rabs * wi[idx];
            if (neg) x = -x;
            if (rabs < ki[idx]) return x;
            if (idx == 0) {
                for (;;) {
                    const double xx = -NPY_ZIG_NOR_INV_R * log1p(-next_double());
                    const double yy = -log1p(-next_double());
                    if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(NPY_ZIG_NOR_R + xx) : NPY_ZIG_NOR_R + xx;
                }
            } else if (((fi[idx - 1] - fi[idx]) * next_double() + fi[idx]) < exp(-0.5 * x * x)) {
                return x;
            }
        }
    }
    __device__ __forceinline__ void normals(double (&z)[V::N], int g)
    {
#pragma unroll
        for (int i = 0; i < V::N; ++i) z[i] = 0.0;
        for (int c = 0; c < d; ++c) {
            const double v = standard_normal();
#pragma unroll
            for (int i = 0; i < V::N; ++i)
                if (V::comp(g, i) == c) z[i] = v;
        }
    }
    __device__ __forceinline__ void step_uniforms(double &u_thr, double &u_theta0, bool need_theta0)
    {
        u_thr = next_double();
        u_theta0 = need_theta0 ? next_double() : 0.0;
    }
    static constexpr bool kTangent3 = false;  // the reference's own draws: d normals
    __device__ __forceinline__ void step_s2(double &, double &, uint32_t &) const {}
    __device__ __forceinline__ void tangent(const double (&)[V::N], double (&)[V::N], int, uint32_t) const {}
    __device__ __forceinline__ double next_try() { return next_double(); }
    // Generator.gamma(shape), shape > 1: numpy's random_standard_gamma (Marsaglia-Tsang on the ziggurat normals)
    __device__ double standard_gamma(double shape)
    {
        const double b = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * b);
        for (;;) {
            double X, Vv;
            do {
                X = standard_normal();
                Vv = 1.0 + c * X;
            } while (Vv <= 0.0);
            Vv = Vv * Vv * Vv;
            const double U = next_double();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * Vv;
            if (log(U) < 0.5 * X * X + b * (1.0 - Vv + log(Vv))) return b * Vv;
        }
    }
    __device__ __forceinline__ double chi(int) { return sqrt(2.0 * standard_gamma(0.5 * (double)d)); }  // mcmc.py:143
    __device__ __forceinline__ double accept_uniform() { return next_double(); }
    __device__ __forceinline__ double mix_uniform() { return next_double(); }
};

// ------------------------------------------------------------------------------------------
// Targets.  Each stages its parameters in LDS (zero-padded to V::DPAD columns) and evaluates
// log_prob(y) from the components of y held by the calling lane group.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fill(double *dst, int n_rows, int dpad, const double *src, int d)
{
    // rows of length d in global memory -> rows of length dpad in LDS, zero padded
    for (int i = threadIdx.x; i < n_rows * dpad; i += kBlock) {
        const int r = i / dpad, c = i - r * dpad;
        dst[i] = (c < d) ? src[(size_t)r * d + c] : 0.0;
    }
}

constexpr size_t kRowsLdsBytes = (size_t)136 * 1024;
// Target rows (component means, knots) live in LDS, zero padded to the layout's DPAD -- or, where they do not fit a workgroup's
// LDS (many components or knots at large d), are read from the parameter blob in global memory as they are (stride d): the same
// products in the same order.  Host and device take the decision from (rows, DPAD) alone; 136 KB leave room for scratch and tables.
// Only the sixty-four-lane layouts (DPAD >= 256) carry the global-memory path: the host runs a target whose rows do not fit the
// LDS of its natural layout on the smallest of them (select_vec_for, gsss_capi.hip), and the kernels of the smaller layouts --
// the ones whose speed matters in exact mode -- keep their code as it was (measured with a run-time choice in every layout:
// 6 .. 20 % slower exact kernels at d <= 10).
template <class V>
__host__ __device__ constexpr bool rows_fit_lds(size_t doubles)
{
    return V::DPAD < 256 || doubles * sizeof(double) <= kRowsLdsBytes;
}
// component c of a row: LDS rows are padded with zeros, global rows end at d
template <bool GLOBAL, class V>
__device__ __forceinline__ double row_at(const double *row, int c, int d)
{
    if constexpr (GLOBAL)
        return c < d ? row[c] : 0.0;
    else
        return row[c];
}

template <class V>
struct VmfMixture {
    const double *mu;    // LDS [K][DPAD]; nullptr: the rows are read from `mug`
    const double *mug;   // global [K][d]
    const double *logc;  // [K], LDS or global like the rows
    int K, d;
    static constexpr bool kMayGlobal = V::DPAD >= 256;  // (rows_fit_lds)

    __host__ __device__ static bool in_lds(int k) { return rows_fit_lds<V>((size_t)k * V::DPAD + k); }
    __host__ __device__ static size_t lds_doubles(int k, int /*d*/) { return in_lds(k) ? (size_t)k * V::DPAD + k : 0; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        K = tb.k;
        d = tb.d;
        mug = tb.blob;
        if (in_lds(K)) {
            lds_fill(lds, K, V::DPAD, tb.blob, tb.d);
            double *lc = lds + (size_t)K * V::DPAD;
            for (int i = threadIdx.x; i < K; i += kBlock) lc[i] = tb.blob[(size_t)K * tb.d + i];
            mu = lds;
            logc = lc;
        } else {
            mu = nullptr;
            logc = tb.blob + (size_t)K * tb.d;
        }
    }
    template <bool G>
    __device__ __forceinline__ const double *row(int k) const
    {
        return G ? mug + (size_t)k * d : mu + (size_t)k * V::DPAD;
    }
    template <bool G>
    __device__ __forceinline__ double comp_logp_t(const double (&y)[V::N], int g, int k) const
    {
        const double *m = row<G>(k);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) s = fma(y[i], row_at<G, V>(m, V::comp(g, i), d), s);
        return V::reduce(s) + logc[k];
    }
    __device__ __forceinline__ double comp_logp(const double (&y)[V::N], int g, int k) const
    {
        return (!kMayGlobal || mu != nullptr) ? comp_logp_t<false>(y, g, k) : comp_logp_t<true>(y, g, k);
    }
    // logsumexp_k( y.mu_k + logc_k )
    __device__ __forceinline__ double logp(const double (&y)[V::N], int g, double * /*scratch*/) const
    {
        double amax = -INFINITY;
        for (int k = 0; k < K; ++k) amax = fmax(amax, comp_logp(y, g, k));
        if (!(amax > -INFINITY) || amax == INFINITY) return amax;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += fm::exp_fast(comp_logp(y, g, k) - amax);
        return amax + fm::log_fast(s);
    }
    // distributions.py:223-227 : sum_k exp(p_k) mu_k / exp(logsumexp(p))  (the softmax-weighted mean of the mu_k)
    template <bool G>
    __device__ __forceinline__ void grad_t(const double (&y)[V::N], int g, double (&out)[V::N]) const
    {
        double amax = -INFINITY;
        for (int k = 0; k < K; ++k) amax = fmax(amax, comp_logp_t<G>(y, g, k));
        double den = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) out[i] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double w = fm::exp_fast(comp_logp_t<G>(y, g, k) - amax);
            den += w;
            const double *m = row<G>(k);
#pragma unroll
            for (int i = 0; i < V::N; ++i) out[i] = fma(w, row_at<G, V>(m, V::comp(g, i), d), out[i]);
        }
#pragma unroll
        for (int i = 0; i < V::N; ++i) out[i] /= den;
    }
    __device__ __forceinline__ void grad(const double (&y)[V::N], int g, double * /*scratch*/, double (&out)[V::N]) const
    {
        if (!kMayGlobal || mu != nullptr)
            grad_t<false>(y, g, out);
        else
            grad_t<true>(y, g, out);
    }
    static constexpr int kScratchPerChain = 0;
};

template <class V>
struct Bingham {
    const double *A;   // LDS [d][DPAD]: rows of A, columns zero padded
    const double *Ag;  // ... or, where d^2 doubles do not fit in LDS (d > 128), the blob's own rows [d][d] in global memory (L2)
    const double *b;   // LDS [DPAD]: BinghamFisher linear term (zeros for a plain Bingham)
    int d;
    static constexpr bool kMayGlobal = V::DPAD >= 256;  // (rows_fit_lds)
    // (host and device take the same decision from (d, DPAD): 136 KB leave room for the groups' scratch rows and the tables)
    __host__ __device__ static bool in_lds(int d) { return rows_fit_lds<V>((size_t)(d + 1) * V::DPAD); }
    __host__ __device__ static size_t lds_doubles(int /*k*/, int d) { return in_lds(d) ? (size_t)(d + 1) * V::DPAD : (size_t)V::DPAD; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        d = tb.d;
        if (in_lds(tb.d)) {
            lds_fill(lds, tb.d + 1, V::DPAD, tb.blob, tb.d);  // blob = A rows followed by b
            A = lds;
            Ag = nullptr;
            b = lds + (size_t)tb.d * V::DPAD;
        } else {
            lds_fill(lds, 1, V::DPAD, tb.blob + (size_t)tb.d * tb.d, tb.d);
            A = nullptr;
            Ag = tb.blob;
            b = lds;
        }
    }
    // row i of A at this lane's slots, from global memory
    __device__ __forceinline__ void row_global(int i, int g, double (&r)[V::N]) const
    {
        const double *row = Ag + (size_t)i * d;
#pragma unroll
        for (int j = 0; j < V::N; ++j) {
            const int c = V::comp(g, j);
            r[j] = c < d ? row[c] : 0.0;
        }
    }
    __device__ __forceinline__ double linear(const double (&y)[V::N], int g) const
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) s = fma(y[i], b[V::comp(g, i)], s);
        return s;  // partial over this lane's slots
    }
    // sum_j (sum_i y_i A_ij) y_j
    __device__ __forceinline__ double logp(const double (&y)[V::N], int g, double *scratch) const
    {
        double s = 0.0;
        if constexpr (V::L == 1) {
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                double xa = 0.0;
#pragma unroll
                for (int i = 0; i < V::N; ++i) xa = fma(y[i], A[i * V::DPAD + j], xa);
                s = fma(xa, y[j], s);
            }
            return s + linear(y, g);
        } else {
            // publish y to the group's LDS row, then every lane forms (yA)_j for its own slots j
#pragma unroll
            for (int i = 0; i < V::N; ++i) scratch[V::comp(g, i)] = y[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double xa[V::N];
#pragma unroll
            for (int j = 0; j < V::N; ++j) xa[j] = 0.0;
            if (!kMayGlobal || Ag == nullptr) {
                for (int i = 0; i < d; ++i) {
                    const double yi = scratch[i];
#pragma unroll
                    for (int j = 0; j < V::N; ++j) xa[j] = fma(yi, A[i * V::DPAD + V::comp(g, j)], xa[j]);
                }
            } else {  // (the same products in the same order: the same bits as the LDS copy would give)
                for (int i = 0; i < d; ++i) {
                    const double yi = scratch[i];
                    double r[V::N];
                    row_global(i, g, r);
#pragma unroll
                    for (int j = 0; j < V::N; ++j) xa[j] = fma(yi, r[j], xa[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < V::N; ++j) s = fma(xa[j], y[j], s);
            s += linear(y, g);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return V::reduce(s);
        }
    }
    // distributions.py:88-89 : 2 A y (also what BinghamFisher inherits in the reference: its b does not enter)
    __device__ __forceinline__ void grad(const double (&y)[V::N], int g, double *scratch, double (&out)[V::N]) const
    {
        if constexpr (V::L == 1) {
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                double xa = 0.0;
#pragma unroll
                for (int i = 0; i < V::N; ++i) xa = fma(A[j * V::DPAD + i], y[i], xa);
                out[j] = 2.0 * xa;
            }
        } else {
#pragma unroll
            for (int i = 0; i < V::N; ++i) scratch[V::comp(g, i)] = y[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < V::N; ++j) out[j] = 0.0;
            if (!kMayGlobal || Ag == nullptr) {
                for (int i = 0; i < d; ++i) {  // (A y)_j = sum_i A_ij y_i for the lane's own components j (A symmetric)
                    const double yi = scratch[i];
#pragma unroll
                    for (int j = 0; j < V::N; ++j) out[j] = fma(yi, A[i * V::DPAD + V::comp(g, j)], out[j]);
                }
            } else {
                for (int i = 0; i < d; ++i) {
                    const double yi = scratch[i];
                    double r[V::N];
                    row_global(i, g, r);
#pragma unroll
                    for (int j = 0; j < V::N; ++j) out[j] = fma(yi, r[j], out[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < V::N; ++j) out[j] *= 2.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    static constexpr int kScratchPerChain = (V::L == 1) ? 0 : V::DPAD + 1;
};

template <class V>
struct CurveVmf {
    const double *knots;   // LDS [K][DPAD]; nullptr: the rows are read from `knotsg` (rows_fit_lds)
    const double *knotsg;  // global [K][d]
    const double *seg;     // [K-1][4] : theta, cos(theta), sin(theta), sin(theta)+1e-10  (spherical_curve.py:28-31); LDS or global like the rows
    int K, d;
    double kappa;
    static constexpr bool kMayGlobal = V::DPAD >= 256;  // (rows_fit_lds)
    __host__ __device__ static bool in_lds(int k) { return rows_fit_lds<V>((size_t)k * V::DPAD + 4 * (size_t)(k - 1)); }
    __host__ __device__ static size_t lds_doubles(int k, int /*d*/) { return in_lds(k) ? (size_t)k * V::DPAD + 4 * (size_t)(k - 1) : 0; }
    __device__ void stage(double *lds, const TargetBlock &tb)
    {
        K = tb.k;
        d = tb.d;
        kappa = tb.kappa;
        knotsg = tb.blob;
        if (in_lds(K)) {
            lds_fill(lds, K, V::DPAD, tb.blob, tb.d);
            double *sg = lds + (size_t)K * V::DPAD;
            for (int i = threadIdx.x; i < 4 * (K - 1); i += kBlock) sg[i] = tb.blob[(size_t)K * tb.d + i];
            knots = lds;
            seg = sg;
        } else {
            knots = nullptr;
            seg = tb.blob + (size_t)K * tb.d;
        }
    }
    template <bool G>
    __device__ __forceinline__ const double *row(int k) const
    {
        return G ? knotsg + (size_t)k * d : knots + (size_t)k * V::DPAD;
    }
    template <bool G>
    __device__ __forceinline__ double kdot(const double (&y)[V::N], int g, int k) const
    {
        const double *a = row<G>(k);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < V::N; ++i) s = fma(row_at<G, V>(a, V::comp(g, i), d), y[i], s);
        return V::reduce(s);
    }
    // the segments in order: y . nearest of each, the first of minimal distance wins; `keep(near)` is called for every new best
    template <bool G, class Keep>
    __device__ __forceinline__ double scan(const double (&y)[V::N], int g, Keep keep) const
    {
        double best = INFINITY, best_dot = 0.0;
        double ay = kdot<G>(y, g, 0);
        for (int s = 0; s + 1 < K; ++s) {
            const double by = kdot<G>(y, g, s + 1);
            const double theta = seg[4 * s], ct = seg[4 * s + 1], st = seg[4 * s + 2], den = seg[4 * s + 3];
            double t = atan2(by - ay * ct, ay * st);
            t = fmin(fmax(t, 0.0), theta);
            const double sa = sin(theta - t), sb = sin(t);
            const double *a = row<G>(s), *b = row<G>(s + 1);
            double near[V::N], xy = 0.0;
#pragma unroll
            for (int i = 0; i < V::N; ++i) {
                const int c = V::comp(g, i);
                near[i] = (sa * row_at<G, V>(a, c, d) + sb * row_at<G, V>(b, c, d)) / den;
                xy = fma(y[i], near[i], xy);
            }
            xy = V::reduce(xy);
            const double dist = acos(fmin(fmax(xy, -1.0), 1.0));
            if (dist < best) {
                best = dist;
                best_dot = xy;
                keep(near);
            }
            ay = by;
        }
        return best_dot;
    }
    // kappa * (y . nearest(y)); nearest = closest point of the first segment of minimal distance
    __device__ __forceinline__ double logp(const double (&y)[V::N], int g, double * /*scratch*/) const
    {
        auto nothing = [](const double (&)[V::N]) {};
        return kappa * ((!kMayGlobal || knots != nullptr) ? scan<false>(y, g, nothing) : scan<true>(y, g, nothing));
    }
    // distributions.py:277-278 : kappa * find_nearest(y)
    __device__ __forceinline__ void grad(const double (&y)[V::N], int g, double * /*scratch*/, double (&out)[V::N]) const
    {
#pragma unroll
        for (int i = 0; i < V::N; ++i) out[i] = 0.0;
        auto keep = [&](const double (&near)[V::N]) {
#pragma unroll
            for (int i = 0; i < V::N; ++i) out[i] = kappa * near[i];
        };
        if (!kMayGlobal || knots != nullptr)
            (void)scan<false>(y, g, keep);
        else
            (void)scan<true>(y, g, keep);
    }
    static constexpr int kScratchPerChain = 0;
};

// ------------------------------------------------------------------------------------------
// The sampler: one lane group per chain, step-synchronous.
// ------------------------------------------------------------------------------------------
template <class V, class T>
__host__ __device__ constexpr size_t scratch_doubles()
{
    return (size_t)T::kScratchPerChain * (kBlock / V::L);
}

template <class V, template <class> class TT, template <class> class DR, bool STATS = false>
__global__ void __launch_bounds__(kBlock) run_kernel(TargetBlock tb, RunBlock a)
{
    using T = TT<V>;
    using Draws = DR<V>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    T tgt;
    tgt.stage(lds, tb);
    double *scratch = lds + T::lds_doubles(tb.k, tb.d) + (size_t)T::kScratchPerChain * (threadIdx.x / V::L);
    Draws dr;
    dr.stage(lds + T::lds_doubles(tb.k, tb.d) + scratch_doubles<V, T>());
    __syncthreads();

    const int d = tb.d;
    const int g = threadIdx.x % V::L;
    const int64_t n = a.n_chains;
    // packed: consecutive lane groups take consecutive chains.  spread (lane layouts, small ensembles):
    // one chain per wavefront -- lane 0 owns it, the other lanes shadow it, so the wave never diverges
    const bool spread = V::L == 1 && a.spread;
    const int64_t c_raw = spread ? (int64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64
                                 : (int64_t)blockIdx.x * (kBlock / V::L) + threadIdx.x / V::L;
    const bool active = c_raw < n && (!spread || threadIdx.x % 64 == 0);
    const int64_t c = c_raw < n ? c_raw : n - 1;  // surplus lanes shadow a chain and store nothing

    double x[V::N];
#pragma unroll
    for (int i = 0; i < V::N; ++i) {
        const int cc = V::comp(g, i);
        x[i] = (cc < d) ? a.state[(size_t)cc * n + c] : 0.0;
    }
    dr.init(a, c, d);

    double px = tgt.logp(x, g, scratch);
    int err = 0;
    int64_t n_rej = 0, n_try = 0;
    const bool shrink = a.sampler == GSSS_SHRINK;
    int64_t until_keep = a.thin, row = 0;

    for (int64_t s = 0; s < a.n_steps; ++s) {
        if (!(px > -INFINITY)) {  // also catches NaN
            err |= GSSS_CHAIN_NONFINITE;
            break;
        }
        dr.begin_step(a.step_offset + (uint64_t)s);

        // u = spherical_projection(z, x)   (sphere.py:29-33)
        double u[V::N];
        double u_thr, u_th0;
        if (Draws::kTangent3 && d == 3) {  // Philox stream on S^2: one block for the step, the unit tangent drawn directly (tangent3)
            uint32_t w_phi = 0u;
            dr.step_s2(u_thr, u_th0, w_phi);
            const double nx = sqrt(vdot<V>(x, x)) + 1e-100;
            double nrm[V::N];
#pragma unroll
            for (int i = 0; i < V::N; ++i) nrm[i] = x[i] / nx;
            dr.tangent(nrm, u, g, w_phi);
        } else {
            dr.normals(u, g);  // u holds z for now
            const double nx = sqrt(vdot<V>(x, x)) + 1e-100;
            double nrm[V::N];
#pragma unroll
            for (int i = 0; i < V::N; ++i) nrm[i] = x[i] / nx;
            const double cz = vdot<V>(u, nrm);
#pragma unroll
            for (int i = 0; i < V::N; ++i) u[i] = fma(-cz, nrm[i], u[i]);
            const double nw = sqrt(vdot<V>(u, u)) + 1e-100;
#pragma unroll
            for (int i = 0; i < V::N; ++i) u[i] = u[i] / nw;
            dr.step_uniforms(u_thr, u_th0, shrink);
        }
        const double thr = px + fm::log_fast(u_thr);  // mcmc.py:389
        double lo, hi;
        if (shrink) {
            hi = kTwoPi * u_th0;             // mcmc.py:391
            lo = hi - kTwoPi;                // mcmc.py:392
        } else {
            lo = 0.0;                        // mcmc.py:367
            hi = kTwoPi;
        }

        int t = 0;
        int step_err = 0;
        for (;;) {
            if (t >= a.max_tries) {
                step_err = GSSS_CHAIN_MAX_TRIES;
                break;
            }
            const double theta = lo + (hi - lo) * dr.next_try();  // mcmc.py:395
            double sn, cs;
            fm::sincos_small(theta, sn, cs);  // |theta| <= 2 pi by construction of the bracket
            double y[V::N];
#pragma unroll
            for (int i = 0; i < V::N; ++i) y[i] = fma(sn, u[i], cs * x[i]);  // mcmc.py:396
            const double py = tgt.logp(y, g, scratch);
            ++t;
            if (py > thr) {                                       // mcmc.py:397
#pragma unroll
                for (int i = 0; i < V::N; ++i) x[i] = y[i];
                px = py;
                if (Draws::kReplay && dr.exhausted) step_err = GSSS_CHAIN_REPLAY_EXHAUSTED;
                break;
            }
            if (shrink) {                                         // mcmc.py:400
                if (theta < 0.0)
                    lo = theta;
                else
                    hi = theta;
            }
            if (Draws::kReplay && dr.exhausted) {
                step_err = GSSS_CHAIN_REPLAY_EXHAUSTED;
                break;
            }
        }
        n_try += t;
        n_rej += step_err ? t : t - 1;
        err |= step_err;

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
                if (active && !step_err) {  // (group layouts: `active` and `step_err` are the same in every lane of a group)
                    if constexpr (V::L == 1)
                        stats_update<V::N>(a, c, x);
                    else
                        stats_update_group<V>(a, c, g, d, x);
                }
            }
            ++row;
        }
        if (step_err) break;
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < V::N; ++i) {
            const int cc = V::comp(g, i);
            if (cc < d) a.state[(size_t)cc * n + c] = x[i];
        }
        if (g == 0) {
            if (a.n_reject) a.n_reject[c] += n_rej;
            if (a.n_tries) a.n_tries[c] += n_try;
            if (a.err && err) a.err[c] |= err;
        }
    }
    dr.finish(a, c, active && g == 0);
}

// Distribution.log_prob (GRAD: Distribution.gradient, rows [n][d]) for rows of a row-major [n][d] array
template <class V, template <class> class TT, bool GRAD>
__global__ void __launch_bounds__(kBlock) logprob_kernel(TargetBlock tb, const double *__restrict__ xin, int64_t n,
                                                         double *__restrict__ out)
{
    using T = TT<V>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    T tgt;
    tgt.stage(lds, tb);
    double *scratch = lds + T::lds_doubles(tb.k, tb.d) + (size_t)T::kScratchPerChain * (threadIdx.x / V::L);
    __syncthreads();
    const int d = tb.d;
    const int g = threadIdx.x % V::L;
    const int64_t c_raw = (int64_t)blockIdx.x * (kBlock / V::L) + threadIdx.x / V::L;
    const bool active = c_raw < n;
    const int64_t c = active ? c_raw : n - 1;
    double x[V::N];
#pragma unroll
    for (int i = 0; i < V::N; ++i) {
        const int cc = V::comp(g, i);
        x[i] = (cc < d) ? xin[(size_t)c * d + cc] : 0.0;
    }
    if constexpr (GRAD) {
        double gr[V::N];
        tgt.grad(x, g, scratch, gr);
#pragma unroll
        for (int i = 0; i < V::N; ++i) {
            const int cc = V::comp(g, i);
            if (active && cc < d) out[(size_t)c * d + cc] = gr[i];
        }
    } else {
        const double p = tgt.logp(x, g, scratch);
        if (active && g == 0) out[c] = p;
    }
}

}  // namespace gsss

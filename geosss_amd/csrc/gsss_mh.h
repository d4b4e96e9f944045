// gsss_mh.h -- the baselines the paper compares the slice samplers with, as many-chain kernels:
//   MetropolisHastings  (random-walk MH on the sphere)   geosss/mcmc.py:118-176
//   SphericalHMC        (leapfrog on the sphere)          geosss/mcmc.py:236-332
//   AdaptiveStepsize    (x 1.02 / x 0.98 during burn-in)  geosss/mcmc.py:80-115
//   IndependenceSampler, MixtureRWMHIndependenceSampler    geosss/mcmc.py:179-234 (the RWMH kernel with another proposal)
// One lane group per chain as in run_kernel (gsss_device.h); a transition has a fixed amount of work, so
// the wavefront never diverges (numpy's gamma stream excepted).  log_prob / gradient are evaluated from the
// point itself, operation by operation as the reference does (the targets' logp / grad functors).
#pragma once
#include "gsss_launch.h"

namespace gsss {

struct MhBlock {
    double *stepsize;     // [n_chains] in/out
    int64_t *n_accept;    // [n_chains] or NULL, ADDED to
    double *momenta;      // [d][n_chains] or NULL: HMC, the momenta the reference keeps in the second half of its state
    int64_t adapt_steps;  // the first adapt_steps steps of this launch adapt the stepsize
    int32_t n_leapfrog;
    int32_t kind;         // GSSS_RWMH kernels: GSSS_RWMH, GSSS_INDEP or GSSS_MIX (the proposal is chosen at run time)
    double mix_alpha;     // GSSS_MIX: probability of the RWMH proposal
    int64_t *adapt_left;  // GSSS_MIX: [n_chains] in/out, RWMH proposals that still adapt the stepsize
    int64_t *n_rwmh;      // GSSS_MIX: [n_chains] or NULL, ADDED to
    double *momenta_samples;  // HMC: NULL or the momenta of the retained rows, laid out like RunBlock::samples (mcmc.py:321-332)
    double *stepsize_trace;   // RWMH kernels: NULL or [n_steps][n_chains], the stepsize after every RWMH proposal, NaN otherwise (mcmc.py:228)
};

template <class V, template <class> class TT, template <class> class DR, int SAMPLER>
__global__ void __launch_bounds__(kBlock) mh_kernel(TargetBlock tb, RunBlock a, MhBlock m)
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
    const int64_t c_raw = (int64_t)blockIdx.x * (kBlock / V::L) + threadIdx.x / V::L;
    const bool active = c_raw < n;
    const int64_t c = active ? c_raw : n - 1;  // surplus lanes shadow a chain and store nothing

    double x[V::N], v[V::N];
#pragma unroll
    for (int i = 0; i < V::N; ++i) {
        const int cc = V::comp(g, i);
        x[i] = (cc < d) ? a.state[(size_t)cc * n + c] : 0.0;
        v[i] = (cc < d && m.momenta) ? m.momenta[(size_t)cc * n + c] : 0.0;
    }
    dr.init(a, c, d);
    double eps = m.stepsize[c];
    int64_t n_acc = 0, until_keep = a.thin, row = 0;
    int64_t adapt_left = (SAMPLER == GSSS_RWMH && m.kind == GSSS_MIX && m.adapt_left) ? m.adapt_left[c] : 0, n_rwmh = 0;
    int err = 0;

    for (int64_t s = 0; s < a.n_steps; ++s) {
        dr.begin_step(a.step_offset + (uint64_t)s);
        bool accepted;
        double y[V::N];
        bool use_rwmh = true;
        if (SAMPLER == GSSS_RWMH) {
            if (m.kind == GSSS_MIX) use_rwmh = dr.mix_uniform() < m.mix_alpha;  // mcmc.py:213
            if (m.kind == GSSS_INDEP) use_rwmh = false;
            double r = 0.0;
            if (use_rwmh) r = dr.chi(g);           // mcmc.py:143: r = sqrt(2 gamma(d / 2))
            double z[V::N];
            dr.normals(z, g);                      // mcmc.py:144 / :181 (the independence proposal is the projected normal vector)
#pragma unroll
            for (int i = 0; i < V::N; ++i) y[i] = use_rwmh ? r * x[i] + eps * z[i] : z[i];
            const double nrm = sqrt(vdot<V>(y, y)) + 1e-100;  // mcmc.py:145, sphere.py:10-18
#pragma unroll
            for (int i = 0; i < V::N; ++i) y[i] = y[i] / nrm;
            const double prob = tgt.logp(y, g, scratch) - tgt.logp(x, g, scratch);  // mcmc.py:152
            accepted = log(dr.accept_uniform()) < prob;                              // mcmc.py:153
        } else {
            double vv[V::N], gr[V::N];
            dr.normals(v, g);                      // mcmc.py:278: v = project(standard_normal, x)
            {
                const double cx = vdot<V>(x, v);
#pragma unroll
                for (int i = 0; i < V::N; ++i) v[i] = v[i] - x[i] * cx;
            }
            const double h0 = 0.5 * vdot<V>(v, v) - tgt.logp(x, g, scratch);  // mcmc.py:285-286
            auto projected_gradient = [&](const double (&at)[V::N]) {  // project(gradient(x), x), mcmc.py:231-235
                tgt.grad(at, g, scratch, gr);
                const double cg = vdot<V>(at, gr);
#pragma unroll
                for (int i = 0; i < V::N; ++i) gr[i] = gr[i] - at[i] * cg;
            };
#pragma unroll
            for (int i = 0; i < V::N; ++i) {
                y[i] = x[i];
                vv[i] = v[i];
            }
            projected_gradient(y);
#pragma unroll
            for (int i = 0; i < V::N; ++i) vv[i] = vv[i] + 0.5 * eps * gr[i];  // mcmc.py:301
            for (int l = 0; l < m.n_leapfrog; ++l) {
                const double norm = sqrt(vdot<V>(vv, vv));
                const double cs = cos(eps * norm), sn = sin(eps * norm);
#pragma unroll
                for (int i = 0; i < V::N; ++i) {
                    const double yi = y[i];
                    y[i] = yi * cs + (vv[i] / norm) * sn;   // mcmc.py:307
                    vv[i] = vv[i] * cs - (yi * norm) * sn;  // mcmc.py:308
                }
                projected_gradient(y);
                const double f = (l < m.n_leapfrog - 1) ? eps : 0.5 * eps;  // mcmc.py:310-313
#pragma unroll
                for (int i = 0; i < V::N; ++i) vv[i] = vv[i] + f * gr[i];
            }
            {
                const double nrm = sqrt(vdot<V>(y, y)) + 1e-100;  // mcmc.py:315
#pragma unroll
                for (int i = 0; i < V::N; ++i) y[i] = y[i] / nrm;
            }
            const double h1 = 0.5 * vdot<V>(vv, vv) - tgt.logp(y, g, scratch);
            accepted = log(dr.accept_uniform()) < h0 - h1;        // mcmc.py:318-319
            if (accepted) {
#pragma unroll
                for (int i = 0; i < V::N; ++i) v[i] = vv[i];
            }
        }
        if (accepted) {
#pragma unroll
            for (int i = 0; i < V::N; ++i) x[i] = y[i];
        }
        n_acc += accepted ? 1 : 0;
        if (SAMPLER == GSSS_RWMH && m.kind == GSSS_MIX) {         // mcmc.py:226-228: only RWMH proposals adapt, and only
            if (use_rwmh) {                                       // they advance the burn-in counter (:113-115)
                ++n_rwmh;
                if (adapt_left > 0) {
                    eps *= accepted ? 1.02 : 0.98;
                    --adapt_left;
                }
            }
        } else if (s < m.adapt_steps) {
            eps *= accepted ? 1.02 : 0.98;                        // mcmc.py:113-115
        }
        if (SAMPLER == GSSS_RWMH && m.stepsize_trace != nullptr && active && g == 0)
            m.stepsize_trace[(size_t)s * n + c] = use_rwmh ? eps : __builtin_nan("");  // mcmc.py:228 (after the adaptation)
        if (a.samples != nullptr && --until_keep == 0) {
            until_keep = a.thin;
            if (active) {
#pragma unroll
                for (int i = 0; i < V::N; ++i) {
                    const int cc = V::comp(g, i);
                    if (cc < d) {
                        a.samples[sample_index(a, row, cc, d, c)] = x[i];
                        if (SAMPLER == GSSS_HMC && m.momenta_samples) m.momenta_samples[sample_index(a, row, cc, d, c)] = v[i];
                    }
                }
            }
            ++row;
        }
        if (Draws::kReplay && dr.exhausted) {
            err |= GSSS_CHAIN_REPLAY_EXHAUSTED;
            break;
        }
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < V::N; ++i) {
            const int cc = V::comp(g, i);
            if (cc < d) {
                a.state[(size_t)cc * n + c] = x[i];
                if (m.momenta) m.momenta[(size_t)cc * n + c] = v[i];
            }
        }
        if (g == 0) {
            m.stepsize[c] = eps;
            if (m.n_accept) m.n_accept[c] += n_acc;
            if (SAMPLER == GSSS_RWMH && m.kind == GSSS_MIX) {
                if (m.adapt_left) m.adapt_left[c] = adapt_left;
                if (m.n_rwmh) m.n_rwmh[c] += n_rwmh;
            }
            if (a.err && err) a.err[c] |= err;
        }
    }
    dr.finish(a, c, active && g == 0);
}

template <class V, template <class> class TT, template <class> class DR, int SAMPLER>
int do_mh(const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st)
{
    using T = TT<V>;
    const size_t lds = (T::lds_doubles(tb.k, tb.d) + scratch_doubles<V, T>() + DR<V>::kLdsDoubles) * sizeof(double);
    if (lds > 160 * 1024) {
        set_error("target parameters need %zu B of LDS", lds);
        return GSSS_E_UNSUPPORTED;
    }
    auto kern = mh_kernel<V, TT, DR, SAMPLER>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GSSS_E_HIP;
        }
    }
    const int64_t per_block = kBlock / V::L;
    const int64_t grid = (rb.n_chains + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, tb, rb, mb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("MH kernel launch failed: %s", hipGetErrorString(e));
        return GSSS_E_HIP;
    }
    return GSSS_OK;
}

template <template <class> class TT>
int launch_mh(int vec_id, int draws, int sampler, const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st);

// one layout: every draw source x both samplers
template <class V, template <class> class TT>
int mh_dispatch(int draws, int sampler, const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st)
{
    if (sampler == GSSS_RWMH || sampler == GSSS_INDEP || sampler == GSSS_MIX) {  // one kernel, the proposal chosen at run time (mb.kind)
        if (draws == kDrawsReplay) return do_mh<V, TT, ReplayDraws, GSSS_RWMH>(tb, rb, mb, st);
        if (draws == kDrawsNumpy) return do_mh<V, TT, NumpyDraws, GSSS_RWMH>(tb, rb, mb, st);
        return do_mh<V, TT, PhiloxDraws, GSSS_RWMH>(tb, rb, mb, st);
    }
    if (draws == kDrawsReplay) return do_mh<V, TT, ReplayDraws, GSSS_HMC>(tb, rb, mb, st);
    if (draws == kDrawsNumpy) return do_mh<V, TT, NumpyDraws, GSSS_HMC>(tb, rb, mb, st);
    return do_mh<V, TT, PhiloxDraws, GSSS_HMC>(tb, rb, mb, st);
}

#define GSSS_DEFINE_MH_LAUNCHER(TT)                                                                              \
    template <>                                                                                                  \
    int launch_mh<TT>(int vec_id, int draws, int sampler, const TargetBlock &tb, const RunBlock &rb,             \
                      const MhBlock &mb, hipStream_t st)                                                         \
    {                                                                                                            \
        switch (vec_id) {                                                                                        \
            GSSS_VEC_LIST(GSSS_MH_CASE_##TT)                                                                     \
        }                                                                                                        \
        set_error("unknown vector layout %d", vec_id);                                                           \
        return GSSS_E_INVALID;                                                                                   \
    }

}  // namespace gsss

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the emitted .npz files.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

What is recorded:

* traj_<case>.npz   one chain of the reference ShrinkageSphericalSliceSampler /
                    RejectionSphericalSliceSampler (geosss/mcmc.py:335-401) with every
                    RNG draw captured in consumption order, the state after every step,
                    the per-step threshold, the number of tries per step, the log-density
                    of every proposal, and the smallest |p(y)-threshold| margin.
* logprob_kat.npz   pdf.log_prob known-answer tables (1-D and 2-D call) per target.
* geometry_kat.npz  sphere.radial/orthogonal/spherical_projection, distance_slerp,
                    SlerpCurve.find_nearest on random (also non-unit) inputs.
* stats_<case>.npz  statistical pins from several independent reference chains
                    (rejections/step, mode occupancy, mean geodesic step, moments).

The RNG proxy: `uniform(lo, hi)` is answered as `lo + (hi-lo)*random()`, which is
bitwise what numpy's Generator.uniform does; the generator asserts that the proxied chain
equals the unpatched chain bit for bit before writing anything.
"""
import os
import sys
import time

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import geosss as gs  # noqa: E402
from geosss import sphere as rsphere  # noqa: E402
from geosss.distributions import CurvedVonMisesFisher  # noqa: E402
from geosss.spherical_curve import SlerpCurve, brownian_curve, distance_slerp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class RecordingRNG:
    """Stands in for sampler.rng; logs every double the sampler consumes."""

    def __init__(self, seed):
        self.g = np.random.default_rng(seed)
        self.draws = []

    def standard_normal(self, n):
        z = self.g.standard_normal(n)
        self.draws.extend(z.tolist())
        return z

    def random(self):
        u = self.g.random()
        self.draws.append(u)
        return u

    def uniform(self, lo, hi):
        u = self.g.random()
        self.draws.append(u)
        return lo + (hi - lo) * u

    def gamma(self, shape):
        v = self.g.gamma(shape)
        self.draws.append(v)
        return v


class LoggingTarget:
    """Wraps pdf.log_prob so every evaluation made by the sampler is logged."""

    def __init__(self, pdf):
        self.pdf = pdf
        self.vals = []

    def log_prob(self, x):
        v = self.pdf.log_prob(x)
        self.vals.append(float(v))
        return v


def record_trajectory(cls, pdf, x0, seed, n_steps):
    # unpatched chain (truth)
    s0 = cls(pdf, np.array(x0, dtype=float), seed)
    truth = [np.array(s0.state)]
    for _ in range(n_steps):
        truth.append(np.copy(next(s0)))
    truth = np.array(truth)

    # proxied chain
    tgt = LoggingTarget(pdf)
    s = cls(tgt, np.array(x0, dtype=float), seed)
    s.rng = RecordingRNG(seed)
    states = [np.array(s.state)]
    step_draw_offset = [0]
    tries = []
    thr = []
    logp_state = []
    prop_logp = []
    prop_offset = [0]
    d = len(x0)
    for _ in range(n_steps):
        nv0 = len(tgt.vals)
        nd0 = len(s.rng.draws)
        y = next(s)
        states.append(np.copy(y))
        vals = tgt.vals[nv0:]
        draws = s.rng.draws[nd0:]
        # order of draws in a step: d normals, U(threshold), then the thetas
        u_thr = draws[d]
        logp_state.append(vals[0])
        thr.append(vals[0] + np.log(u_thr))
        tries.append(len(vals) - 1)
        prop_logp.extend(vals[1:])
        prop_offset.append(len(prop_logp))
        step_draw_offset.append(len(s.rng.draws))
    states = np.array(states)
    assert np.array_equal(states, truth), "proxy RNG changed the chain"
    prop_logp = np.array(prop_logp)
    thr = np.array(thr)
    margin = np.inf
    for i in range(n_steps):
        seg = prop_logp[prop_offset[i]:prop_offset[i + 1]]
        margin = min(margin, np.min(np.abs(seg - thr[i])))
    return dict(
        states=states,
        draws=np.array(s.rng.draws),
        step_draw_offset=np.array(step_draw_offset, dtype=np.int64),
        tries=np.array(tries, dtype=np.int64),
        threshold=thr,
        logp_state=np.array(logp_state),
        prop_logp=prop_logp,
        prop_offset=np.array(prop_offset, dtype=np.int64),
        n_reject=np.int64(s.n_reject),
        min_margin=np.float64(margin),
        seed=np.int64(seed),
    )


# ----------------------------------------------------------------------------- targets


def target_params(pdf):
    """Flatten a reference pdf into the plain arrays our C-ABI takes."""
    from geosss.registration import CoherentPointDrift, GaussianMixtureModel
    if isinstance(pdf, GaussianMixtureModel):  # registration.py:62-293 (CoherentPointDrift is a subclass)
        return dict(kind="cpd", source=np.array(pdf.source.positions, dtype=float), source_w=np.array(pdf.source.weights, dtype=float),
                    target=np.array(pdf.target.positions, dtype=float), target_w=np.array(pdf.target.weights, dtype=float),
                    sigma=np.float64(pdf.sigma), k_nn=np.int64(pdf.k), beta=np.float64(pdf.beta),
                    omega=np.float64(getattr(pdf, "omega", 0.0)), outlier=np.bool_(isinstance(pdf, CoherentPointDrift)))
    if isinstance(pdf, gs.MixtureModel):
        mu = np.array([p.mu for p in pdf.pdfs])
        return dict(kind="vmf_mixture", mu=mu, weights=np.array(pdf.weights))
    if isinstance(pdf, gs.BinghamFisher):  # distributions.py:106-114
        return dict(kind="bingham", A=np.array(pdf.A, dtype=float), b=np.array(pdf.b, dtype=float))
    if isinstance(pdf, gs.Bingham):
        return dict(kind="bingham", A=np.array(pdf.A))
    if isinstance(pdf, CurvedVonMisesFisher):
        return dict(kind="curve_vmf", knots=np.array(pdf.curve.knots), kappa=np.float64(pdf.kappa))
    raise TypeError(pdf)


def readme_mixture():
    mus = np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
    return gs.MixtureModel([gs.VonMisesFisher(80.0 * mu) for mu in mus])


def mixture(d, K, kappa, weights=None):
    modes = gs.sphere.sample_sphere(d - 1, K, seed=1234)  # scripts/mixture_vMF.py:406-411
    return gs.MixtureModel([gs.VonMisesFisher(kappa * mu) for mu in modes], weights)


def curve_target(d, kappa):
    knots = brownian_curve(n_points=10, dimension=d, step_size=0.5, seed=4562)  # scripts/curve_vMF.py:577-589
    return CurvedVonMisesFisher(SlerpCurve(knots), kappa)


def cases():
    out = {}
    out["vmfmix_readme"] = (readme_mixture(), np.array([-0.86, 0.19, -0.47]), 3521, 1099)
    out["vmfmix_k10_kappa500"] = (mixture(3, 10, 500.0), gs.sphere.sample_sphere(2, seed=1345), 77, 400)
    out["vmfmix_d10_k5_kappa100"] = (mixture(10, 5, 100.0), gs.sphere.sample_sphere(9, seed=1345), 78, 300)
    out["vmfmix_d4_k4_weighted"] = (
        mixture(4, 4, 30.0, weights=[1.0, 2.0, 3.0, 0.5]), gs.sphere.sample_sphere(3, seed=5), 79, 300)
    b10 = gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982)  # scripts/bingham.py:131
    out["bingham_d10_vmax30"] = (b10, np.array(b10.mode), 80, 400)
    b5 = gs.random_bingham(d=5, vmax=20.0, vmin=-3.0, eigensystem=False, seed=11)
    out["bingham_d5_dense"] = (b5, gs.sphere.sample_sphere(4, seed=6), 81, 300)
    b50 = gs.random_bingham(d=50, vmax=300.0, vmin=0.0, eigensystem=True, seed=6982)
    out["bingham_d50_vmax300"] = (b50, np.array(b50.mode), 82, 150)
    # tests/test_bingham_fisher.py:20-35 (Byrne & Girolami / Brubaker et al. examples)
    bf5 = gs.BinghamFisher(np.diag([-20.0, -10.0, 0.0, 10.0, 20.0]), np.array([40.0, 0.0, 0.0, 0.0, 0.0]))
    out["binghamfisher_d5"] = (bf5, gs.sphere.sample_sphere(4, seed=7), 83, 300)
    bf6 = gs.BinghamFisher(np.diag([-1000.0, -600.0, -200.0, 200.0, 600.0, 1000.0]),
                           np.array([100.0, 0.0, 0.0, 0.0, 0.0, 0.0]))
    out["binghamfisher_d6"] = (bf6, gs.sphere.sample_sphere(5, seed=8), 84, 300)
    # d = 17, 40, 100, 160: one chain for every lane-group shape of the group-speculative curve kernel
    # (gsss_fast_curvespec.hip: <4,2>, <4,3>, <8,4>, <16,3>) next to the BASELINE dimensions
    for d, kappa, n in ((3, 300.0, 300), (10, 800.0, 300), (10, 500.0, 200), (24, 800.0, 150),
                        (50, 800.0, 150), (200, 800.0, 100), (17, 800.0, 120), (40, 800.0, 100), (100, 800.0, 80),
                        (160, 800.0, 60)):
        out[f"curve_d{d}_kappa{int(kappa)}"] = (
            curve_target(d, kappa), gs.sphere.sample_sphere(d - 1, seed=1345), 90 + d, n)
    return out


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def flat_params(p):
    return {f"target_{k}": (np.array(v) if not isinstance(v, str) else np.array(v)) for k, v in p.items()}


def make_trajectories(only=None):
    for name, (pdf, x0, seed, n) in cases().items():
        if only and name not in only:
            continue
        t0 = time.time()
        rec = record_trajectory(gs.ShrinkageSphericalSliceSampler, pdf, x0, seed, n)
        print(f"{name}: {n} steps, rej/step={rec['n_reject'] / n:.3f}, "
              f"min margin={rec['min_margin']:.3e}, {time.time() - t0:.1f}s")
        save(f"traj_{name}.npz", x0=np.array(x0), sampler=np.array("shrink"), **flat_params(target_params(pdf)), **rec)
    if only:
        return
    # rejection sampler (mcmc.py:357-374): README target and the small Bingham
    for name, n in (("vmfmix_readme", 60), ("bingham_d10_vmax30", 150)):
        pdf, x0, seed, _ = cases()[name]
        rec = record_trajectory(gs.RejectionSphericalSliceSampler, pdf, x0, seed, n)
        print(f"reject_{name}: {n} steps, rej/step={rec['n_reject'] / n:.3f}, min margin={rec['min_margin']:.3e}")
        save(f"traj_reject_{name}.npz", x0=np.array(x0), sampler=np.array("reject"),
             **flat_params(target_params(pdf)), **rec)

    # The README call itself: cls(pdf, init, seed).sample(1000, 100)  (README.md:44-64)
    pdf = readme_mixture()
    pdf.log_prob.reset_counters()
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.array([-0.86, 0.19, -0.47]), 3521)
    samples = s.sample(1000, 100)
    save("readme_sample_call.npz", samples=samples, n_reject=np.int64(s.n_reject),
         num_calls=np.int64(pdf.log_prob.num_calls))


def make_logprob_kat():
    rng = np.random.default_rng(20260101)
    arrays = {}
    for name, (pdf, x0, _, _) in cases().items():
        d = len(x0)
        X = rsphere.radial_projection(rng.standard_normal((256, d)))
        X[200:] *= rng.uniform(0.5, 1.5, size=(56, 1))  # non-unit rows: the chain state is never renormalised
        lp2 = np.asarray(pdf.log_prob(X), dtype=float)
        lp1 = np.array([float(pdf.log_prob(x)) for x in X])
        assert np.allclose(lp1, lp2, rtol=0, atol=1e-9)
        arrays[f"{name}__X"] = X
        arrays[f"{name}__logp"] = lp1
        arrays[f"{name}__logp_batched"] = lp2
        for k, v in flat_params(target_params(pdf)).items():
            arrays[f"{name}__{k}"] = v
    save("logprob_kat.npz", **arrays)


def make_geometry_kat():
    rng = np.random.default_rng(424242)
    arrays = {}
    for d in (3, 10, 50):
        X = rng.standard_normal((64, d))  # pole, not unit
        Z = rng.standard_normal((64, d))
        arrays[f"d{d}_x"] = X
        arrays[f"d{d}_z"] = Z
        arrays[f"d{d}_radial"] = np.array([rsphere.radial_projection(x) for x in X])
        arrays[f"d{d}_ortho"] = np.array([rsphere.orthogonal_projection(z, x) for z, x in zip(Z, X)])
        arrays[f"d{d}_spherical"] = np.array([rsphere.spherical_projection(z, x) for z, x in zip(Z, X)])
        # slerp distance incl. clip edges: a few queries beyond either end of the arc
        a = rsphere.radial_projection(rng.standard_normal((64, d)))
        b = rsphere.radial_projection(a + 0.6 * rng.standard_normal((64, d)))
        q = rsphere.radial_projection(rng.standard_normal((64, d)))
        q[:16] = rsphere.radial_projection(a[:16] + 0.3 * (a[:16] - b[:16]))   # t < 0 side
        q[16:32] = rsphere.radial_projection(b[16:32] + 0.3 * (b[16:32] - a[16:32]))  # t > theta side
        q[32:40] = rsphere.radial_projection(0.4 * a[32:40] + 0.6 * b[32:40])  # on the arc
        dist, near = zip(*[distance_slerp(x, aa, bb) for x, aa, bb in zip(q, a, b)])
        arrays[f"d{d}_slerp_a"] = a
        arrays[f"d{d}_slerp_b"] = b
        arrays[f"d{d}_slerp_q"] = q
        arrays[f"d{d}_slerp_dist"] = np.array(dist)
        arrays[f"d{d}_slerp_near"] = np.array(near)
        knots = brownian_curve(n_points=10, dimension=d, step_size=0.5, seed=4562)
        curve = SlerpCurve(knots)
        arrays[f"d{d}_knots"] = knots
        arrays[f"d{d}_nearest"] = np.array([curve.find_nearest(x) for x in q])
    save("geometry_kat.npz", **arrays)


def make_tangent_kat():
    """What the set-up of a step draws on S^2, from the reference: pairs (x, u = sphere.spherical_projection(z, x)) with
    z ~ N(0, I_3) (mcmc.py:387, sphere.py:29-33) -- x unit, non-unit (|x| = 0.998: the sampler never renormalises its
    state) and at the two poles, where the tangent basis of the device's direct draw changes its formula.  The GPU test
    expresses these u in the device's tangent basis at x and holds their angles to the uniform law the library's
    one-angle draw assumes.  10^5 pairs; x is stored for the first 8 * 10^4 only (the poles are +-e_z)."""
    rng = np.random.default_rng(20261004)
    n_unit, n_short, n_pole = 60000, 20000, 10000
    x = rsphere.radial_projection(rng.standard_normal((n_unit + n_short, 3)))
    x[n_unit:] *= 0.998
    xs = np.concatenate([x, np.tile([0.0, 0.0, 1.0], (n_pole, 1)), np.tile([0.0, 0.0, -1.0], (n_pole, 1))])
    z = rng.standard_normal(xs.shape)
    u = np.array([rsphere.spherical_projection(zi, xi) for zi, xi in zip(z, xs)])  # one call per pair, as the sampler makes it
    assert np.max(np.abs(np.linalg.norm(u, axis=1) - 1.0)) < 1e-14
    assert np.max(np.abs(np.sum(u * xs, axis=1))) < 1e-12  # (cancellation when z is nearly parallel to x)
    save("tangent_kat.npz", x=x, u=u, n_unit=np.int64(n_unit), n_short=np.int64(n_short), n_pole=np.int64(n_pole))


def _stat_chain(args):
    name, seed_state, n_steps, burn = args
    pdf, x0, _, _ = cases()[name]
    ss = np.random.SeedSequence(entropy=seed_state[0], spawn_key=seed_state[1])
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.array(x0), ss)
    s.sample(burn + 1, burnin=0)          # `burn` transitions
    rej_burn = s.n_reject
    X = s.sample(n_steps, burnin=0)       # row 0 = the state after burn-in, n_steps - 1 further transitions: the same chain as
    rej_total = s.n_reject                # one sample(n_steps + burn) call, rows [burn:]
    geo = np.arccos(np.clip(np.sum(X[1:] * X[:-1], axis=-1), -1, 1))
    from geosss.utils import IAT
    out = dict(iat=np.array([IAT(X[:, j]) for j in range(X.shape[1])]),  # per chain and coordinate, utils.py:119-131
               mean=X.mean(0), rej_per_step=rej_total / (n_steps + burn - 1),
               rej_per_step_post=(rej_total - rej_burn) / (n_steps - 1),  # rejections per step after burn-in only
               geo_step=geo.mean(), logp_mean=np.mean(pdf.log_prob(X)) if name.startswith(("vmf", "bing")) else np.nan)
    if X.shape[1] <= 16:  # (d (d + 1) / 2 numbers per chain: left out for the d = 50 / 200 curves)
        out["second"] = (X[:, :, None] * X[:, None, :]).mean(0)
    if name.startswith("vmfmix"):
        modes = np.array([p.mu / np.linalg.norm(p.mu) for p in pdf.pdfs])
        occ = np.bincount(np.argmax(X @ modes.T, axis=1), minlength=len(modes)) / len(X)
        out["occupancy"] = occ
    return out


def make_stats(plan=None):
    from concurrent.futures import ProcessPoolExecutor

    # name -> (steps kept, burn-in steps before them)
    plan = plan or {"vmfmix_readme": (12000, 1200), "vmfmix_k10_kappa500": (8000, 800), "bingham_d10_vmax30": (60000, 6000),
                    "curve_d10_kappa800": (4000, 400)}
    for name, (n_steps, burn) in plan.items():
        t0 = time.time()
        seeds = np.random.SeedSequence(48385).spawn(8)  # scripts/bingham.py:87-88 pattern
        jobs = [(name, (s.entropy, s.spawn_key), n_steps, burn) for s in seeds]
        with ProcessPoolExecutor(8) as ex:
            res = list(ex.map(_stat_chain, jobs))
        arrays = {k: np.array([r[k] for r in res]) for k in res[0]}
        arrays["n_steps"] = np.int64(n_steps)
        arrays["burn"] = np.int64(burn)
        arrays["n_chains"] = np.int64(len(res))
        print(f"stats {name}: rej/step={arrays['rej_per_step'].mean():.3f} geo={arrays['geo_step'].mean():.3f} "
              f"({time.time() - t0:.0f}s)")
        save(f"stats_{name}.npz", **arrays)


def _time_chain(args):
    name, n_steps, seed = args
    pdf, x0, _, _ = cases()[name]
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.array(x0), seed)
    t0 = time.perf_counter()
    s.sample(n_steps + 1)
    return n_steps / (time.perf_counter() - t0), s.n_reject / n_steps


def make_timing():
    """The reference itself timed in THIS container: 1 process and one process per core."""
    import json
    import platform
    from concurrent.futures import ProcessPoolExecutor

    ncpu = os.cpu_count()
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    out = {"host": {"cpu": cpu[0] if cpu else platform.processor(), "cores": ncpu, "python": platform.python_version(),
                    "numpy": np.__version__}, "sampler": "geosss.ShrinkageSphericalSliceSampler", "targets": {}}
    import scipy
    out["host"]["scipy"] = scipy.__version__
    for name, n in (("vmfmix_readme", 6000), ("vmfmix_k10_kappa500", 3000), ("bingham_d10_vmax30", 40000),
                    ("curve_d10_kappa800", 1500), ("curve_d200_kappa800", 1200)):
        one, rej = _time_chain((name, n, 1))
        t0 = time.perf_counter()
        with ProcessPoolExecutor(ncpu) as ex:
            res = list(ex.map(_time_chain, [(name, n, 10 + i) for i in range(ncpu)]))
        wall = time.perf_counter() - t0
        out["targets"][name] = {"steps_per_s_1_process": one, "steps_per_s_per_process_all_cores": float(np.mean([r[0] for r in res])),
                                "steps_per_s_aggregate_all_cores": float(np.sum([r[0] for r in res])),
                                "steps_per_s_aggregate_incl_spawn": ncpu * n / wall, "rejections_per_step": rej,
                                "n_steps": n}
        print(name, out["targets"][name])
    with open(os.path.join(OUT, "cpu_reference_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


def make_diagnostics():
    """Known answers of the reference's diagnostics (utils.py:96-134 acf/acf_fft/IAT/n_eff,
    sphere.py:64-68 distance, scripts/bingham.py:23-25 hopping frequency, the mode-occupancy KL of
    scripts/vMF_diagnostics.py:335-342) on reference chains."""
    from geosss.utils import IAT, acf, acf_fft, n_eff
    pdf, x0, seed, _ = cases()["vmfmix_readme"]
    X = gs.ShrinkageSphericalSliceSampler(pdf, np.array(x0), 11).sample(4001)
    arrays = {"vmf_X": X}
    arrays["vmf_acf_fft"] = np.array([acf_fft(X[:, j]) for j in range(3)])
    arrays["vmf_acf"] = np.array([acf(X[:, j], 50) for j in range(3)])
    arrays["vmf_IAT"] = np.array([IAT(X[:, j]) for j in range(3)])
    arrays["vmf_IAT_200"] = np.array([IAT(X[:, j], 200) for j in range(3)])
    arrays["vmf_n_eff"] = np.array([n_eff(X[:, j]) for j in range(3)])
    arrays["vmf_distance"] = rsphere.distance(X[1:], X[:-1])
    modes = np.array([p.mu for p in pdf.pdfs])
    m = np.argmax(X @ modes.T, axis=1)
    i, c = np.unique(m, return_counts=True)
    p = np.full(len(modes), 1e-100)
    p[i] = c
    p[i] /= p.sum()
    arrays["vmf_modes"] = modes
    arrays["vmf_occupancy"] = p
    arrays["vmf_kl"] = p @ np.log(p / pdf.weights)
    b10, xb, _, _ = cases()["bingham_d10_vmax30"]
    Xb = gs.ShrinkageSphericalSliceSampler(b10, np.array(xb), 12).sample(3000)
    arrays["bingham_X"] = Xb
    arrays["bingham_mode"] = np.array(b10.mode)
    arrays["bingham_hop"] = np.mean(np.diff(np.sign(Xb @ b10.mode)) != 0.0)
    arrays["bingham_IAT"] = np.array([IAT(Xb[:, j]) for j in range(10)])
    save("diagnostics_kat.npz", **arrays)


def record_mh(kind, pdf, x0, seed, n_steps, burnin, stepsize, n_leapfrog=10, alpha=0.5):
    """One chain of the reference's MetropolisHastings (mcmc.py:118-176) or SphericalHMC (:236-332) with every draw
    recorded in consumption order (RWMH: gamma(d/2), d normals, 1 uniform; HMC: d normals, 1 uniform), the state,
    the accept flag and the adapted stepsize after every step (AdaptiveStepsize adapts during the first `burnin`
    steps, as Sampler.sample(n, burnin) arranges, mcmc.py:169-176)."""
    def build():
        if kind == "rwmh":
            return gs.MetropolisHastings(pdf, np.array(x0, dtype=float), seed, stepsize=stepsize)
        if kind == "indep":   # mcmc.py:179-182: d normals, 1 uniform per step
            return gs.mcmc.IndependenceSampler(pdf, np.array(x0, dtype=float), seed, stepsize=stepsize)
        if kind == "mix":     # mcmc.py:185-234: 1 uniform, then the chosen kernel's draws, 1 uniform
            return gs.mcmc.MixtureRWMHIndependenceSampler(pdf, np.array(x0, dtype=float), seed, stepsize=stepsize, mixing_probability=alpha)
        return gs.SphericalHMC(pdf, np.array(x0, dtype=float), seed, stepsize=stepsize, n_steps=n_leapfrog)
    d = len(x0)
    s0 = build()
    s0.reset(burnin)
    truth = [np.copy(next(s0))[:d] for _ in range(n_steps)]
    s = build()
    s.reset(burnin)
    s.rng = RecordingRNG(seed)
    states, accept, eps, offs = [np.array(x0, dtype=float)], [], [], [0]
    mom = [np.zeros(d)]                       # SphericalHMC: the momentum half of every row, mcmc.py:262, 321-332
    for _ in range(n_steps):
        n0 = s.n_accept
        y = np.copy(next(s))
        states.append(y[:d])
        mom.append(y[d:] if kind == "hmc" else np.zeros(d))
        accept.append(s.n_accept - n0)
        eps.append(s.stepsize)
        offs.append(len(s.rng.draws))
    states = np.array(states)
    assert np.array_equal(states[1:], np.array(truth)), "proxy RNG changed the chain"
    out = dict(states=states, accept=np.array(accept, dtype=np.int8), stepsize_trace=np.array(eps),
               draws=np.array(s.rng.draws), step_draw_offset=np.array(offs, dtype=np.int64),
               n_accept=np.int64(s.n_accept), seed=np.int64(seed), burnin=np.int64(burnin),
               stepsize0=np.float64(stepsize), n_leapfrog=np.int64(n_leapfrog), sampler=np.array(kind))
    if kind == "hmc":
        out["momenta"] = np.array(s.state[d:])
        out["momenta_trace"] = np.array(mom)   # what sample(return_momenta=True) returns next to the positions
    if kind == "mix":
        per_step = np.diff(np.array(offs))
        out.update(use_rwmh=(per_step == d + 3).astype(np.int8), rwmh_counter=np.int64(s.rwmh_counter),
                   indep_counter=np.int64(s.indep_counter), rwmh_stepsize_vals=np.array(s.rwmh_stepsize_vals),
                   alpha=np.float64(alpha))
        assert int(out["use_rwmh"].sum()) == s.rwmh_counter
    return out


def make_mh():
    """RWMH / spherical HMC reference chains + gradient known answers."""
    rng = np.random.default_rng(77)
    plan = [("vmfmix_readme", 600, 200, 0.1), ("bingham_d10_vmax30", 500, 150, 0.1), ("curve_d10_kappa800", 300, 100, 0.1),
            ("vmfmix_d10_k5_kappa100", 300, 100, 0.1), ("bingham_d5_dense", 300, 100, 0.1), ("binghamfisher_d5", 300, 100, 0.05),
            ("curve_d50_kappa800", 120, 40, 0.05)]
    for name, n, burn, eps in plan:
        pdf, x0, seed, _ = cases()[name]
        for kind in ("rwmh", "hmc"):
            rec = record_mh(kind, pdf, x0, seed + 1000, n, burn, eps)
            print(f"{kind}_{name}: {n} steps, accept rate {rec['n_accept'] / n:.3f}, final stepsize {rec['stepsize_trace'][-1]:.4f}")
            d = len(x0)
            X = rsphere.radial_projection(rng.standard_normal((64, d)))
            grad = np.array([pdf.gradient(x) for x in X])
            save(f"mh_{kind}_{name}.npz", x0=np.array(x0), grad_X=X, grad=grad, **flat_params(target_params(pdf)), **rec)


def make_mh_kernels():
    """IndependenceSampler / MixtureRWMHIndependenceSampler reference chains (mcmc.py:179-234)."""
    plan = [("vmfmix_readme", 600, 200, 0.1, 0.5), ("bingham_d10_vmax30", 500, 150, 0.1, 0.7), ("curve_d10_kappa800", 400, 150, 0.1, 0.5),
            ("bingham_d5_dense", 300, 100, 0.1, 0.3)]
    for name, n, burn, eps, alpha in plan:
        pdf, x0, seed, _ = cases()[name]
        for kind in ("indep", "mix"):
            rec = record_mh(kind, pdf, x0, seed + 2000, n, burn, eps, alpha=alpha)
            extra = f", {int(rec['rwmh_counter'])} RWMH proposals" if kind == "mix" else ""
            print(f"{kind}_{name}: {n} steps, accept rate {rec['n_accept'] / n:.3f}, final stepsize {rec['stepsize_trace'][-1]:.4f}{extra}")
            save(f"mh_{kind}_{name}.npz", x0=np.array(x0), **flat_params(target_params(pdf)), **rec)


def cpd_cases():
    """Registration targets (geosss/registration.py) on unit quaternions: the protein example of
    scripts/protein_reg3d3d.py:561-594 (data/protein_registration.npz: 214 + 214 points, sigma 1, k 20, omega 0.4), the
    3D-2D toy of tests/test_cpd.py:79-107 with seeded noise, and a GaussianMixtureModel (no outlier term)."""
    from geosss.pointcloud import PointCloud, RotationMatrix, RotationProjection
    from geosss.registration import CoherentPointDrift, GaussianMixtureModel
    data = np.load(os.path.join(REF, "data", "protein_registration.npz"))
    n = len(data["source"])
    prot_t, prot_s = PointCloud(data["target"]), PointCloud(data["source"], np.full(n, 1 / n))
    out = {"cpd_protein": CoherentPointDrift(prot_t, prot_s, sigma=float(data["sigma"]), k=20, omega=float(data["prob_outlier"]))}
    out["gmm_protein_k10"] = GaussianMixtureModel(prot_t, prot_s, sigma=2.0, k=10, beta=0.5)
    rng = np.random.default_rng(5)
    cube = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
    src = RotationProjection(cube)
    R_true = RotationMatrix().rotation3d(euler_angles=np.array([0.2, 0.3, 0.1]))
    tgt = src.transform_positions(R_true) + rng.normal(0, 0.05, (8, 2))
    tgt = np.vstack([tgt, rng.uniform(-3, 3, (4, 2))])
    out["cpd_cube_3d2d"] = CoherentPointDrift(PointCloud(tgt), src, sigma=0.5, k=8, omega=0.2)
    return out


def make_cpd():
    rng = np.random.default_rng(99)
    for name, pdf in cpd_cases().items():
        Q = rsphere.radial_projection(rng.standard_normal((48, 4)))
        Q[40:] *= rng.uniform(0.5, 1.5, size=(8, 1))                    # non-unit quaternions: quat2matrix normalises
        logp = np.array([float(pdf.log_prob(q)) for q in Q])
        x0 = rsphere.radial_projection(np.random.default_rng(7).standard_normal(4))
        n = 40 if "protein" in name else 150
        rec = record_trajectory(gs.ShrinkageSphericalSliceSampler, pdf, x0, 321, n)
        print(f"{name}: {n} slice steps, rej/step={rec['n_reject'] / n:.2f}, min margin={rec['min_margin']:.2e}")
        save(f"traj_{name}.npz", x0=x0, sampler=np.array("shrink"), kat_q=Q, kat_logp=logp,
             **flat_params(target_params(pdf)), **rec)
        GX = rsphere.radial_projection(rng.standard_normal((24, 4)))
        grad = np.array([pdf.gradient(q) for q in GX])                  # Registration.gradient, registration.py:55-60
        mh = record_mh("rwmh", pdf, x0, 654, 80 if "protein" in name else 200, 30, 0.1)
        print(f"rwmh_{name}: accept rate {mh['n_accept'] / len(mh['accept']):.2f}")
        save(f"mh_rwmh_{name}.npz", x0=x0, grad_X=GX, grad=grad, **flat_params(target_params(pdf)), **mh)
        hm = record_mh("hmc", pdf, x0, 655, 10 if "protein" in name else 120, 4 if "protein" in name else 10, 1e-3 if "protein" in name else 0.05)
        print(f"hmc_{name}: accept rate {hm['n_accept'] / len(hm['accept']):.2f}, final stepsize {hm['stepsize_trace'][-1]:.2e}")
        save(f"mh_hmc_{name}.npz", x0=x0, grad_X=GX, grad=grad, **flat_params(target_params(pdf)), **hm)


def make_helpers():
    """Known answers of the reference's helper functions either side of the sampler: geosss/sphere.py coordinate maps,
    great-circle interpolation / rotation, sample_subsphere / sample_marginal, the host-side densities of
    geosss/distributions.py (marginal vMF, multivariate normal, ACG) and the clipped elementary functions of utils.py."""
    from geosss import utils as rutils
    from geosss.distributions import ACG, MarginalVonMisesFisher, MixtureModel, MultivariateNormal
    rng = np.random.default_rng(20250607)
    a = {}
    X3 = rsphere.sample_sphere(2, 32, seed=11)
    a["X3"] = X3
    a["polar"] = rsphere.cartesian2polar(X3)
    a["sph_phi"], a["sph_theta"] = rsphere.cartesian2spherical(X3)
    a["angles"] = rng.uniform(0, 2 * np.pi, 16)
    a["polar2cart"] = rsphere.polar2cartesian(a["angles"])
    a["theta"] = rng.uniform(0, np.pi, 16)
    a["sph2cart"] = rsphere.spherical2cartesian(a["angles"], a["theta"])
    for d in (4, 10):
        v = rsphere.sample_sphere(d - 1, seed=7 + d)
        a[f"d{d}_pole"] = v
        a[f"d{d}_subsphere"] = rsphere.sample_subsphere(v, seed=3)
        a[f"d{d}_marginal"] = rsphere.sample_marginal(d, size=8, seed=5)
        u = rsphere.sample_subsphere(v, seed=9)
        x = rsphere.sample_sphere(d - 1, seed=21 + d)
        a[f"d{d}_u"], a[f"d{d}_x"] = u, x
        tangent = 0.7 * rng.standard_normal(d)
        a[f"d{d}_tangent"] = tangent
        a[f"d{d}_wrap"] = rsphere.wrap(tangent, u, v)
        phis = rng.uniform(0, 1.2, 6)
        a[f"d{d}_phis"] = phis
        a[f"d{d}_slerp"] = np.array([rsphere.slerp(v, x)(p) for p in phis])
        a[f"d{d}_givens"] = np.array([rsphere.givens(u, v, x)(p) for p in phis])
        a[f"d{d}_distance"] = rsphere.distance(np.stack([u, v, x]), np.stack([x, x, x]))
    grid = np.linspace(-0.999, 0.999, 41)
    a["grid"] = grid
    for d, kappa in ((3, 80.0), (10, 100.0)):
        mus = kappa * rsphere.sample_sphere(d - 1, 3, seed=1234)
        a[f"marg_d{d}_mus"] = mus
        a[f"marg_d{d}_prob"] = np.array([[MarginalVonMisesFisher(i, mu).prob(grid) for i in range(d)] for mu in mus])
        a[f"marg_d{d}_log_prob"] = np.array([[MarginalVonMisesFisher(i, mu).log_prob(grid) for i in range(d)] for mu in mus])
        a[f"marg_d{d}_mixture"] = np.array([MixtureModel([MarginalVonMisesFisher(i, mu) for mu in mus]).log_prob(grid)
                                            for i in range(d)])
    M = rng.standard_normal((5, 5))
    C = M @ M.T + 5 * np.eye(5)
    mu = rng.standard_normal(5)
    Y = rng.standard_normal((12, 5))
    a["mvn_C"], a["mvn_mu"], a["mvn_Y"] = C, mu, Y
    # the reference leaves `gradient` abstract on these two, so they cannot be instantiated as they stand: close them
    concrete = lambda cls: type(cls.__name__, (cls,), {"gradient": lambda self, x: None})
    a["mvn_log_prob"] = concrete(MultivariateNormal)(mu, C).log_prob(Y)
    a["acg_log_prob"] = concrete(ACG)(C).log_prob(Y)
    z = np.array([-1e4, -308.5, -1.0, 0.0, 1.0, 709.5, 1e4])
    a["clip_in"] = z
    a["clip_exp"] = rutils.exp(z)
    w = np.array([0.0, 1e-320, 1e-308, 1.0, 1e308, np.inf])
    a["clip_log_in"] = w
    a["clip_log"] = rutils.log(w)
    p = rng.dirichlet(np.ones(8))
    q = rng.dirichlet(np.ones(8))
    q[2] = 0.0
    a["kl_p"], a["kl_q"], a["kl"] = p, q, np.float64(rutils.relative_entropy(p, q))
    a["format_time"] = np.array([rutils.format_time(t) for t in (0.0, 3.2, 0.0123, 4.5e-5, 7e-8, 2e-11)])
    a["format_time_in"] = np.array([0.0, 3.2, 0.0123, 4.5e-5, 7e-8, 2e-11])
    from geosss.spherical_curve import constrained_brownian_curve
    for d in (3, 7):
        knots = constrained_brownian_curve(n_points=12, dimension=d, step_size=0.3, seed=77 + d)
        curve = SlerpCurve(knots)
        a[f"cbc_d{d}_knots"] = knots
        a[f"cbc_d{d}_theta"], a[f"cbc_d{d}_bins"], a[f"cbc_d{d}_widths"] = curve.theta, curve.bins, curve.widths
        ts = np.concatenate([[0.0, 1.0], rng.uniform(0, 1, 14), curve.bins[3:5]])
        a[f"cbc_d{d}_t"] = ts
        a[f"cbc_d{d}_points"] = curve(ts)
    # registration with a translation, after the reference's tests/test_cpd.py: the cube, a z-y-z rotation, noise, outliers
    from geosss.pointcloud import PointCloud, RotationMatrix, RotationProjection, matrix2quat
    from geosss.registration import CoherentPointDrift
    cube = np.array([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], dtype=float)
    R_true = RotationMatrix().rotation3d(np.array([0.2, 0.3, 0.1]))
    a["cpdt_R"], a["cpdt_q"] = R_true, matrix2quat(R_true)
    a["cpdt_euler_deg"] = np.array([10.0, 20.0, 30.0])
    a["cpdt_R_deg"] = RotationMatrix(degree=True).create(a["cpdt_euler_deg"])
    a["cpdt_R_2d"] = RotationMatrix().rotation2d(0.7)
    qs = rsphere.sample_sphere(3, 6, seed=77)
    a["cpdt_qs"] = qs
    for tag, src, t_true in (("3d", PointCloud(cube), np.array([0.5, -0.3, 0.2])),
                             ("2d", RotationProjection(cube), np.array([0.5, -0.3]))):
        tgt = src.transform_positions(R_true, t_true) + rng.normal(0, 0.05, (8, len(t_true)))
        tgt = np.vstack([tgt, rng.uniform(-3, 3, (4, len(t_true)))])
        a[f"cpdt_{tag}_target"], a[f"cpdt_{tag}_t"] = tgt, t_true
        for w in (0.0, 0.2, 0.4):
            cpd = CoherentPointDrift(PointCloud(tgt), src, sigma=0.5, k=8, beta=1.0, omega=w)
            key = f"cpdt_{tag}_w{int(10 * w)}"
            a[key + "_logp_true"] = np.float64(cpd.log_prob(R_true, t_true))
            a[key + "_logp_identity"] = np.float64(cpd.log_prob(np.eye(3), np.zeros(len(t_true))))
            a[key + "_logp_qs"] = np.array([cpd.log_prob(q, t_true) for q in qs])
            a[key + "_grad_qs"] = np.array([cpd.gradient(q, t_true) for q in qs])
            a[key + "_grad_qs_not"] = np.array([cpd.gradient(q) for q in qs])
    # result files as the reference's scripts write them (geosss/io.py: pickle protocol 2, plain and gzip)
    from geosss.io import dump
    runs = {m: rng.standard_normal((3, 5, 4)) for m in ("sss-reject", "sss-shrink", "rwmh", "hmc")}
    here = os.path.dirname(os.path.abspath(__file__))
    dump(runs, os.path.join(here, "ref_dump.pkl"))
    dump({"ess": [runs["rwmh"][0], 3.5, "text"], "n": 7}, os.path.join(here, "ref_dump.pkl.gz"), gzip=True)
    for m, v in runs.items():
        a["dump_" + m] = v
    save("helpers_kat.npz", **a)


if __name__ == "__main__":
    what = sys.argv[1:] or ["traj", "logprob", "geometry", "stats", "timing", "diagnostics", "mh", "mhk", "cpd", "helpers"]
    if "helpers" in what:
        make_helpers()
    if "mhk" in what:
        make_mh_kernels()
    if "cpd" in what:
        make_cpd()
    if "mh" in what:
        make_mh()
    if "timing" in what:
        make_timing()
    if "diagnostics" in what:
        make_diagnostics()
    if "traj" in what:
        make_trajectories()
    only = [w[5:] for w in what if w.startswith("only:")]
    if only:
        make_trajectories(only)
    if "logprob" in what:
        make_logprob_kat()
    if "geometry" in what:
        make_geometry_kat()
    if "stats" in what:
        make_stats()
    if "tangent" in what:
        make_tangent_kat()
    if "stats_cfg4" in what:  # round 4: cfg4's other two dimensions, 8 chains x 5000 steps after 3000 of burn-in
        make_stats({"curve_d50_kappa800": (5000, 3000), "curve_d200_kappa800": (5000, 3000)})

"""GPU parity of the paper's baselines -- MetropolisHastings (geosss/mcmc.py:118-176), SphericalHMC (:236-332),
IndependenceSampler (:179-182) and MixtureRWMHIndependenceSampler (:185-234) -- through the C ABI (GSSS_RWMH / GSSS_HMC /
GSSS_INDEP / GSSS_MIX) against the reference's recorded chains and the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import product_target, variants_for

pytestmark = pytest.mark.gpu

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("mh_") and f.endswith(".npz"))


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def build(gs, z, x0, seed, **kw):
    pdf = product_target(z)
    if str(z["sampler"]) == "rwmh":
        return gs.MetropolisHastings(pdf, x0, seed, stepsize=float(z["stepsize0"]), **kw)
    if str(z["sampler"]) == "indep":
        return gs.IndependenceSampler(pdf, x0, seed, stepsize=float(z["stepsize0"]), **kw)
    if str(z["sampler"]) == "mix":
        return gs.MixtureRWMHIndependenceSampler(pdf, x0, seed, stepsize=float(z["stepsize0"]),
                                                 mixing_probability=float(z["alpha"]), **kw)
    return gs.SphericalHMC(pdf, x0, seed, stepsize=float(z["stepsize0"]), n_steps=int(z["n_leapfrog"]), **kw)


def tol_for(z):  # see tests/test_oracle_mh.py: HMC trajectories amplify rounding
    return 1e-9 if str(z["sampler"]) == "hmc" else 1e-10


def horizon(z):  # see tests/test_oracle_mh.py: HMC on the k-nearest-neighbour force field of a registration target
    n = len(z["states"]) - 1
    return min(n, 15) if str(z["sampler"]) == "hmc" and str(z["target_kind"]) == "cpd" else n


def _params():
    out = []
    for name in CASES:
        d = int(golden(name + ".npz")["x0"].shape[0])
        out += [(name, v) for v in variants_for(d, max_coop=1)]
    return out


@pytest.mark.parametrize("name,variant", _params())
def test_replay_reproduces_reference_chain(gs, name, variant):
    """The reference's recorded draws through the HIP kernel: every state, every accept decision, the adapted stepsize."""
    z = golden(name + ".npz")
    s = build(gs, z, z["x0"], 1, variant=variant)
    n = len(z["states"]) - 1
    s.reset(int(z["burnin"]))
    kept = s.advance(n, thin=1, replay=z["draws"][None])
    got = kept[:, :, 0].cpu().numpy()
    assert s.errors[0] == 0
    h = horizon(z)
    acc = np.any(got != np.vstack([z["x0"][None], got[:-1]]), axis=1)          # the state moved = accepted
    assert np.array_equal(acc[:h], z["accept"].astype(bool)[:h])
    assert np.max(np.abs(got[:h] - z["states"][1:h + 1])) < tol_for(z)
    if h == n:
        assert s.n_accept == int(z["n_accept"])
        assert abs(s.stepsize / z["stepsize_trace"][-1] - 1) < 1e-12
    if str(z["sampler"]) == "mix":
        assert s.rwmh_counter == int(z["rwmh_counter"]) and s.indep_counter == int(z["indep_counter"])


@pytest.mark.parametrize("name", CASES)
def test_reference_chain_from_seed(gs, name):
    """Sampler(pdf, x0, seed).sample(n, burnin) from the seed alone on numpy's own stream (gamma, normal, uniform)."""
    z = golden(name + ".npz")
    s = build(gs, z, z["x0"], int(z["seed"]), rng="numpy")
    n = len(z["states"]) - 1
    s.reset(int(z["burnin"]))
    got = s.advance(n, thin=1)[:, :, 0].cpu().numpy()
    h = horizon(z)
    assert np.max(np.abs(got[:h] - z["states"][1:h + 1])) < tol_for(z)
    if h == n:
        assert s.n_accept == int(z["n_accept"])
        if "momenta" in z.files:
            assert np.max(np.abs(s.momenta - z["momenta"])) < 1e-7


@pytest.mark.parametrize("name", ["mh_rwmh_vmfmix_readme", "mh_hmc_vmfmix_readme", "mh_rwmh_bingham_d10_vmax30",
                                  "mh_hmc_bingham_d10_vmax30", "mh_rwmh_curve_d10_kappa800", "mh_hmc_curve_d10_kappa800",
                                  "mh_rwmh_curve_d50_kappa800", "mh_hmc_curve_d50_kappa800", "mh_hmc_bingham_d5_dense",
                                  "mh_indep_vmfmix_readme", "mh_mix_vmfmix_readme", "mh_mix_bingham_d10_vmax30",
                                  "mh_indep_bingham_d5_dense", "mh_mix_curve_d10_kappa800"])
def test_philox_stream_matches_oracle(gs, oracle, name):
    """Many chains on the library's counter-based stream: device = oracle (accept counts exactly, adapted stepsizes and
    states to rounding), for any split of the steps over launches."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    d = len(z["x0"])
    hmc = str(z["sampler"]) == "hmc"
    kind = {"rwmh": oracle.RWMH, "hmc": oracle.HMC, "indep": oracle.INDEP, "mix": oracle.MIX}[str(z["sampler"])]
    n_chains, n_steps, burn = (300, 30, 12) if d <= 10 else (64, 16, 6)
    x0 = oracle.sample_sphere(4, n_chains, d, chain_offset=50)
    want = oracle.mh_run(tgt, x0, n_steps, sampler=kind, stepsize=float(z["stepsize0"]),
                         adapt_steps=burn, n_leapfrog=10, seed=31, chain_offset=50, step_offset=3, n_threads=8,
                         mixing_probability=float(z["alpha"]) if kind == oracle.MIX else 0.5)
    s = build(gs, z, x0, 31, chain_offset=50, step_offset=3)
    s.reset(burn)
    s.advance(7)
    s.advance(n_steps - 7)
    # chains whose trajectories passed within rounding of an accept threshold may legitimately differ: none here
    assert np.array_equal(s.n_accept_per_chain, want["n_accept"])
    if kind == oracle.MIX:
        assert np.array_equal(s.rwmh_counter_per_chain, want["n_rwmh"]) and 0 < s.rwmh_counter < n_chains * n_steps
    assert np.max(np.abs(s.stepsize / want["stepsize"] - 1)) < 1e-12
    assert np.max(np.abs(s.state[:, :d] - want["state"])) < (1e-8 if hmc else 1e-10)


def test_sample_api_and_launcher(gs):
    """Reference call shapes: sample(n, burnin) with stepsize adaptation during burn-in (mcmc.py:169-176), the [x, v]
    state of SphericalHMC (mcmc.py:262), SamplerLauncher.run('rwmh' / 'hmc') (utils.py:210-232), call counters."""
    z = golden("traj_bingham_d10_vmax30.npz")
    pdf = product_target(z)
    type(pdf).log_prob.reset_counters()
    s = gs.MetropolisHastings(pdf, z["x0"], 5)
    out = s.sample(200, burnin=0.2)
    assert out.shape == (200, 10) and 0 < s.n_accept < 239 and isinstance(s.stepsize, float) and s.stepsize != 0.1
    assert pdf.log_prob.num_calls == 2 * 239
    e = s.stepsize
    s.advance(10)                                       # after burn-in the stepsize stays put (detailed balance)
    assert s.stepsize == e
    h = gs.SphericalHMC(pdf, z["x0"], 5, stepsize=0.05, n_steps=7)
    pos = h.sample(50, burnin=10)
    assert pos.shape == (50, 10) and h.state.shape == (20,) and h.momenta.shape == (10,)
    assert abs(np.dot(h.state[:10], h.state[10:])) < 1e-10     # momenta are tangent
    many = gs.MetropolisHastings(pdf, gs.sample_sphere(9, 1000, seed=2), 6, stepsize=0.2)
    X = many.sample(40, burnin=20)
    assert X.shape == (1000, 40, 10) and many.stepsize.shape == (1000,) and np.ptp(many.stepsize) > 0
    assert np.max(np.abs(np.linalg.norm(X, axis=-1) - 1)) < 1e-12
    L = gs.SamplerLauncher(pdf, z["x0"], 30, burnin=0.2, seed=3)
    assert L.run("rwmh").shape == (30, 10) and L.run("hmc").shape == (30, 10) and L.hmc.n_steps == 10
    with pytest.raises(ValueError):
        L.run("kent")
    with pytest.raises(ValueError):
        gs.MetropolisHastings(pdf, z["x0"], 1, mode="fast")


def test_independence_and_mixture_kernels_api(gs):
    """IndependenceSampler / MixtureRWMHIndependenceSampler as the reference's classes are called (mcmc.py:179-234,
    scripts/mixture_vMF_rwmh_indep.py): ctor keywords, sample(n, burnin), counters; on the two-mode README-like mixture the
    mixture kernel visits both hemispheres while plain RWMH with a small step stays where it started."""
    mus = 80.0 * np.array([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]])
    pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus])
    x0 = np.tile([0.0, 0.0, 1.0], (4000, 1))
    mix = gs.MixtureRWMHIndependenceSampler(pdf, x0, 11, stepsize=0.1, mixing_probability=0.8)
    X = mix.sample(300, burnin=100)
    assert X.shape == (4000, 300, 3) and mix.alpha == 0.8
    assert mix.rwmh_counter + mix.indep_counter == 399 * 4000
    assert abs(mix.rwmh_counter / (399 * 4000) - 0.8) < 0.01
    frac_south = float(np.mean(mix.state[:, 2] < 0))
    assert 0.35 < frac_south < 0.65, frac_south
    rw = gs.MetropolisHastings(pdf, x0, 11, stepsize=0.1)
    rw.sample(300, burnin=100)
    assert float(np.mean(rw.state[:, 2] < 0)) < 0.01
    ind = gs.IndependenceSampler(pdf, x0[0], 3)
    Y = ind.sample(500, burnin=0.2)
    assert Y.shape == (500, 3) and 0 < ind.n_accept < 599 and np.max(np.abs(np.linalg.norm(Y, axis=1) - 1)) < 1e-12
    sd = mix.state_dict()
    again = gs.MixtureRWMHIndependenceSampler(pdf, x0, 11, stepsize=0.1, mixing_probability=0.8)
    again.load_state_dict(sd)
    mix.advance(20)
    again.advance(20)
    assert np.array_equal(mix.state, again.state) and again.rwmh_counter == mix.rwmh_counter
    with pytest.raises(ValueError):
        gs.MixtureRWMHIndependenceSampler(pdf, x0, 1, mixing_probability=1.5)


def test_baselines_sample_the_target(gs):
    """RWMH and HMC ensembles agree with the slice sampler on the Bingham target's second moments (the eigenbasis target
    of scripts/bingham.py:131): three samplers, one distribution."""
    z = golden("traj_bingham_d10_vmax30.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(9, 20000, seed=8)
    ref = gs.ShrinkageSphericalSliceSampler(pdf, x0, 1)
    ref.advance(300)
    m_ref = np.mean(ref.state ** 2, axis=0)
    for cls, kw, steps in ((gs.MetropolisHastings, dict(stepsize=0.3), 3000), (gs.SphericalHMC, dict(stepsize=0.1), 400)):
        s = cls(pdf, x0, 2, **kw)
        s.reset(steps // 4)
        s.advance(steps)
        m = np.mean(s.state[:, :10] ** 2, axis=0)
        assert np.max(np.abs(m - m_ref)) < 0.02, (cls.__name__, m, m_ref)


# ------------------------------------------------------------------ round 3


def _per_step_rows(draws, offsets):
    n = len(offsets) - 1
    out = np.full((n, int(np.max(np.diff(offsets)))), 0.5)
    for i in range(n):
        out[i, : offsets[i + 1] - offsets[i]] = draws[offsets[i]: offsets[i + 1]]
    return out


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("mh_hmc_")])
def test_hmc_single_transition_teacher_forced(gs, name):
    """Every recorded transition of the reference's SphericalHMC at once: chain i starts from reference state i with the
    stepsize the reference used in step i and replays that step's draws; the state after ONE transition (ten leapfrog steps,
    the accept decision) equals the reference's at the north-star tolerance 1e-10 -- the free-running chains above are held to
    1e-9 because rounding is amplified from transition to transition, not within one."""
    z = golden(name + ".npz")
    states, n = z["states"], len(z["states"]) - 1
    eps = np.concatenate([[float(z["stepsize0"])], z["stepsize_trace"][:-1]])       # the stepsize step i starts with
    s = build(gs, z, states[:-1], 1)
    s.reset(0)                                                                       # (adaptation comes after the move)
    s.stepsize = eps
    s.advance(1, replay=_per_step_rows(z["draws"], z["step_draw_offset"]))
    assert np.all(s.errors == 0)
    got = s.state[:, : states.shape[1]]
    moved = np.any(got != states[:-1], axis=1)
    h = horizon(z) if str(z["target_kind"]) != "cpd" else n       # one transition each: the whole recorded chain, registration too
    assert np.array_equal(moved[:h], z["accept"].astype(bool)[:h])
    assert np.max(np.abs(got[:h] - states[1:h + 1])) < 1e-10
    if "momenta_trace" in z.files:                                 # the momentum half of the reference's rows (mcmc.py:262)
        assert np.max(np.abs(s.momenta[:h] - z["momenta_trace"][1:h + 1])) < 1e-8


@pytest.mark.parametrize("name", ["mh_hmc_vmfmix_readme", "mh_hmc_bingham_d10_vmax30", "mh_hmc_curve_d10_kappa800"])
def test_hmc_sample_returns_momenta(gs, name):
    """SphericalHMC.sample(n, burnin, return_momenta=True) -> (positions, momenta), the two halves of the reference's [x, v]
    rows (mcmc.py:321-332), from the seed alone on numpy's stream; row 0 is the initial state with zero momenta."""
    z = golden(name + ".npz")
    n, burn, d = len(z["states"]) - 1, int(z["burnin"]), len(z["x0"])
    s = build(gs, z, z["x0"], int(z["seed"]), rng="numpy")
    pos, mom = s.sample(n + 1 - burn, burn, return_momenta=True, return_all_samples=True)
    assert pos.shape == mom.shape == (n + 1, d)
    assert np.array_equal(pos[0], z["x0"]) and np.all(mom[0] == 0.0)
    assert np.max(np.abs(pos - z["states"])) < 1e-9 and np.max(np.abs(mom - z["momenta_trace"])) < 1e-7
    # without the burn-in rows, many chains, device tensors
    t = build(gs, z, np.tile(z["x0"], (5, 1)), 3)
    p2, m2 = t.sample(12, burnin=6, return_momenta=True, as_tensor=True)
    assert p2.shape == m2.shape == (5, 12, d) and p2.is_cuda and m2.is_cuda
    assert np.allclose(m2[:, -1].cpu().numpy(), t.momenta) and np.array_equal(p2[:, -1].cpu().numpy(), t.state[:, :d])


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("mh_mix_")])
def test_mixture_kernel_records_rwmh_stepsizes(gs, name):
    """MixtureRWMHIndependenceSampler.rwmh_stepsize_vals (mcmc.py:201, 228): the stepsize after every RWMH proposal, from the
    seed alone on numpy's stream -- the reference's list, entry for entry."""
    z = golden(name + ".npz")
    s = build(gs, z, z["x0"], int(z["seed"]), rng="numpy")
    s.reset(int(z["burnin"]))
    s.advance(len(z["states"]) - 1)
    vals = s.rwmh_stepsize_vals
    assert isinstance(vals, list) and len(vals) == int(z["rwmh_counter"]) == s.rwmh_counter
    assert np.max(np.abs(np.array(vals) / z["rwmh_stepsize_vals"] - 1)) < 1e-12
    many = gs.MixtureRWMHIndependenceSampler(product_target(z), np.tile(z["x0"], (3, 1)), 5, mixing_probability=float(z["alpha"]))
    with pytest.raises(ValueError):
        many.rwmh_stepsize_vals                                   # an ensemble records them on request only
    many = gs.MixtureRWMHIndependenceSampler(product_target(z), np.tile(z["x0"], (3, 1)), 5, mixing_probability=float(z["alpha"]),
                                             record_stepsize=True)
    many.reset(10)
    many.advance(7)
    many.advance(30)
    per = many.rwmh_stepsize_vals
    assert [len(v) for v in per] == list(many.rwmh_counter_per_chain) and all(np.all(v > 0) for v in per)

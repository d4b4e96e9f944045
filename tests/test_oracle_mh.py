"""The oracle's restatement of MetropolisHastings (geosss/mcmc.py:118-176), SphericalHMC (:236-332), IndependenceSampler
(:179-182) and MixtureRWMHIndependenceSampler (:185-234) against chains the reference itself produced
(tests/golden/mh_*.npz, written by make_golden.py mh / mhk): replaying the recorded draws and -- numpy's gamma / normal /
uniform stream restated -- from the seed alone."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("mh_") and f.endswith(".npz"))
TOL = 1e-10


def tol_for(z):
    """RWMH: 1e-10 like the slice samplers.  HMC integrates ten leapfrog steps per transition through a kappa ~ 100
    force field: rounding differences between numpy's BLAS dots and a plain C loop grow along the trajectory (all
    accept decisions still coincide; worst case the README mixture at stepsize 0.1, beyond the integrator's stability
    limit: 5e-11 per transition), so free-running chains are held to 1e-9 and single transitions started from the
    reference's own states (test_hmc_single_transitions) to 1e-10."""
    return 1e-9 if str(z["sampler"]) == "hmc" else TOL


def horizon(z):
    """Steps over which a free-running chain is compared.  HMC on a registration target integrates a force field with
    jumps (the k-nearest-neighbour sets change along the trajectory): once rounding flips one neighbour, trajectories part
    for good (all accept decisions of the fixtures still coincide), so those chains are compared over their first 15
    steps and transition by transition (test_hmc_single_transitions)."""
    n = len(z["states"]) - 1
    return min(n, 15) if str(z["sampler"]) == "hmc" and str(z["target_kind"]) == "cpd" else n


def _run(oracle, z, **kw):
    tgt = oracle.Target.from_fixture(z)
    kind = {"rwmh": oracle.RWMH, "hmc": oracle.HMC, "indep": oracle.INDEP, "mix": oracle.MIX}[str(z["sampler"])]
    n = len(z["states"]) - 1
    if kind == oracle.MIX:
        kw["mixing_probability"] = float(z["alpha"])
    return oracle.mh_run(tgt, z["x0"], n, sampler=kind, stepsize=float(z["stepsize0"]), adapt_steps=int(z["burnin"]),
                         n_leapfrog=int(z["n_leapfrog"]), trace=True, **kw)


def _with_gradient_answers():
    """The fixtures that hold gradient known answers (the RWMH / HMC fixtures of a target; registration targets are log_prob only) --
    picked here rather than skipped inside the test: a skip is a line in the driver's record."""
    out = []
    for name in CASES:
        z = golden(name + ".npz")
        if "grad_X" in z.files and len(z["grad_X"]) > 0:
            out.append(name)
    return out


@pytest.mark.parametrize("name", _with_gradient_answers())
def test_gradient_kat(oracle, name):
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    got = np.array([oracle.gradient(tgt, x) for x in z["grad_X"]])
    assert np.max(np.abs(got - z["grad"]) / np.maximum(1.0, np.abs(z["grad"]))) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_replay_reproduces_reference_chain(oracle, name):
    z = golden(name + ".npz")
    out = _run(oracle, z, replay=z["draws"][None])
    assert out["err"][0] == 0
    assert np.array_equal(out["accept"][0], z["accept"])
    assert int(out["n_accept"][0]) == int(z["n_accept"])
    h = horizon(z)
    assert np.max(np.abs(out["samples"][0][:h] - z["states"][1:h + 1])) < tol_for(z)
    assert np.max(np.abs(out["stepsize_trace"][0] / z["stepsize_trace"] - 1)) < 1e-13
    if "momenta" in z.files and h == len(z["states"]) - 1:
        assert np.max(np.abs(out["momenta"] - z["momenta"])) < 1e-8
    if str(z["sampler"]) == "mix":     # which kernel proposed, the counters, the stepsize after every RWMH proposal
        assert np.array_equal(out["use_rwmh"][0], z["use_rwmh"])
        assert int(out["n_rwmh"][0]) == int(z["rwmh_counter"]) and len(z["states"]) - 1 - int(out["n_rwmh"][0]) == int(z["indep_counter"])
        assert np.max(np.abs(out["stepsize_trace"][0][z["use_rwmh"].astype(bool)] / z["rwmh_stepsize_vals"] - 1)) < 1e-13
        assert int(out["adapt_left"][0]) == max(0, int(z["burnin"]) - int(z["rwmh_counter"]))


@pytest.mark.parametrize("name", CASES)
def test_from_seed_reproduces_reference_chain(oracle, name):
    """default_rng(seed): gamma(d/2) (Marsaglia-Tsang on the ziggurat normals), standard_normal, random restated."""
    z = golden(name + ".npz")
    out = _run(oracle, z, numpy_seed=int(z["seed"]))
    h = horizon(z)
    assert np.array_equal(out["accept"][0][:h], z["accept"][:h])
    assert np.max(np.abs(out["samples"][0][:h] - z["states"][1:h + 1])) < tol_for(z)


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("mh_hmc")])
def test_hmc_single_transitions(oracle, name):
    """Every transition of the reference HMC chain on its own: chain i starts at reference state i with the stepsize the
    reference had at that step and replays the draws of step i."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    n = len(z["states"]) - 1
    offs = z["step_draw_offset"]
    width = int(np.max(np.diff(offs)))
    rep = np.stack([z["draws"][offs[i]:offs[i] + width] if offs[i] + width <= len(z["draws"]) else
                    np.pad(z["draws"][offs[i]:], (0, offs[i] + width - len(z["draws"]))) for i in range(n)])
    eps = np.concatenate([[float(z["stepsize0"])], z["stepsize_trace"][:-1]])
    out = oracle.mh_run(tgt, z["states"][:-1], 1, sampler=oracle.HMC, stepsize=eps, n_leapfrog=int(z["n_leapfrog"]),
                        replay=rep, trace=True)
    assert np.array_equal(out["accept"][:, 0], z["accept"])
    assert np.max(np.abs(out["state"] - z["states"][1:])) < 1e-10


def test_numpy_gamma_stream(oracle):
    for seed, shape in ((1, 1.5), (2, 5.0), (3, 25.0), (4, 100.0)):
        want = np.random.default_rng(seed).gamma(shape, size=20000)
        got, _ = oracle.npy_gamma(oracle.pcg64_words(seed)[0], shape, 20000)
        assert np.array_equal(got, want)

"""GPU parity of the registration targets (geosss/registration.py: CoherentPointDrift, GaussianMixtureModel on unit
quaternions) through the C ABI: log_prob against the reference's values, slice-sampler and RWMH chains against the
oracle on the Philox stream.  (Replay and from-seed parity with the reference's recorded chains: the parametrised
tests of test_hip_parity.py / test_hip_mh.py pick the traj_cpd_* / mh_rwmh_cpd_* fixtures up by name.)"""
import numpy as np
import pytest

from conftest import golden
from helpers import product_target

pytestmark = pytest.mark.gpu
CASES = ["cpd_protein", "gmm_protein_k10", "cpd_cube_3d2d"]


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


@pytest.mark.parametrize("name", CASES)
def test_log_prob_kat(gs, name):
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    got = pdf.log_prob(z["kat_q"])
    assert np.max(np.abs(got - z["kat_logp"]) / np.maximum(1.0, np.abs(z["kat_logp"]))) < 1e-10
    one = pdf.log_prob(z["kat_q"][5])
    assert isinstance(one, float) and abs(one - z["kat_logp"][5]) < 1e-10 * max(1.0, abs(z["kat_logp"][5]))
    R = gs.registration.quat2matrix(z["kat_q"][7])                     # the reference also takes rotation matrices
    assert abs(pdf.log_prob(R) - z["kat_logp"][7]) < 1e-9 * max(1.0, abs(z["kat_logp"][7]))
    g = pdf.gradient(z["kat_q"][0])                                    # (the values: test_helpers.test_device_gradient_known_answers)
    assert g.shape == (4,) and np.all(np.isfinite(g))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("sampler", ["shrink", "reject", "rwmh"])
def test_philox_stream_matches_oracle(gs, oracle, name, sampler):
    z = golden(f"traj_{name}.npz")
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    n, steps = (96, 6) if "protein" in name else (300, 25)
    if sampler == "reject":
        steps = max(2, steps // 3)
    x0 = oracle.sample_sphere(9, n, 4)
    if sampler == "rwmh":
        want = oracle.mh_run(tgt, x0, steps, sampler=oracle.RWMH, stepsize=0.1, adapt_steps=steps // 2, seed=4, n_threads=8)
        s = gs.MetropolisHastings(pdf, x0, 4, stepsize=0.1)
        s.reset(steps // 2)
        s.advance(steps)
        assert np.array_equal(s.n_accept_per_chain, want["n_accept"])
        assert np.max(np.abs(s.state - want["state"])) < 1e-10
        return
    kind = oracle.REJECT if sampler == "reject" else oracle.SHRINK
    want = oracle.run(tgt, x0, steps, seed=4, sampler=kind, n_threads=8)
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    s = cls(pdf, x0, 4)
    assert s.mode == "exact"
    s.advance(steps)
    assert np.all(s.errors == 0)
    assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
    assert np.max(np.abs(s.state - want["state"])) < 1e-10


def test_hmc_runs_on_a_registration_target(gs, oracle):
    """SphericalHMC with Registration.gradient on the device: single transitions from 200 poses against the oracle (free
    running chains part once rounding flips a nearest neighbour: tests/test_oracle_mh.py::horizon)."""
    z = golden("traj_cpd_cube_3d2d.npz")
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    x0 = oracle.sample_sphere(2, 200, 4)
    want = oracle.mh_run(tgt, x0, 1, sampler=oracle.HMC, stepsize=0.05, n_leapfrog=10, seed=8, n_threads=8)
    h = gs.SphericalHMC(pdf, x0, 8, stepsize=0.05, n_steps=10)
    h.advance(1)
    assert np.array_equal(h.n_accept_per_chain, want["n_accept"])
    assert np.max(np.abs(h.state[:, :4] - want["state"])) < 1e-10
    assert np.max(np.abs(h.momenta - want["momenta"])) < 1e-9


def test_registration_finds_the_pose(gs):
    """The protein example of scripts/protein_reg3d3d.py: chains started at random poses concentrate where the reference
    pose scores (data/protein_registration.npz holds the true rotation; its arrays travel in the fixture)."""
    z = golden("traj_cpd_protein.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(3, 256, seed=3)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 7)
    before = pdf.log_prob(s.state)
    s.advance(60)
    after = pdf.log_prob(s.state)
    assert np.median(after) > np.median(before) + 50.0
    assert np.all(np.abs(np.linalg.norm(s.state, axis=1) - 1) < 1e-12)

"""The direct generators of geosss/rand.py on the device (geosss_amd.rand): Wood's von Mises-Fisher scheme and the
Kent-Ganeiber-Mardia envelope for the Bingham distribution, checked against closed forms and against the slice sampler."""
import numpy as np
import pytest
from scipy.special import ive

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


@pytest.mark.parametrize("d,kappa", [(3, 80.0), (3, 2.0), (10, 100.0), (5, 0.0)])
def test_sample_vmf_moments(gs, d, kappa):
    """E[x] = A_d(kappa) mu / |mu| with A_d = I_{d/2}(kappa) / I_{d/2-1}(kappa); the draws are unit vectors; the call
    shapes of geosss.rand.sample_vMF (size = 1 -> one point)."""
    rng = np.random.default_rng(d)
    m = rng.standard_normal(d)
    m /= np.linalg.norm(m)
    pdf = gs.VonMisesFisher(kappa * m if kappa > 0 else np.zeros(d))
    n = 400_000
    X = gs.sample_vMF(pdf, n, seed=11)
    assert X.shape == (n, d) and np.max(np.abs(np.linalg.norm(X, axis=1) - 1)) < 1e-12
    want = (ive(d / 2, kappa) / ive(d / 2 - 1, kappa)) * m if kappa > 0 else np.zeros(d)
    assert np.max(np.abs(X.mean(0) - want)) < 5.0 / np.sqrt(n)
    assert gs.sample_vMF(pdf, seed=1).shape == (d,)
    assert np.array_equal(gs.sample_vMF(pdf, 100, seed=5), gs.sample_vMF(pdf, 100, seed=5))
    with pytest.raises(AssertionError):
        gs.sample_vMF(gs.Bingham(np.eye(d)), 3)


def test_sample_bingham_agrees_with_the_slice_sampler(gs):
    """Second moments of direct Bingham draws (rand.sample_bingham) against an ensemble of the shrinkage slice sampler on
    the paper's d = 10 target, and a dense 5-dimensional one; the efficiency the reference reports."""
    pdf = gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982)
    X, eff = gs.sample_bingham(pdf.A, 300_000, return_efficiency=True, seed=3)
    assert X.shape == (300_000, 10) and 0.0 < eff <= 1.0 and np.max(np.abs(np.linalg.norm(X, axis=1) - 1)) < 1e-12
    s = gs.ShrinkageSphericalSliceSampler(pdf, gs.sample_sphere(9, 50_000, seed=8), 1)
    s.advance(400)
    assert np.max(np.abs((X ** 2).mean(0) - (s.state ** 2).mean(0))) < 0.01
    dense = gs.random_bingham(d=5, vmax=12.0, vmin=-1.0, eigensystem=False, seed=4)
    Y = gs.sample_bingham(dense, 300_000, seed=9)
    t = gs.ShrinkageSphericalSliceSampler(dense, gs.sample_sphere(4, 50_000, seed=2), 3)
    t.advance(400)
    assert np.max(np.abs(Y.T @ Y / len(Y) - t.state.T @ t.state / len(t.state))) < 0.01
    with pytest.raises(ValueError):
        gs.sample_bingham_3d(pdf, 5)
    assert gs.sample_bingham_3d(gs.Bingham(np.diag([3.0, 1.0, 0.0])), 7, seed=1).shape == (7, 3)


def test_slice_sampler_on_s2_matches_direct_mixture_draws(gs):
    """The README mixture on S^2, two independent routes to the same distribution: 10^6 shrinkage chains on the library
    stream (which draws the tangent direction directly on S^2) after 600 steps, and 10^6 exact draws
    (a component by its weight, then Wood's vMF generator).  First and second moments agree to sampling error."""
    mus = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
    pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus])
    n = 1_000_000
    s = gs.ShrinkageSphericalSliceSampler(pdf, gs.sample_sphere(2, n, seed=3), 17)
    s.advance(600)
    X = s.state
    Y = np.concatenate([gs.sample_vMF(gs.VonMisesFisher(m), n // 3 + (i < n % 3), seed=40 + i) for i, m in enumerate(mus)])
    assert np.max(np.abs(X.mean(0) - Y.mean(0))) < 5e-3          # chains still carry their start's memory at the 1e-3 level
    assert np.max(np.abs(X.T @ X / n - Y.T @ Y / len(Y))) < 5e-3
    r = gs.RejectionSphericalSliceSampler(pdf, gs.sample_sphere(2, 200_000, seed=5), 19)
    r.advance(300)
    assert np.max(np.abs(r.state.mean(0) - Y.mean(0))) < 1e-2

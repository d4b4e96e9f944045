"""geosss_amd.diagnostics against the reference's own estimators (tests/golden/diagnostics_kat.npz was
written by running geosss.utils / geosss.sphere on reference chains)."""
import numpy as np
import pytest
import torch

from conftest import golden


def _check(dg, dev):
    z = golden("diagnostics_kat.npz")
    X = z["vmf_X"]
    Xt = torch.as_tensor(X).to(dev)
    series = Xt.T.contiguous()                       # (3, n): three scalar series at once
    assert np.allclose(dg.acf_fft(series).cpu().numpy(), z["vmf_acf_fft"], rtol=0, atol=1e-12)
    assert np.allclose(dg.acf(series, 50).cpu().numpy(), z["vmf_acf"], rtol=0, atol=1e-12)
    assert np.allclose(dg.IAT(series).cpu().numpy(), z["vmf_IAT"], rtol=1e-10)
    assert np.allclose(dg.IAT(series, 200).cpu().numpy(), z["vmf_IAT_200"], rtol=1e-10)
    assert np.allclose(dg.n_eff(series).cpu().numpy(), z["vmf_n_eff"], rtol=1e-10)
    assert np.allclose(dg.distance(Xt[1:], Xt[:-1]).cpu().numpy(), z["vmf_distance"], rtol=0, atol=1e-12)
    occ = dg.mode_occupancy(Xt, torch.as_tensor(z["vmf_modes"]).to(dev))
    assert np.allclose(occ.cpu().numpy(), z["vmf_occupancy"], atol=1e-15)
    assert abs(float(dg.mode_kl(occ, torch.full((3,), 1 / 3, dtype=torch.float64, device=dev))) - float(z["vmf_kl"])) < 1e-12
    Xb = torch.as_tensor(z["bingham_X"]).to(dev)
    assert abs(float(dg.hopping_frequency(Xb, z["bingham_mode"])) - float(z["bingham_hop"])) < 1e-15
    assert np.allclose(dg.IAT(Xb.T.contiguous()).cpu().numpy(), z["bingham_IAT"], rtol=1e-10)
    # numpy in -> numpy / scalar out, like the reference
    assert isinstance(dg.IAT(X[:, 0]), float) and abs(dg.IAT(X[:, 0]) - z["vmf_IAT"][0]) < 1e-9
    assert isinstance(dg.acf_fft(X[:, 1]), np.ndarray)


def test_diagnostics_cpu():
    from geosss_amd import diagnostics as dg
    _check(dg, "cpu")


@pytest.mark.gpu
def test_diagnostics_gpu():
    from geosss_amd import diagnostics as dg
    _check(dg, "cuda")


def running_rows(d, n_modes, n_lags):
    return 1 + 2 * d + d * (d + 1) // 2 + 2 + n_modes + 2 + 3 * n_lags


def accumulate_numpy(X, w, h, modes, L):
    """What the sampler kernels accumulate per retained draw (gsss_device.h: stats_update), restated in numpy for one
    chain: the fixture for the host-side derivations."""
    n, d = X.shape
    K = len(modes)
    acc = np.zeros(running_rows(d, K, L))
    T = d * (d + 1) // 2
    r_prev, r_sum, r_xx = 1, 1 + d, 1 + 2 * d
    r_dist = r_xx + T
    r_hop, r_mode = r_dist + 1, r_dist + 2
    r_p = r_mode + K
    r_lag, r_ring, r_head = r_p + 2, r_p + 2 + L, r_p + 2 + 2 * L
    iu = np.triu_indices(d)
    for t, x in enumerate(X):
        p = x @ w
        if t > 0:
            prev = acc[r_prev:r_prev + d]
            acc[r_dist] += np.arccos(np.clip(prev @ x, -1, 1))
            acc[r_hop] += float(np.sign(x @ h) != np.sign(prev @ h))
        acc[r_prev:r_prev + d] = x
        acc[r_sum:r_sum + d] += x
        acc[r_xx:r_xx + T] += np.outer(x, x)[iu]
        if K:
            acc[r_mode + int(np.argmax(modes @ x))] += 1
        acc[r_p] += p
        acc[r_p + 1] += p * p
        for l in range(1, min(L, t) + 1):
            acc[r_lag + l - 1] += p * acc[r_ring + (t - l) % L]
        acc[r_ring + t % L] = p
        if t < L:
            acc[r_head + t] = p
        acc[0] = t + 1
    return acc


def test_running_statistics_host_math_against_reference_kat():
    """diagnostics.from_running turns the kernels' accumulators into the reference's estimators: utils.acf exactly
    (utils.py:96-110), sphere.distance of consecutive draws, mode occupancy, hopping frequency -- checked against the
    values geosss itself produced for the same series (diagnostics_kat.npz)."""
    from geosss_amd import diagnostics as dg
    z = golden("diagnostics_kat.npz")
    X, modes, L = z["vmf_X"], z["vmf_modes"], 49
    accs = np.stack([accumulate_numpy(X, np.eye(3)[j], modes[0], modes, L) for j in range(3)], axis=1)
    assert int(dg.__dict__["_t"](accs).shape[0]) == running_rows(3, 3, L)
    r = dg.from_running(torch.as_tensor(accs), 3, 3, L)
    assert np.allclose(r["acf"].numpy(), z["vmf_acf"], rtol=0, atol=1e-11)            # lags 0 .. 49 of each coordinate
    assert np.allclose(r["geodesic_step"].numpy(), z["vmf_distance"].mean(), rtol=1e-12)
    assert np.allclose(r["mode_occupancy"].numpy(), z["vmf_occupancy"][None], atol=1e-15)
    assert np.allclose(r["mean"][0].numpy(), X.mean(0), atol=1e-13)
    assert np.allclose(r["second_moment"][0].numpy(), X.T @ X / len(X), atol=1e-13)
    want_iat = dg.iat_from_acf(torch.as_tensor(z["vmf_acf"])).numpy()
    assert np.allclose(r["iat"].numpy(), want_iat, rtol=1e-9)
    Xb, mode = z["bingham_X"], z["bingham_mode"]
    rb = dg.from_running(torch.as_tensor(accumulate_numpy(Xb, np.eye(10)[0], mode, np.zeros((0, 10)), 8)[:, None]), 10, 0, 8)
    assert abs(float(rb["hopping_frequency"][0]) - float(z["bingham_hop"])) < 1e-15


def test_spiral_grid_and_grid_kl():
    """diagnostics.saff_sphere lays out the cells of the reference's KL estimate (scripts/visualize_curve_vMF.ipynb
    `saff_sphere`, `calc_kld`): unit vectors from pole to pole, heights equally spaced, azimuth advancing by
    3.6 / sqrt(n (1 - h^2)); grid_kl of draws laid exactly on the grid in proportion to p is ~0, of uniform draws the
    KL between p and the uniform histogram."""
    from geosss_amd import diagnostics as D
    g = D.saff_sphere(1500)
    assert g.shape == (1500, 3) and np.max(np.abs(np.linalg.norm(g, axis=1) - 1)) < 1e-15
    assert np.allclose(g[0], [0, 0, -1], atol=1e-7) and np.allclose(g[-1], [0, 0, 1], atol=1e-7)
    assert np.allclose(np.diff(g[:, 2]), 2 / 1499)
    az = np.unwrap(np.arctan2(g[1:-1, 1], g[1:-1, 0]))
    assert np.allclose(np.diff(az), 3.6 / np.sqrt(1500 * (1 - g[2:-1, 2] ** 2)), atol=1e-9)

    class Pdf:                                         # a smooth density on S^2
        def log_prob(self, x):
            return 3.0 * np.asarray(x)[:, 2]
    p = np.exp(3.0 * g[:, 2]); p /= p.sum()
    counts = np.rint(p * 500_000).astype(int)
    on_grid = np.repeat(g, counts, axis=0)
    assert abs(float(D.grid_kl(Pdf(), torch.from_numpy(on_grid), 1500))) < 1e-3
    rng = np.random.default_rng(3)
    u = rng.standard_normal((100_000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    cell = np.argmax(u @ g.T, axis=1)
    q = np.bincount(cell, minlength=1500) / len(u) + 1e-12
    want = float(np.sum(p * (np.log(p) - np.log(q))))
    got = D.grid_kl(Pdf(), np.stack([u, u]), 1500)
    assert got.shape == (2,) and abs(got[0] - want) < 1e-9


def test_ess_bulk_known_regimes():
    """diagnostics.ess_bulk (rank-normalised split-chain bulk ESS, the estimator behind the paper's "relative ESS",
    scripts/bingham.py:43-57) in regimes with a known answer: independent draws -> relative ESS ~ 1; an AR(1) series with
    coefficient phi -> (1 - phi) / (1 + phi); invariance under monotone transforms of the data (it only sees ranks); chains
    that disagree about the mean -> a small ESS whatever their own mixing; antithetic draws -> above 1 (capped by log10)."""
    from geosss_amd import diagnostics as D
    rng = np.random.default_rng(5)
    iid = rng.standard_normal((10, 20000))
    r = D.ess_bulk(iid, relative=True)
    assert 0.93 < r < 1.07, r
    assert abs(D.ess_bulk(iid) / iid.size - r) < 1e-12
    assert abs(D.ess_bulk(np.exp(3 * iid), relative=True) - r) < 1e-12          # ranks only
    for phi in (0.5, 0.9):
        e = rng.standard_normal((10, 50000))
        x = np.empty_like(e)
        x[:, 0] = e[:, 0] / np.sqrt(1 - phi * phi)
        for t in range(1, e.shape[1]):
            x[:, t] = phi * x[:, t - 1] + e[:, t]
        got, want = D.ess_bulk(x, relative=True), (1 - phi) / (1 + phi)
        assert abs(got / want - 1) < 0.08, (phi, got, want)
    shifted = iid + np.arange(10)[:, None]
    assert D.ess_bulk(shifted, relative=True) < 1e-3
    anti = np.empty((4, 10000))
    anti[:, 0::2] = rng.standard_normal((4, 5000))
    anti[:, 1::2] = -anti[:, 0::2]
    assert D.ess_bulk(anti, relative=True) > 1.5
    t = torch.as_tensor(iid)
    assert abs(D.ess_bulk(t, relative=True) - r) < 1e-12
    with pytest.raises(ValueError):
        D.ess_bulk(np.zeros(10))
    ties = np.round(iid, 1)                                                     # heavy ties: average ranks, like scipy's rankdata
    from scipy.stats import rankdata
    assert np.array_equal(D._average_ranks(torch.as_tensor(ties.reshape(-1))).numpy(), rankdata(ties.reshape(-1)))


def test_ess_between_chains_recovers_ar1_tau():
    """AR(1) chains x_t = phi x_{t-1} + e_t have tau = (1 + phi) / (1 - phi): 4000 stationary chains of 2000 draws give it back
    within the estimator's own standard error (3 sigma), from per-chain mean / count / variance alone -- and the reference's
    windowed IAT heuristic (utils.py:119-131) on the same chains agrees where the window holds the whole autocorrelation."""
    from geosss_amd import diagnostics as dg
    rng = np.random.default_rng(12)
    C, n = 4000, 2000
    for phi in (0.0, 0.6, 0.9, 0.97):
        x = np.empty((C, n))
        x[:, 0] = rng.standard_normal(C) / np.sqrt(1.0 - phi * phi)       # the stationary law: no burn-in needed
        e = rng.standard_normal((C, n))
        for t in range(1, n):
            x[:, t] = phi * x[:, t - 1] + e[:, t]
        r = dg.ess_between_chains(x.mean(1), np.full(C, n), x.var(1))
        tau = (1.0 + phi) / (1.0 - phi)
        tau_n = tau - 2.0 * phi * (1.0 - phi ** n) / (n * (1.0 - phi) ** 2)  # the finite-n value the estimator targets
        assert abs(r["tau"] / tau_n - 1.0) < 3.0 * r["rel_se"] + 1e-3, (phi, r, tau_n)
        assert abs(r["ess_per_chain"] * r["tau"] - n) < 1e-9 and r["chains"] == C
        if phi <= 0.9:
            iat = np.mean(dg.IAT(x[:400]))
            assert abs(iat / tau - 1.0) < 0.12, (phi, iat, tau)
    with pytest.raises(ValueError):
        dg.ess_between_chains(np.zeros(1), np.ones(1), np.ones(1))
    with pytest.raises(ValueError):
        dg.ess_between_chains(np.zeros(3), np.array([5.0, 5.0, 6.0]), np.ones(3))

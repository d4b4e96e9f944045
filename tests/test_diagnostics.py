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

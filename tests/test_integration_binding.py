"""INTEGRATION.md section B -- the ctypes module a geosss maintainer would add -- is EXECUTED here, as it stands in the
document, against a stand-in for `geosss.mcmc` (constructor attributes and determine_burnin, mcmc.py:13-19, 28-45, 340-355:
the reference's files cannot travel to the GPU box).  It is the only caller of the torch-free helpers of the C ABI
(gsss_malloc / gsss_free / gsss_memcpy_h2d / gsss_memcpy_d2h / gsss_memset, include/gsss.h) in the suite.
Interface replaced: Sampler.sample geosss/mcmc.py:55-77 for SamplerLauncher.run_sss_shrink geosss/utils.py:204-208."""
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def binding_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(# geosss/hip\.py.*?)```", text, flags=re.S)
    assert len(blocks) == 1, "INTEGRATION.md section B must hold exactly one geosss/hip.py block"
    return blocks[0]


def stand_in_geosss():
    """`geosss.mcmc` reduced to what section B imports: the constructor's attributes and determine_burnin."""
    pkg, mcmc = types.ModuleType("geosss"), types.ModuleType("geosss.mcmc")

    def determine_burnin(n_samples, burnin):
        if isinstance(burnin, float):
            assert 0 <= burnin <= 1.0
            return int(burnin * n_samples)
        assert burnin >= 0
        return burnin

    class RejectionSphericalSliceSampler:
        def __init__(self, distribution, initial_state, seed=None):
            self.target = distribution
            self.state = initial_state
            self.rng = np.random.default_rng(seed)
            self.n_reject = 0

    mcmc.determine_burnin = determine_burnin
    mcmc.RejectionSphericalSliceSampler = RejectionSphericalSliceSampler
    pkg.mcmc = mcmc
    return {"geosss": pkg, "geosss.mcmc": mcmc}


class CountingLib:
    """Passes every call through to the CDLL and counts allocations against releases."""

    def __init__(self, lib):
        self._lib = lib
        self.calls = {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            self.calls[name] = self.calls.get(name, 0) + 1
            return fn(*a)
        return call if name != "gsss_last_error" else fn


def load_binding(monkeypatch):
    from geosss_amd import _lib
    monkeypatch.setenv("GSSS_HIP_LIB", _lib.LIB_PATH)
    for name, mod in stand_in_geosss().items():
        monkeypatch.setitem(sys.modules, name, mod)
    ns = {"__name__": "geosss.hip"}
    exec(compile(binding_source(), "INTEGRATION.md#B", "exec"), ns)
    ns["_lib"] = CountingLib(ns["_lib"])
    return ns


def test_binding_block_loads_the_library(monkeypatch):
    """No GPU needed: the block compiles, loads libgsss_hip.so, finds every symbol it sets argument types for, and its two
    structures have the sizes of the library's own binding."""
    from geosss_amd import _lib
    src = binding_source()
    ns = load_binding(monkeypatch)
    import ctypes as C
    assert C.sizeof(ns["_Run"]) == C.sizeof(_lib.RunArgs) and C.sizeof(ns["_Desc"]) == C.sizeof(_lib.TargetDesc)
    assert issubclass(ns["HipShrinkageSphericalSliceSampler"], sys.modules["geosss.mcmc"].RejectionSphericalSliceSampler)
    for sym in ("gsss_malloc", "gsss_free", "gsss_memcpy_h2d", "gsss_memcpy_d2h", "gsss_memset", "gsss_rows_to_components",
                "gsss_run", "gsss_samples_to_chains", "gsss_target_create", "gsss_target_destroy"):
        assert sym in src, sym


def _targets():
    from types import SimpleNamespace as NS
    import bench
    mix = NS(pdfs=[NS(mu=m) for m in bench.README_MUS], weights=np.full(3, 1.0 / 3.0))
    A = np.diag([20.0, 11.0, 5.0, 1.0, 0.0])
    A[0, 1] = A[1, 0] = 2.5
    bing = NS(A=A)
    rng = np.random.default_rng(5)
    knots = rng.standard_normal((10, 10))
    knots /= np.linalg.norm(knots, axis=1)[:, None]
    for i in range(1, 10):  # a random walk on the sphere rather than ten unrelated points
        knots[i] = knots[i - 1] + 0.5 * knots[i]
        knots[i] /= np.linalg.norm(knots[i])
    curve = NS(curve=NS(knots=knots), kappa=300.0)
    return {"mixture": mix, "bingham": bing, "curve": curve}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mixture", "bingham", "curve"])
@pytest.mark.parametrize("n_chains", [1, 300])
def test_reference_side_binding_runs_and_matches_oracle(oracle, monkeypatch, kind, n_chains):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    ns = load_binding(monkeypatch)
    pdf = _targets()[kind]
    if kind == "mixture":
        tgt = oracle.Target.vmf_mixture([p.mu for p in pdf.pdfs], pdf.weights)
    elif kind == "bingham":
        tgt = oracle.Target.bingham(pdf.A)
    else:
        tgt = oracle.Target.curve_vmf(pdf.curve.knots, pdf.kappa)
    d = tgt.d
    x0 = oracle.sample_sphere(21, n_chains, d)
    init = x0[0] if n_chains == 1 else x0
    seed = 3521
    s = ns["HipShrinkageSphericalSliceSampler"](pdf, init.copy(), seed)
    out = s.sample(50, 10)
    key = int(np.random.default_rng(seed).integers(2**63))                       # what the block keys the Philox stream with
    want = oracle.run(tgt, x0, 59, seed=key, n_threads=8)
    chain = np.concatenate([x0[:, None, :], want["samples"]], axis=1)[:, 10:]
    assert out.shape == ((50, d) if n_chains == 1 else (n_chains, 50, d))
    got = out[None] if n_chains == 1 else out
    assert np.max(np.abs(got - chain)) < 1e-10
    assert s.n_reject == int(want["n_reject"].sum())
    assert np.array_equal(np.atleast_2d(s.state), got[:, -1])
    # return_all_samples keeps the burn-in rows; a second call continues from the state the first one left
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]                                         # (after the first call: code objects are loaded)
    s2 = ns["HipShrinkageSphericalSliceSampler"](pdf, init.copy(), seed)
    allrows = s2.sample(50, 10, return_all_samples=True)
    all3 = allrows[None] if n_chains == 1 else allrows
    assert all3.shape[1] == 60 and np.array_equal(all3[:, 10:], got) and np.array_equal(all3[:, 0], x0)
    # nothing stays allocated on the device
    c = ns["_lib"].calls
    assert c["gsss_malloc"] == c["gsss_free"] == 10 and c["gsss_target_create"] == c["gsss_target_destroy"] == 2
    assert c["gsss_memcpy_h2d"] == 2 and c["gsss_memcpy_d2h"] == 4 and c["gsss_memset"] == 2
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) <= 8 << 20, "device memory left allocated by the binding"

"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures the
reference produced and against the CPU oracle on the same seeded inputs.

Tolerance: 1e-10 absolute on state components and relative-or-absolute on log-densities -- the
bar BASELINE.json's north_star states; integer outputs (tries, rejections, error bits) exact.
"""
import numpy as np
import pytest

from conftest import golden, trajectory_names
from helpers import modes_for, pad_replay, product_target, variants_for

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def _kat_cases():
    z = golden("logprob_kat.npz")
    return sorted({k.split("__")[0] for k in z.files})


@pytest.mark.parametrize("name", _kat_cases())
def test_logprob_kat(gs, name):
    z = golden("logprob_kat.npz")
    pdf = product_target(z, prefix=f"{name}__target_")
    X, want = z[f"{name}__X"], z[f"{name}__logp"]
    got = pdf.log_prob(X)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < TOL
    one = pdf.log_prob(X[3])
    assert isinstance(one, float) and abs(one - want[3]) / max(1.0, abs(want[3])) < TOL


# the kernel families of packed fast mode -- what bench.py times (gsss_fast_*.hip pick one per shape)
PACKED_FAST_KERNELS = ("screened_kernel", "curvespec_kernel", "coopfast_kernel")


def _packed_kernel(s):
    """Name of the kernel a packed fast-mode launch of sampler `s` runs (what a rocprofv3 trace shows)."""
    return s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()


def _traj_params():
    out = []
    for name in trajectory_names("shrink") + trajectory_names("reject"):
        d = int(golden(name + ".npz")["x0"].shape[0])
        z = golden(name + ".npz")
        for v in variants_for(d):
            out.append((name, v, "exact", "auto"))
        out.append((name, 0, "exact", "packed"))
        if "fast" in modes_for(z):
            out.append((name, 0, "fast", "auto"))
            out.append((name, 0, "fast", "packed"))
    return out


@pytest.mark.parametrize("name,variant,mode,placement", _traj_params())
def test_trajectory_replay(gs, name, variant, mode, placement):
    """Replaying the reference's recorded draws through the HIP kernel reproduces the
    reference chain: every state (1e-10), tries per chain and n_reject (exact).  placement 'auto' spreads a
    one-chain ensemble (one wavefront per chain); 'packed' runs the throughput kernels bench.py times
    (screened_kernel / curvespec_kernel / coopfast_kernel with their replay draw source)."""
    z = golden(name + ".npz")
    pdf = product_target(z)
    cls = gs.RejectionSphericalSliceSampler if str(z["sampler"]) == "reject" else gs.ShrinkageSphericalSliceSampler
    s = cls(pdf, z["x0"], seed=1, variant=variant, mode=mode, placement=placement)
    assert s.mode == mode
    if mode == "fast" and placement == "packed":
        assert _packed_kernel(s).startswith(PACKED_FAST_KERNELS), _packed_kernel(s)
    n = len(z["states"]) - 1
    kept = s.advance(n, thin=1, replay=z["draws"][None])
    got = kept[:, :, 0].cpu().numpy()
    assert s.errors[0] == 0
    assert np.max(np.abs(got - z["states"][1:])) < TOL
    assert s.n_reject == int(z["n_reject"])
    assert int(s.n_tries_per_chain[0]) == int(z["tries"].sum())
    assert np.max(np.abs(s.state - z["states"][-1])) < TOL


def _tf_params():
    return [(n, m, p) for n in trajectory_names("shrink") + trajectory_names("reject") for m in modes_for(golden(n + ".npz"))
            for p in ("auto", "packed")]


@pytest.mark.parametrize("name,mode,placement", _tf_params())
def test_trajectory_teacher_forced(gs, name, mode, placement):
    """All steps of the reference chain at once: chain i starts from reference state i and
    replays the draws of step i; compares the next state and the number of tries.  Both placements: the
    packed leg runs every recorded transition of the reference through the throughput kernels."""
    z = golden(name + ".npz")
    pdf = product_target(z)
    states = z["states"]
    cls = gs.RejectionSphericalSliceSampler if str(z["sampler"]) == "reject" else gs.ShrinkageSphericalSliceSampler
    s = cls(pdf, states[:-1], seed=1, mode=mode, placement=placement)
    if mode == "fast" and placement == "packed":
        assert _packed_kernel(s).startswith(PACKED_FAST_KERNELS), _packed_kernel(s)
    s.advance(1, replay=pad_replay(z["draws"], z["step_draw_offset"]))
    assert np.all(s.errors == 0)
    assert np.max(np.abs(s.state - states[1:])) < TOL
    assert np.array_equal(s.n_tries_per_chain, z["tries"])


ORACLE_CASES = [("vmfmix_readme", 512, 60), ("vmfmix_k10_kappa500", 256, 40), ("vmfmix_d10_k5_kappa100", 256, 40),
                ("bingham_d10_vmax30", 256, 40), ("bingham_d5_dense", 256, 40), ("bingham_d50_vmax300", 96, 25),
                ("curve_d10_kappa800", 128, 30), ("curve_d24_kappa800", 64, 20), ("curve_d50_kappa800", 64, 20),
                ("curve_d200_kappa800", 48, 12)]


def _oracle_cases():
    """(fixture, chains, steps, sampler, mode): every fixture target under the shrinkage sampler in the modes built for its shape; the
    rejection sampler on two of them.  (Formed here rather than skipped inside the test: a skip is a line in the driver's record.)"""
    out = []
    for name, n_chains, n_steps in ORACLE_CASES:
        for sampler in ("shrink", "reject"):
            if sampler == "reject" and not name.startswith(("vmfmix_readme", "bingham_d10")):
                continue
            for mode in modes_for(golden(f"traj_{name}.npz")):
                out.append((name, n_chains, n_steps, sampler, mode))
    return out


@pytest.mark.parametrize("name,n_chains,n_steps,sampler,mode", _oracle_cases())
def test_philox_stream_matches_oracle(gs, oracle, name, n_chains, n_steps, sampler, mode):
    """Same seed, same chain ids -> the device's Philox-driven chains equal the oracle's:
    states within 1e-10 after every step, tries / rejections exactly."""
    z = golden(f"traj_{name}.npz")
    if mode == "fast":
        n_chains, n_steps = 4 * n_chains + 77, 2 * n_steps  # ragged tail block, longer chains
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    d = len(z["x0"])
    x0 = oracle.sample_sphere(11, n_chains, d, chain_offset=1000)
    kind = oracle.REJECT if sampler == "reject" else oracle.SHRINK
    want = oracle.run(tgt, x0, n_steps, seed=2024, chain_offset=1000, step_offset=7, sampler=kind, n_threads=8)
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    # both placements: packed = lane-per-chain kernels, spread = one wavefront per chain (in fast mode the
    # speculative wave kernel)
    for placement in ("packed", "spread"):
        s = cls(pdf, x0, seed=2024, chain_offset=1000, step_offset=7, mode=mode, placement=placement)
        kept = s.advance(n_steps, thin=1).permute(2, 0, 1).cpu().numpy()  # (chains, steps, d)
        assert np.all(s.errors == 0) and np.all(want["err"] == 0)
        assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
        assert np.array_equal(s.n_reject_per_chain, want["n_reject"])
        assert np.max(np.abs(kept - want["samples"])) < TOL


def test_sample_sphere_matches_oracle(gs, oracle):
    for d in (3, 10, 51):
        got = gs.sample_sphere_device(d - 1, 1000, seed=99, chain_offset=5).T.cpu().numpy()
        want = oracle.sample_sphere(99, 1000, d, chain_offset=5)
        assert np.max(np.abs(got - want)) < 1e-13
        assert np.max(np.abs(np.linalg.norm(got, axis=1) - 1)) < 1e-14
    x = gs.sample_sphere(2, 7, seed=3)
    assert x.shape == (7, 3)
    assert gs.sample_sphere(2, seed=3).shape == (3,)


def test_partition_and_resume_invariance(gs):
    """Chains split over two 'ranks' (chain_offset) and steps split over two calls give the
    same bits as one run -- what makes the multi-GPU sharding exact."""
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(2, 4096, seed=5)
    full = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=42)
    a = full.advance(30, thin=1).cpu().numpy()
    lo = gs.ShrinkageSphericalSliceSampler(pdf, x0[:1500], seed=42, chain_offset=0)
    hi = gs.ShrinkageSphericalSliceSampler(pdf, x0[1500:], seed=42, chain_offset=1500)
    b = np.concatenate([lo.advance(30, thin=1).cpu().numpy(), hi.advance(30, thin=1).cpu().numpy()], axis=2)
    assert np.array_equal(a, b)
    two = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=42)
    two.advance(13)
    two.advance(17)
    assert np.array_equal(two.state, full.state)
    assert np.array_equal(two.n_reject_per_chain, full.n_reject_per_chain)
    # cooperative layout computes the same chains (bitwise is not required, 1e-10 is)
    coop = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=42, variant=8)
    c = coop.advance(30, thin=1).cpu().numpy()
    assert np.max(np.abs(a - c)) < TOL
    assert np.array_equal(coop.n_reject_per_chain, full.n_reject_per_chain)


def test_sample_api_semantics(gs):
    """README.md:44-64 call shape: sample(1000, 100) -> (1000, 3); row bookkeeping of
    mcmc.py:55-77; n_reject and log_prob.num_calls bookkeeping."""
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    type(pdf).log_prob.reset_counters()
    x0 = z["x0"]
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521)
    out = s.sample(1000, 100)
    assert out.shape == (1000, 3) and out.dtype == np.float64
    assert s._step == 1099
    assert pdf.log_prob.num_calls == 1099 + int(s.n_tries_per_chain.sum())
    assert s.n_reject == int(s.n_tries_per_chain.sum()) - 1099
    assert np.array_equal(out[-1], s.state)
    # the same chain again, all rows: row 0 is the initial state, rows [100:] are the call above
    s2 = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521)
    allrows = s2.sample(1000, 100, return_all_samples=True)
    assert allrows.shape == (1100, 3)
    assert np.array_equal(allrows[0], x0)
    assert np.array_equal(allrows[100:], out)
    # many chains: (chains, draws, dims); float burnin = fraction
    x0n = gs.sample_sphere(2, 300, seed=1)
    s3 = gs.ShrinkageSphericalSliceSampler(pdf, x0n, 7)
    o3 = s3.sample(50, burnin=0.2)
    assert o3.shape == (300, 50, 3) and s3._step == 59
    s4 = gs.ShrinkageSphericalSliceSampler(pdf, x0n, 7)
    o4 = s4.sample(60, burnin=0, return_all_samples=True)
    assert np.array_equal(o4[:, 0], x0n) and np.array_equal(o4[:, 10:], o3)
    # iterator protocol
    s5 = gs.ShrinkageSphericalSliceSampler(pdf, x0n, 7)
    y = next(s5)
    assert y.shape == (300, 3) and np.array_equal(y, o4[:, 1])
    # thinning keeps every t-th transition
    s6 = gs.ShrinkageSphericalSliceSampler(pdf, x0n, 7)
    o6 = s6.sample(20, thin=3)
    assert np.array_equal(o6, o4[:, 0:58:3])
    # launcher
    L = gs.SamplerLauncher(pdf, x0n, 50, burnin=0.2, seed=7)
    assert np.array_equal(L.run("sss-shrink"), o3) and L.ssss.n_reject == s3.n_reject
    with pytest.raises(ValueError):
        L.run("nope")

@pytest.mark.parametrize("name,rng,n,kw", [("vmfmix_readme", "philox", 3001, {}), ("vmfmix_readme", "numpy", 1500, {}),
                                          ("vmfmix_readme", "philox", 4000, {"reject": True, "placement": "packed"}),
                                          ("bingham_d10_vmax30", "philox", 5000, {"placement": "packed"}),
                                          ("curve_d10_kappa800", "philox", 2100, {"placement": "packed"}),
                                          ("bingham_d50_vmax300", "philox", 700, {})])
def test_sample_in_blocks_equals_one_launch(gs, name, rng, n, kw):
    """`sample()` -> ndarray runs large ensembles as blocks of chains, the device-to-host copy of one block under the kernel of the
    next (mcmc.py:55-77's return value at the speed of the link): any number of blocks gives the array, the final states, the
    counters and the stream position of the one-launch call, bit for bit -- ragged last block, burn-in, thinning and a second
    call that continues the chains included."""
    import torch
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    d = len(z["x0"])
    x0 = gs.sample_sphere(d - 1, n, seed=5)
    kw = dict(kw)
    cls = gs.RejectionSphericalSliceSampler if kw.pop("reject", False) else gs.ShrinkageSphericalSliceSampler
    runs = {}
    for blocks in (1, 2, 7, 16):
        s = cls(pdf, x0, seed=77, rng=rng, chain_offset=0 if rng == "numpy" else 123, **kw)
        a = s.sample(23, burnin=9, thin=3, blocks=blocks)
        b = s.sample(5, blocks=blocks)                               # continues where the first call ended
        assert a.shape == (n, 23, d) and a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
        assert s._step == 9 + 22 * 3 + 4
        runs[blocks] = (a, b, s.state, s.n_reject_per_chain, s.n_tries_per_chain, s._step)
        t = cls(pdf, x0, seed=77, rng=rng, chain_offset=0 if rng == "numpy" else 123, **kw).sample(23, burnin=9, thin=3, as_tensor=True)
        assert np.array_equal(t.cpu().numpy(), a)
    for blocks in (2, 7, 16):
        for i in range(6):
            assert np.array_equal(runs[1][i], runs[blocks][i]), (blocks, i)
    assert np.array_equal(runs[1][1][:, 0], runs[1][0][:, -1])      # row 0 of the second call = last row of the first


def test_sample_falls_back_to_a_pageable_array_when_pages_cannot_be_locked(gs, monkeypatch):
    """Where the system refuses to page-lock the array (a memlock limit), `sample()` still returns the reference's ndarray
    (mcmc.py:55-77) -- an ordinary one, the same values, with one warning -- rather than failing: the return path degrades, the
    sampling itself has no fallback."""
    from geosss_amd import _lib, _pinned
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere_device(2, 60_000, seed=3).T
    want = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=5).sample(12, blocks=3)

    def refuse(self, nbytes, device):
        raise _lib.GsssError("gsss: hipHostMalloc refused (test)")

    monkeypatch.setattr(_pinned._Block, "__init__", refuse)
    monkeypatch.setattr(_pinned, "_warned", False)
    with pytest.warns(RuntimeWarning, match="page-locked host memory were refused"):
        got = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=5).sample(12, blocks=3)
    assert isinstance(got, np.ndarray) and got.base is None and np.array_equal(got, want)


def test_sample_plans_blocks_for_a_large_array(gs, monkeypatch):
    """400 000 README chains x 40 rows (384 MB): the default plan pipelines it in several blocks, the array is the one-block
    array; samplers with further per-chain launch state (RWMH) and running statistics stay in one launch sequence; page-locked
    blocks go back to the pool when the array is dropped and are reused."""
    from geosss_amd import _pinned
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    n = 400_000
    x0 = gs.sample_sphere_device(2, n, seed=3).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=5)
    assert 2 <= s._plan_blocks(0, 40, 1, None) <= 16 and s._plan_blocks(0, 40, 1, 1) == 1
    assert s._plan_blocks(0, 2, 1, None) == 1                        # 19 MB: one piece
    a = s.sample(40)
    b = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=5).sample(40, blocks=1)
    assert np.array_equal(a, b)
    addr = a.ctypes.data
    _pinned.trim()
    del a
    c = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=5).sample(40)
    assert c.ctypes.data == addr and np.array_equal(c, b)           # the dropped array's pages, again
    m = gs.MetropolisHastings(pdf, x0[:1000], seed=5)
    assert m._plan_blocks(0, 10**6, 1, None) == 1
    with pytest.raises(ValueError):
        m._plan_blocks(0, 10, 1, 2)
    _pinned.trim()



def test_error_reporting(gs):
    from geosss_amd._lib import GsssError
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(2, 128, seed=1)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 1, max_tries=1)
    s.advance(5)
    assert np.any(s.errors & 1) and np.all((s.errors & ~1) == 0)
    with pytest.raises(GsssError):
        s._check_errors()
    # replay stream too short -> flagged, not read out of bounds
    s = gs.ShrinkageSphericalSliceSampler(pdf, z["x0"], 1)
    s.advance(50, replay=z["draws"][None, :40])
    assert s.errors[0] & 4
    # NaN state -> non-finite log_prob -> flagged instead of spinning forever
    bad = np.array(x0)
    bad[3] = np.nan
    s = gs.ShrinkageSphericalSliceSampler(pdf, bad, 1)
    s.advance(3)
    e = s.errors
    assert e[3] & 2 and np.all(np.delete(e, 3) == 0)
    with pytest.raises(ValueError):
        gs.ShrinkageSphericalSliceSampler(pdf, np.zeros(4), 1)
    with pytest.raises(ValueError):
        pdf.log_prob(np.zeros((5, 4)))
    big = gs.Bingham(np.diag(np.arange(200.0)))
    with pytest.raises(ValueError):  # fast mode is not built for this shape: refused, no silent fallback
        gs.ShrinkageSphericalSliceSampler(big, np.eye(200)[0], 1, mode="fast").advance(1)


@pytest.mark.parametrize("d", [10, 50, 200])
def test_group_speculative_curve_kernel_error_paths(gs, oracle, d):
    """The group-speculative curve kernel (packed fast mode, d >= 4): max_tries, a NaN state and a replay stream that runs
    out flag the chain concerned -- and only it --, counters stay consistent, nothing spins."""
    z = golden(f"traj_curve_d{d}_kappa800.npz")
    pdf = product_target(z)
    n = 40
    x0 = gs.sample_sphere(d - 1, n, seed=4)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 2, mode="fast", placement="packed", max_tries=3)
    assert s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode().startswith("curvespec_kernel")
    s.advance(30)
    assert np.any(s.errors & 1) and np.all((s.errors & ~1) == 0)          # kappa = 800 needs ~7 tries a step: some chains give up
    ok = s.errors == 0
    assert np.all(s.n_tries_per_chain[ok] <= 3 * 30) and np.all(s.n_tries_per_chain[ok] >= 30)
    bad = np.array(x0)
    bad[7, 1] = np.nan                                                    # one NaN component: every kernel family flags it
    for kw in (dict(mode="fast", placement="packed"), dict(mode="fast", placement="packed", screen=False),
               dict(mode="fast", placement="spread"), dict(mode="exact", placement="packed")):
        s = gs.ShrinkageSphericalSliceSampler(pdf, bad, 2, **kw)
        s.advance(4)
        assert s.errors[7] & 2 and np.all(np.delete(s.errors, 7) == 0), kw
    # replay: chain 0 has the reference's full stream, chain 1 a truncated one
    m = 12
    need = int(z["step_draw_offset"][m])
    full = np.stack([z["draws"][:need], z["draws"][:need]])
    t = gs.ShrinkageSphericalSliceSampler(pdf, np.stack([z["x0"], z["x0"]]), 1, mode="fast", placement="packed")
    t.advance(m, replay=full)
    assert np.all(t.errors == 0) and np.max(np.abs(t.state - z["states"][m])) < TOL
    t = gs.ShrinkageSphericalSliceSampler(pdf, z["x0"], 1, mode="fast", placement="packed")
    t.advance(m, replay=z["draws"][None, : need - 2])
    assert t.errors[0] & 4


def test_layout_round_trip(gs):
    import torch
    lib = gs._lib.load()
    for n, d in ((1, 3), (1000, 3), (777, 10), (65, 200)):
        x = torch.randn(n, d, dtype=torch.float64, device="cuda")
        c = torch.empty(d, n, dtype=torch.float64, device="cuda")
        r = torch.empty(n, d, dtype=torch.float64, device="cuda")
        gs._lib.check(lib.gsss_rows_to_components(x.data_ptr(), c.data_ptr(), n, d, 0, None))
        gs._lib.check(lib.gsss_components_to_rows(c.data_ptr(), r.data_ptr(), n, d, 0, None))
        torch.cuda.synchronize()
        assert torch.equal(c, x.T.contiguous()) and torch.equal(r, x)
    s = torch.randn(7, 3, 130, dtype=torch.float64, device="cuda")
    o = torch.empty(130, 7, 3, dtype=torch.float64, device="cuda")
    gs._lib.check(lib.gsss_samples_to_chains(s.data_ptr(), o.data_ptr(), 130, 7, 3, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(o, s.permute(2, 0, 1).contiguous())


SYNTH = [("vmf", 11, 6), ("vmf", 13, 4), ("vmf", 14, 2), ("vmf", 12, 7), ("vmf", 16, 10), ("vmf", 15, 11), ("bingham", 11, 0), ("bingham", 14, 0), ("bingham", 16, 0), ("bingham_diag", 12, 0),
         ("bingham_diag", 15, 0), ("bingham_diag", 16, 0),
         ("bingham", 12, 0), ("bingham", 40, 0), ("bingham", 100, 0), ("bingham", 126, 0), ("bingham", 127, 0), ("bingham", 128, 0),
         # dense A beyond what LDS holds (d > 128): the exact kernels read its rows from global memory (Bingham<V>::Ag)
         ("bingham", 129, 0), ("bingham", 200, 0), ("bingham", 300, 0),
         # d = 513 .. 2048: sixty-four lanes with sixteen / thirty-two slots each (round 4)
         ("vmf", 1025, 4), ("curve", 2048, 10), ("bingham", 1100, 0),
         ("bingham", 600, 0), ("bingham_diag", 1024, 0), ("vmf", 513, 3), ("vmf", 1000, 12), ("curve", 700, 10), ("curve", 1024, 17),
         # target rows read from global memory: more components / knots than a workgroup's LDS holds at that d
         ("vmf", 600, 40), ("vmf", 3, 7000), ("curve", 300, 60), ("curve", 1000, 25), ("vmf", 100, 200), ("vmf", 16, 3), ("vmf", 50, 5), ("vmf", 200, 10), ("vmf", 7, 2), ("bingham", 24, 0), ("bingham", 3, 0),
         ("curve", 6, 10), ("curve", 12, 10), ("curve", 100, 10), ("curve", 300, 10), ("curve", 9, 7),
         ("curve", 9, 10), ("curve", 15, 10), ("curve", 18, 10), ("curve", 21, 10), ("bingham", 7, 0), ("bingham", 9, 0),
         ("vmf", 3, 6), ("vmf", 3, 8), ("vmf", 5, 5), ("vmf", 10, 3), ("vmf", 10, 10),
         # any K <= 16 at any d: component buckets with exact padding (MixtureModel takes any K, distributions.py:209-227)
         ("curve", 9, 4), ("curve", 10, 12), ("curve", 7, 10), ("curve", 30, 16), ("curve", 50, 6), ("curve", 100, 15), ("curve", 3, 2),
         ("vmf", 3, 7), ("vmf", 6, 3), ("vmf", 8, 13), ("vmf", 9, 16), ("vmf", 4, 1), ("vmf", 7, 4), ("vmf", 30, 7), ("vmf", 100, 12),
         # the group-speculative curve kernel's layout boundaries (lanes per chain 4 | 16 at d = 16 | 17, quads per lane at
         # d = 64 | 65, 128 | 129, 192 | 193, 256; knot builds 10 | 17; every lane holding normals: d = 13 .. 16, 61 .. 64)
         ("curve", 9, 2), ("curve", 13, 10), ("curve", 16, 10), ("curve", 17, 3), ("curve", 12, 16), ("curve", 61, 10),
         ("curve", 64, 11), ("curve", 65, 10), ("curve", 128, 5), ("curve", 129, 10), ("curve", 192, 17), ("curve", 193, 10),
         ("curve", 256, 10), ("curve", 4, 10), ("curve", 5, 3), ("curve", 8, 10),
         # round 5, the uneven layouts (three quads + one tail component per lane): d = 49 .. 52 in four-lane groups, 97 .. 104 in
         # eight-lane groups, 193 .. 208 in sixteen-lane groups -- one tail component, a full tail (every lane's slot taken: block 0 and the tries get a round of their own)
         ("curve", 49, 10), ("curve", 51, 7), ("curve", 52, 10), ("curve", 97, 10), ("curve", 101, 3), ("curve", 104, 10),
         ("curve", 200, 10), ("curve", 205, 4), ("curve", 208, 10),   # (193: above, with the layout boundaries)
         # ... and two tail components per lane (d = 21 .. 24 in <4, 1, +2>, ...), one behind one and two quads (d = 17 .. 20, 33 .. 36)
         ("curve", 18, 10), ("curve", 20, 4), ("curve", 22, 10), ("curve", 23, 7), ("curve", 34, 10), ("curve", 36, 3), ("curve", 37, 10),
         ("curve", 39, 5), ("curve", 53, 10), ("curve", 56, 10), ("curve", 105, 10), ("curve", 111, 6), ("curve", 209, 10),
         # cooperative mixture kernels: slots per lane 4 | 8 | 16 at d = 64 | 65, 128 | 129; up to 256
         ("vmf", 64, 3), ("vmf", 65, 5), ("vmf", 128, 3), ("vmf", 129, 2), ("vmf", 256, 16)]


@pytest.mark.parametrize("kind,d,k", SYNTH)
def test_synthetic_shapes_match_oracle(gs, oracle, kind, d, k, monkeypatch):
    """Shapes without a reference fixture (every kernel family and layout boundary): device chains
    (default mode) equal the oracle's on the same Philox stream; log_prob agrees as well.  (The lane kernels pack an ensemble of
    this size one chain per lane; "packed2" forces the large ensembles' two chains per lane, GSSS_ONE_PER_LANE=0.)"""
    rng = np.random.default_rng(1000 * d + k)
    if kind == "vmf":
        mu = 40.0 * oracle.sample_sphere(5, k, d)
        w = rng.uniform(0.5, 2.0, k)
        pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mu], w)
        tgt = oracle.Target.vmf_mixture(mu, w)
    elif kind.startswith("bingham"):
        pdf = gs.random_bingham(d=d, vmax=25.0, vmin=-2.0, eigensystem=kind == "bingham_diag", seed=d)
        tgt = oracle.Target.bingham(pdf.A)
        kind = "bingham"
    else:
        knots = gs.brownian_curve(k, d, 0.5, seed=d)
        pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), 300.0)
        tgt = oracle.Target.curve_vmf(knots, 300.0)
    n_chains, n_steps = (70, 12) if d > 64 else (333, 25)
    x0 = oracle.sample_sphere(3, n_chains, d)
    want = oracle.run(tgt, x0, n_steps, seed=77, n_threads=8)
    for placement in ("auto", "packed", "packed2"):  # small ensemble: one wavefront per chain; packed: the throughput kernels
        if placement == "packed2":
            if d > 10 or kind == "curve":
                continue                                       # not a lane kernel: nothing parks a second chain
            monkeypatch.setenv("GSSS_ONE_PER_LANE", "0")
            placement = "packed"
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=77, placement=placement)
        if (kind == "vmf" and k <= 16 and d <= 256) or (kind == "curve" and d <= 256):
            assert s.mode == "fast"
        if kind == "curve" and 4 <= d <= 256 and placement == "packed":      # the group-speculative kernel from d = 4 on
            assert _packed_kernel(s).startswith("curvespec_kernel")
        if kind != "curve" and d <= 16 and placement == "packed" and (kind == "bingham" or k <= (10 if d > 10 else 16)):
            assert _packed_kernel(s).startswith("screened_kernel")            # lane kernels up to d = 16 (round 4: 11 .. 16)
        kept = s.advance(n_steps, thin=1).permute(2, 0, 1).cpu().numpy()
        assert np.all(s.errors == 0)
        assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
        assert np.max(np.abs(kept - want["samples"])) < TOL
    lp = pdf.log_prob(kept[:, -1])
    ref = tgt.log_prob(kept[:, -1])
    assert np.max(np.abs(lp - ref) / np.maximum(1, np.abs(ref))) < TOL


@pytest.mark.parametrize("d,k,even", [(18, 10, "curvespec_kernel<4, 2, 10"), (24, 10, "curvespec_kernel<4, 2, 10"), (36, 7, "curvespec_kernel<4, 3, 10"),
                                      (50, 10, "curvespec_kernel<4, 4, 10"), (55, 10, "curvespec_kernel<4, 4, 10"), (100, 7, "curvespec_kernel<8, 4, 10"),
                                      (110, 10, "curvespec_kernel<8, 4, 10"), (200, 10, "curvespec_kernel<16, 4, 10")])
def test_even_layouts_behind_the_switch_match_oracle(gs, oracle, d, k, even, monkeypatch):
    """GSSS_CURVE_TAIL=0: the four-quad builds the uneven layouts replaced at d = 49 .. 52, 97 .. 104, 193 .. 208 are still what
    statistics launches of those dimensions run, and what the A/B of profiles/r05_ab_curve_tail.log timed: held to the oracle too
    (the uneven layouts by now also cover d = 17 .. 24, 33 .. 40, 53 .. 56 and 105 .. 112)."""
    knots = gs.brownian_curve(k, d, 0.5, seed=d)
    pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), 300.0)
    tgt = oracle.Target.curve_vmf(knots, 300.0)
    n_chains, n_steps = 70, 12
    x0 = oracle.sample_sphere(3, n_chains, d)
    want = oracle.run(tgt, x0, n_steps, seed=77, n_threads=8)
    assert _packed_kernel(gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=77, placement="packed")).endswith((", 10, +1>", ", 10, +2>"))
    monkeypatch.setenv("GSSS_CURVE_TAIL", "0")
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=77, placement="packed")
    assert _packed_kernel(s).startswith(even)
    kept = s.advance(n_steps, thin=1).permute(2, 0, 1).cpu().numpy()
    assert np.all(s.errors == 0) and np.array_equal(s.n_tries_per_chain, want["n_tries"])
    assert np.max(np.abs(kept - want["samples"])) < TOL


@pytest.mark.parametrize("d", [129, 150, 256, 400])
def test_large_dense_bingham_reads_A_from_global_memory(gs, oracle, d):
    """A dense A of d > 128 (more than 136 KB of rows) does not fit a workgroup's LDS: the exact kernels, log_prob / gradient and
    the baseline samplers read its rows from global memory instead (Bingham<V>::Ag) -- same products in the same order.  A
    Fisher-Bingham target (distributions.py:106-114) with a dense A, against the oracle: slice sampler chains on the Philox stream
    at 1e-10 with exact tries, log_prob and gradient (2 A x, distributions.py:88-89), RWMH and HMC chains."""
    rng = np.random.default_rng(d)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = (q * rng.uniform(0.0, 25.0, d)) @ q.T
    A = 0.5 * (A + A.T)
    b = rng.standard_normal(d) * 3.0
    pdf, tgt = gs.BinghamFisher(A, b), oracle.Target.bingham(A, b)
    n, n_steps = 48, 10
    x0 = oracle.sample_sphere(5, n, d)
    want = oracle.run(tgt, x0, n_steps, seed=9, n_threads=8)
    with pytest.warns(RuntimeWarning, match="no fast-mode kernel"):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=9)
    assert s.mode == "exact"
    kept = s.advance(n_steps, thin=1).permute(2, 0, 1).cpu().numpy()
    assert np.all(s.errors == 0) and np.array_equal(s.n_tries_per_chain, want["n_tries"])
    assert np.max(np.abs(kept - want["samples"])) < TOL
    X = kept[:, -1]
    lp, ref = pdf.log_prob(X), tgt.log_prob(X)
    assert np.max(np.abs(lp - ref) / np.maximum(1, np.abs(ref))) < TOL
    assert np.max(np.abs(pdf.gradient(X[:5]) - np.array([oracle.gradient(tgt, x) for x in X[:5]]))) < 1e-10 * np.abs(A).sum(axis=1).max()
    for cls, kind, kw, okw in ((gs.MetropolisHastings, oracle.RWMH, dict(stepsize=0.05), {}),
                               (gs.SphericalHMC, oracle.HMC, dict(stepsize=0.01, n_steps=4), dict(n_leapfrog=4))):
        okw = dict(okw, stepsize=kw["stepsize"])
        ref = oracle.mh_run(tgt, x0, 8, sampler=kind, adapt_steps=3, seed=2, n_threads=8, **okw)
        m = cls(pdf, x0, 2, **kw)
        m.reset(3)
        m.advance(8)
        assert np.array_equal(m.n_accept_per_chain, ref["n_accept"])
        assert np.max(np.abs(m.state[:, :d] - ref["state"])) < (1e-8 if kind == oracle.HMC else 1e-10)


def test_shape_limits(gs):
    """What the kernels are not built for is refused with a ValueError that names the limit, never a wrong
    answer or a fault: d > 2048.  Inside the limit everything runs (mode auto falls back to the exact kernels where no fast
    kernel is built)."""
    import warnings

    def run(pdf, d):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = gs.ShrinkageSphericalSliceSampler(pdf, gs.sample_sphere(d - 1, 40, seed=1), seed=1)
            s.advance(4)
        assert np.all(s.errors == 0) and np.all(np.isfinite(s.state)) and np.all(np.isfinite(pdf.log_prob(s.state)))
        return s.mode

    def vmf(d, k):
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in 30.0 * gs.sample_sphere(d - 1, k, seed=2)])

    def curve(d, k):
        return gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(k, d, 0.5, seed=4)), 300.0)

    assert run(vmf(3, 500), 3) == "exact" and run(vmf(50, 100), 50) == "exact" and run(vmf(512, 3), 512) == "exact"
    assert run(gs.random_bingham(d=512, vmax=20.0, vmin=0.0, eigensystem=False, seed=3), 512) == "exact"
    assert run(curve(10, 100), 10) == "exact" and run(curve(300, 18), 300) == "exact" and run(curve(512, 10), 512) == "fast"
    assert run(vmf(513, 3), 513) == "exact" and run(curve(1000, 10), 1000) == "exact" and run(vmf(1024, 2), 1024) == "exact"
    assert run(gs.random_bingham(d=600, vmax=20.0, vmin=0.0, eigensystem=True, seed=3), 600) == "exact"
    # target rows beyond a workgroup's LDS are read from global memory (round 4: 60 knots at d = 300, 40 components at d = 600,
    # 9 000 components on S^2 were "target parameters need ... B of LDS" before)
    assert run(curve(300, 60), 300) == "exact" and run(vmf(600, 40), 600) == "exact" and run(vmf(3, 9000), 3) == "exact"
    assert run(vmf(1025, 3), 1025) == "exact" and run(curve(2000, 10), 2000) == "exact"
    assert run(gs.random_bingham(d=1100, vmax=20.0, vmin=0.0, eigensystem=True, seed=3), 1100) == "exact"
    for pdf, d, what in ((vmf(2049, 3), 2049, "max 2048"), (curve(3000, 10), 3000, "max 2048"),
                         (gs.random_bingham(d=2100, vmax=20.0, vmin=0.0, eigensystem=True, seed=3), 2100, "max 2048")):
        with pytest.raises(ValueError, match=what):
            run(pdf, d)


# ------------------------------------------------------------------ numpy's own stream on the device


def test_readme_call_from_seed(gs):
    """README.md:44-64 verbatim with rng='numpy': ShrinkageSphericalSliceSampler(pdf, init, 3521)
    .sample(1000, 100) equals the reference's output, n_reject and num_calls, from the seed alone."""
    z, t = golden("readme_sample_call.npz"), golden("traj_vmfmix_readme.npz")
    pdf = product_target(t)
    type(pdf).log_prob.reset_counters()
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.array([-0.86, 0.19, -0.47]), 3521, rng="numpy")
    assert s.mode == "fast"                      # one wavefront per chain, 8 speculative tries per step
    out = s.sample(1000, 100)
    assert out.shape == (1000, 3)
    assert np.max(np.abs(out - z["samples"])) < TOL
    assert s.n_reject == int(z["n_reject"])
    assert pdf.log_prob.num_calls == int(z["num_calls"])
    # sampler.rng sits where the reference's generator would: both continue identically
    ref = np.random.default_rng(3521)
    ref.bit_generator.state = s.rng.bit_generator.state
    nxt = next(s)
    assert s.rng.random() != ref.random()  # the device consumed the draws of one more step
    assert nxt.shape == (3,)


@pytest.mark.parametrize("name", trajectory_names("shrink") + trajectory_names("reject"))
@pytest.mark.parametrize("mode", ["exact", "auto", "packed"])
def test_reference_chain_from_seed(gs, name, mode):
    """Every golden reference chain reproduced on the GPU from (pdf, x0, seed) alone, by the exact kernels
    and (mode auto: where built) by the speculative one-wavefront-per-chain kernel and (packed) by the lane-per-chain
    kernel that serves numpy's stream to large ensembles (fast_kernel<..., NUMPY>)."""
    z = golden(name + ".npz")
    pdf = product_target(z)
    cls = gs.RejectionSphericalSliceSampler if str(z["sampler"]) == "reject" else gs.ShrinkageSphericalSliceSampler
    s = cls(pdf, z["x0"], int(z["seed"]), rng="numpy", **({"mode": "auto", "placement": "packed"} if mode == "packed" else {"mode": mode}))
    # (a shape without a fast kernel on numpy's stream lands on the exact kernels in every mode: the same check again, no skip)
    n = len(z["states"]) - 1
    out = s.sample(n + 1)
    assert np.array_equal(out[0], z["x0"])
    assert np.max(np.abs(out - z["states"])) < TOL
    assert s.n_reject == int(z["n_reject"])
    # the generator the caller sees has moved exactly as the reference's would have
    ref = np.random.default_rng(int(z["seed"]))
    consumed = len(z["draws"])
    d = len(z["x0"])
    for i in range(n):  # replay the reference's consumption pattern on a plain numpy generator
        ref.standard_normal(d)
        for _ in range(int(z["step_draw_offset"][i + 1] - z["step_draw_offset"][i]) - d):
            ref.random()
    assert consumed == int(z["step_draw_offset"][-1])
    assert s.rng.bit_generator.state["state"] == ref.bit_generator.state["state"]


def test_numpy_stream_many_chains(gs, oracle):
    """One default_rng per chain (SeedSequence.spawn, scripts/bingham.py:87-88): device = oracle's numpy
    stream; cooperative layouts run the identical generator on every lane of a group."""
    z = golden("traj_curve_d50_kappa800.npz")
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    seeds = list(np.random.SeedSequence(48385).spawn(37))
    x0 = oracle.sample_sphere(2, 37, 50)
    want = oracle.run(tgt, x0, 15, numpy_seed=seeds)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, np.random.SeedSequence(48385), rng="numpy")
    got = s.advance(15, thin=1).permute(2, 0, 1).cpu().numpy()
    assert np.max(np.abs(got - want["samples"])) < TOL
    assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
    with pytest.raises(ValueError):  # d = 50: a cooperative shape, fast mode cannot serve numpy's stream
        gs.ShrinkageSphericalSliceSampler(pdf, x0, 1, rng="numpy", mode="fast")


@pytest.mark.parametrize("name,n_chains", [("vmfmix_readme", 2500), ("vmfmix_d10_k5_kappa100", 700), ("bingham_d10_vmax30", 900),
                                           ("vmfmix_k10_kappa500", 1000), ("vmfmix_d4_k4_weighted", 800), ("bingham_d5_dense", 800),
                                           ("binghamfisher_d6", 600), ("curve_d10_kappa800", 600), ("curve_d24_kappa800", 300)])
@pytest.mark.parametrize("sampler", ["shrink", "reject"])
def test_numpy_stream_lane_kernel(gs, oracle, name, n_chains, sampler):
    """numpy's stream for LARGE ensembles of the lane-per-chain shapes: one lane per chain, each with its own PCG64 / ziggurat
    generator (fast_kernel<..., NUMPY>), against the oracle's numpy stream seeded the same way -- states at 1e-10, tries exact,
    and every generator left where the exact kernels leave it (the same number of draws consumed, bit for bit)."""
    z = golden(f"traj_{name}.npz")
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    d = len(z["x0"])
    root = np.random.SeedSequence(2024)
    x0 = oracle.sample_sphere(3, n_chains, d)   # (seed, chains, dimension)
    n_steps = 25 if sampler == "shrink" else 6
    kind = oracle.REJECT if sampler == "reject" else oracle.SHRINK
    want = oracle.run(tgt, x0, n_steps, numpy_seed=list(root.spawn(n_chains)), sampler=kind)
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    # round 4: the screened lane kernel serves numpy's stream too (screened_kernel<.., NUMPY>: mixtures K <= 10 and Bingham
    # targets); screen=False is the all-double fast_kernel<.., NUMPY> of round 3 -- the same chains bit for bit
    a = cls(pdf, x0, np.random.SeedSequence(2024), rng="numpy", mode="fast", placement="packed", screen=False)
    kept_a = a.advance(n_steps, thin=1)
    s = cls(pdf, x0, np.random.SeedSequence(2024), rng="numpy", mode="fast", placement="packed")
    kept_s = s.advance(n_steps, thin=1)
    import torch
    assert torch.equal(kept_s, kept_a) and torch.equal(s._n_tries, a._n_tries) and torch.equal(s._rng_state, a._rng_state)
    got = kept_s.permute(2, 0, 1).cpu().numpy()
    assert np.max(np.abs(got - want["samples"])) < TOL
    assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
    e = cls(pdf, x0, np.random.SeedSequence(2024), rng="numpy", mode="exact")
    e.advance(n_steps)
    import torch
    assert torch.equal(s._rng_state, e._rng_state)
    assert np.array_equal(s._rng_state.cpu().numpy().view(np.uint64), np.asarray(want["pcg"]).reshape(n_chains, 4))
    # and a second launch continues each generator
    more = s.advance(5, thin=1).permute(2, 0, 1).cpu().numpy()
    want2 = oracle.run(tgt, x0, n_steps + 5, numpy_seed=list(np.random.SeedSequence(2024).spawn(n_chains)), sampler=kind)
    assert np.max(np.abs(more - want2["samples"][:, n_steps:])) < TOL


def test_edge_shapes(gs):
    """Empty ensemble, one chain, one draw, ragged block tails, zero steps."""
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.zeros((0, 3)), 1)
    assert s.n_chains == 0
    s.advance(5)
    assert s.sample(3).shape == (0, 3, 3) and s.n_reject == 0 and s.state.shape == (0, 3)
    assert pdf.log_prob(np.zeros((0, 3))).shape == (0,)
    one = gs.ShrinkageSphericalSliceSampler(pdf, z["x0"], 5)
    assert np.array_equal(one.sample(1), z["x0"][None])            # n_samples = 1: the initial state, no step
    assert one._step == 0
    one.advance(0)
    for n in (1, 63, 64, 65, 511, 512, 513, 1025):                    # around wave / workgroup boundaries
        x0 = gs.sample_sphere(2, n, seed=n).reshape(n, 3)
        a = gs.ShrinkageSphericalSliceSampler(pdf, x0, 9, mode="fast")
        b = gs.ShrinkageSphericalSliceSampler(pdf, x0, 9, mode="exact")
        xa, xb = a.sample(8), b.sample(8)
        assert xa.shape == (n, 8, 3) and np.max(np.abs(xa - xb)) < TOL
        assert np.array_equal(a.n_reject_per_chain, b.n_reject_per_chain)


@pytest.mark.parametrize("rng", ["philox", "numpy"])
def test_checkpoint_resume(gs, rng):
    """state_dict / load_state_dict: an interrupted run continues bit-identically (both streams)."""
    z = golden("traj_bingham_d10_vmax30.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(9, 200, seed=4)
    ref = gs.ShrinkageSphericalSliceSampler(pdf, x0, 21, rng=rng)
    ref.advance(30)
    a = gs.ShrinkageSphericalSliceSampler(pdf, x0, 21, rng=rng)
    a.advance(12)
    ck = a.state_dict()
    b = gs.ShrinkageSphericalSliceSampler(pdf, np.zeros_like(x0) + x0[:1], 99 if rng == "philox" else 21, rng=rng)
    b.load_state_dict(ck)
    b.advance(18)
    assert np.array_equal(b.state, ref.state)
    assert np.array_equal(b.n_reject_per_chain, ref.n_reject_per_chain) and b._step == 30


def test_c_abi_argument_errors(gs):
    """The C entry points refuse bad arguments with a negative code and a message (no exceptions cross
    the ABI, nothing is launched)."""
    import ctypes as C
    import torch
    from geosss_amd import _lib
    lib = _lib.load()
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    h = pdf._device_target(0).handle
    st = torch.zeros(3, 8, dtype=torch.float64, device="cuda")
    words = torch.ones(8, 4, dtype=torch.int64, device="cuda")   # stands in for 8 PCG64 states

    def run(**kw):
        a = _lib.RunArgs(state_dev=st.data_ptr(), n_chains=8, n_steps=1, thin=1, seed=1, sampler=0, mode=0,
                         max_tries=10)
        for k, v in kw.items():
            setattr(a, k, v)
        rc = lib.gsss_run(h, C.byref(a), None)
        return rc, lib.gsss_last_error().decode()

    assert run()[0] == 0
    for kw, code in ((dict(thin=0), -1), (dict(n_chains=-1), -1), (dict(max_tries=0), -1), (dict(sampler=5), -1),
                     (dict(mode=7), -1), (dict(state_dev=None), -1), (dict(variant=99), -1),
                     (dict(variant=7), -2),                               # lane10 does not cover d = 3
                     (dict(chain_offset=2**48), -1),
                     (dict(replay_dev=st.data_ptr(), replay_stride=0), -1),
                     (dict(replay_dev=st.data_ptr(), replay_stride=4, rng_state_dev=words.data_ptr()), -1),
                     (dict(placement=3), -1), (dict(samples_chain_rows=-1), -1)):
        rc, msg = run(**kw)
        assert rc == code and msg, (kw, rc, msg)
    assert run(rng_state_dev=words.data_ptr(), mode=1, placement=1)[0] == 0   # numpy's stream, one lane per chain
    # numpy's stream in fast mode on a cooperative shape (a chain is several lanes'): unsupported
    zc = golden("traj_curve_d50_kappa800.npz")
    pdf_c = product_target(zc)                       # (kept alive: the handle is its)
    hc = pdf_c._device_target(0).handle
    stc = torch.zeros(50, 8, dtype=torch.float64, device="cuda")
    stc[0] = 1.0
    a = _lib.RunArgs(state_dev=stc.data_ptr(), n_chains=8, n_steps=1, thin=1, seed=1, sampler=0, mode=1, max_tries=10,
                     rng_state_dev=words.data_ptr())
    assert lib.gsss_run(hc, C.byref(a), None) == -2 and b"lane-per-chain" in lib.gsss_last_error()
    assert lib.gsss_run(None, None, None) == -1
    assert run(n_chains=0)[0] == 0
    # target descriptions
    out = C.c_void_p()
    mu = np.ones((2, 3))
    bad = _lib.TargetDesc(1, 1, 2, 0, mu.ctypes.data, mu.ctypes.data, None, None, 0.0)   # d < 2
    assert lib.gsss_target_create(C.byref(bad), 0, C.byref(out)) == -1 and not out.value
    bad = _lib.TargetDesc(1, 3, 2, 0, None, None, None, None, 0.0)                        # missing arrays
    assert lib.gsss_target_create(C.byref(bad), 0, C.byref(out)) == -1
    bad = _lib.TargetDesc(9, 3, 2, 0, mu.ctypes.data, mu.ctypes.data, None, None, 0.0)   # unknown kind
    assert lib.gsss_target_create(C.byref(bad), 0, C.byref(out)) == -1
    ok = _lib.TargetDesc(1, 3, 2, 0, mu.ctypes.data, mu.ctypes.data, None, None, 0.0)
    assert lib.gsss_target_create(C.byref(ok), 99, C.byref(out)) == -4                    # no such device
    eye = np.eye(2100)
    big = _lib.TargetDesc(2, 2100, 0, 0, None, None, eye.ctypes.data, None, 0.0)
    assert lib.gsss_target_create(C.byref(big), 0, C.byref(out)) == -2                    # d beyond every layout
    assert lib.gsss_target_dim(h) == 3 and lib.gsss_mode_supported(h, 1) == 1 and lib.gsss_mode_supported(h, 5) == 0
    assert lib.gsss_variant_name(h, 0, 0) == b"lane3" and lib.gsss_variant_name(h, 1, 0) == b"fast-lane"


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_placement_does_not_change_results(gs, mode):
    """Spread (one chain per wavefront, the default for small ensembles) and packed placements of the
    lane-per-chain kernels give bit-identical chains."""
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    x0 = gs.sample_sphere(2, 37, seed=6)
    out = {}
    for pl in ("packed", "spread", "auto"):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 17, mode=mode, placement=pl)
        out[pl] = (s.sample(40, thin=2), s.n_reject_per_chain)
    for pl in ("spread", "auto"):
        assert np.array_equal(out[pl][0], out["packed"][0]) and np.array_equal(out[pl][1], out["packed"][1])
    with pytest.raises(ValueError):
        gs.ShrinkageSphericalSliceSampler(pdf, x0, 1, placement="sideways")


def test_long_chains_stay_on_the_oracle_trajectory(gs, oracle):
    """20 000 steps: rounding differences between device and host arithmetic do not accumulate beyond
    1e-10, no accept decision flips (tries equal exactly), in either kernel family."""
    z = golden("traj_vmfmix_readme.npz")
    pdf, tgt = product_target(z), oracle.Target.from_fixture(z)
    x0 = oracle.sample_sphere(4, 48, 3)
    n_steps = 20_000
    want = oracle.run(tgt, x0, n_steps, seed=31, keep_samples=False, n_threads=8)
    for mode, placement in (("fast", "spread"), ("fast", "packed"), ("exact", "packed")):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=31, mode=mode, placement=placement)
        s.advance(n_steps)
        assert np.array_equal(s.n_tries_per_chain, want["n_tries"]), (mode, placement)
        assert np.max(np.abs(s.state - want["state"])) < TOL, (mode, placement)


_RCCL_BODY = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import geosss_amd as gs
from geosss_amd import ensemble
from conftest import golden
from helpers import product_target
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = {port!r}
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    x = gs.sample_sphere_device(2, 1000, seed=1)
    out = ensemble.gather_states(x, always_collective=True)
    assert out.shape == x.shape and torch.equal(out, x)
    out = ensemble.gather_states(x, always_collective=True, counts=[1000])
    assert torch.equal(out, x)
    t = torch.tensor([5, 7], dtype=torch.int64, device="cuda")
    assert torch.equal(ensemble.reduce_sum(t.clone()), t)
    dist.barrier()
    assert ensemble.shard_bounds(1000) == (0, 1000)
    s = ensemble.sharded_sampler(gs.ShrinkageSphericalSliceSampler, product_target(golden("traj_vmfmix_readme.npz")),
                                 5000, seed=3)
    s.advance(5)
    assert s.n_chains == 5000 and np.all(s.errors == 0)
    print("RCCL-OK")
finally:
    dist.destroy_process_group()
"""


def test_rccl_collectives_single_rank(gs):
    """The exact collective calls of the multi-GPU path (all_gather_into_tensor, all_reduce, barrier over
    the 'nccl' = RCCL backend, device tensors) run on this box with one rank.  In a child process with a time
    limit: a communicator that does not come up on a box must not take the test session with it."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    try:
        r = subprocess.run([sys.executable, "-c", _RCCL_BODY.format(root=ROOT, port=port)], capture_output=True,
                           text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("the RCCL communicator did not come up within 150 s on this box")
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stderr[-3000:]


# ------------------------------------------------------------------ round-2 regressions


@pytest.mark.parametrize("kind", ["bingham_d12", "bingham_d12_dense", "vmfmix_d12_k3", "vmfmix_d14_k5"])
def test_numpy_stream_on_cooperative_fast_shapes(gs, oracle, kind):
    """Shapes whose fast path is the COOPERATIVE kernel (10 < d <= 16) cannot read numpy's stream: mode='auto'
    must fall back to the exact kernels and reproduce the oracle's numpy-stream chains from the seeds, and
    mode='fast' (Python and C ABI) must refuse instead of silently drawing from Philox."""
    import ctypes as C
    rng = np.random.default_rng(12)
    if kind.startswith("bingham"):
        d = 12
        pdf = gs.random_bingham(d=d, vmax=25.0, vmin=0.0, eigensystem=not kind.endswith("dense"), seed=4)
        tgt = oracle.Target.bingham(pdf.A)
    else:
        d, k = (12, 3) if kind == "vmfmix_d12_k3" else (14, 5)
        mus = 40.0 * gs.sample_sphere(d - 1, k, seed=8)
        pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus], rng.uniform(0.5, 2.0, k))
        tgt = oracle.Target.vmf_mixture(mus, pdf.weights)
    n, steps = 24, 20
    seeds = list(np.random.SeedSequence(777).spawn(n))
    x0 = oracle.sample_sphere(3, n, d)
    want = oracle.run(tgt, x0, steps, numpy_seed=seeds)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, np.random.SeedSequence(777), rng="numpy")
    assert s.mode == "exact"
    got = s.advance(steps, thin=1).permute(2, 0, 1).cpu().numpy()
    assert np.max(np.abs(got - want["samples"])) < TOL
    assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
    assert np.array_equal(s._rng_state.cpu().numpy().view(np.uint64)[:, :2], want["pcg"][:, :2])
    # one chain, from an int seed: two different seeds give two different chains (the advisor's symptom was one chain for all)
    a = gs.ShrinkageSphericalSliceSampler(pdf, x0[0], 1, rng="numpy").sample(6)
    b = gs.ShrinkageSphericalSliceSampler(pdf, x0[0], 2, rng="numpy").sample(6)
    assert not np.allclose(a[1:], b[1:])
    with pytest.raises(ValueError):
        gs.ShrinkageSphericalSliceSampler(pdf, x0, 1, rng="numpy", mode="fast")
    # C ABI: fast mode + rng_state_dev on this shape is GSSS_E_UNSUPPORTED, in both placements
    from geosss_amd import _lib
    lib = _lib.load()
    for placement in (1, 2):
        args = _lib.RunArgs()
        args.state_dev = s._state.data_ptr()
        args.n_chains, args.n_steps, args.thin, args.max_tries = n, 1, 1, 1000
        args.mode, args.sampler, args.placement = _lib.MODE_FAST, _lib.SHRINK, placement
        args.rng_state_dev = s._rng_state.data_ptr()
        assert lib.gsss_run(s._target_dev.handle, C.byref(args), None) == -2
        assert b"numpy stream" in lib.gsss_last_error()


@pytest.mark.parametrize("d,k", [(3, 3), (3, 5), (3, 10), (10, 5), (12, 3)])
def test_zero_weight_component_in_fast_mode(gs, oracle, d, k):
    """A mixture weight of zero (log w = -inf, distributions.py:220; scipy's logsumexp ignores the term) must not
    poison the fast kernels' exponentials: fast == exact == oracle."""
    mus = 30.0 * gs.sample_sphere(d - 1, k, seed=21)
    w = np.ones(k)
    w[1] = 0.0
    pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus], w)
    tgt = oracle.Target.vmf_mixture(mus, w)
    n, steps = 333, 30
    x0 = oracle.sample_sphere(5, n, d)
    want = oracle.run(tgt, x0, steps, seed=9, n_threads=8)
    for placement in ("packed", "spread"):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=9, mode="fast", placement=placement)
        got = s.advance(steps, thin=1).permute(2, 0, 1).cpu().numpy()
        assert np.all(s.errors == 0)
        assert np.max(np.abs(got - want["samples"])) < TOL
        assert np.array_equal(s.n_tries_per_chain, want["n_tries"])
    assert np.all(np.isfinite(pdf.log_prob(want["samples"][:, -1])))


@pytest.mark.parametrize("name", ["curve_d50_kappa800", "bingham_d50_vmax300", "curve_d200_kappa800"])
def test_cooperative_fast_resume(gs, name):
    """Cooperative fast kernels: a run split over launches equals the uninterrupted run to rounding (the a_i.x
    recurrence is refreshed from x at launch start, so the split is visible at the 1e-13 level, never in the
    integer outputs); DESIGN.md section 2 states the bitwise claim for the lane kernels and the exact mode only."""
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    d = len(z["x0"])
    x0 = gs.sample_sphere(d - 1, 96, seed=31)
    ref = gs.ShrinkageSphericalSliceSampler(pdf, x0, 17, mode="fast", placement="packed")
    ref.advance(150)
    a = gs.ShrinkageSphericalSliceSampler(pdf, x0, 17, mode="fast", placement="packed")
    for m in (37, 64, 1, 48):
        a.advance(m)
    assert np.max(np.abs(a.state - ref.state)) < 1e-11
    assert np.array_equal(a.n_tries_per_chain, ref.n_tries_per_chain)


def _wide_target(gs, name):
    """Targets of the d = 11 .. 16 lane kernels (no reference fixture at these shapes): wide:<kind>:<d>[:<k>]"""
    _, kind, d, *rest = name.split(":")
    d = int(d)
    if kind == "vmf":
        k = int(rest[0])
        mu = 60.0 * gs.sample_sphere(d - 1, k, seed=100 + d)
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in mu], np.linspace(1.0, 2.0, k)), d
    return gs.random_bingham(d=d, vmax=40.0, vmin=0.0, eigensystem=kind == "bingham_diag", seed=200 + d), d


SCREEN_CASES = [("wide:vmf:12:3", 60_000, 40), ("wide:vmf:16:6", 40_000, 30), ("wide:vmf:13:10", 40_000, 30), ("wide:bingham_diag:12", 60_000, 40), ("wide:bingham_diag:16", 40_000, 30),
                ("wide:bingham:11", 40_000, 30), ("wide:bingham:16", 30_000, 30),
                ("vmfmix_readme", 200_000, 60), ("vmfmix_k10_kappa500", 100_000, 40), ("vmfmix_d10_k5_kappa100", 50_000, 40),
                ("vmfmix_d4_k4_weighted", 50_000, 40), ("bingham_d10_vmax30", 100_000, 60), ("bingham_d5_dense", 100_000, 60),
                ("binghamfisher_d5", 100_000, 60), ("binghamfisher_d6", 50_000, 40), ("curve_d3_kappa300", 50_000, 40),
                ("curve_d10_kappa800", 50_000, 40), ("curve_d10_kappa500", 50_000, 40), ("curve_d24_kappa800", 20_000, 30)]


@pytest.mark.parametrize("name,n_chains,n_steps", SCREEN_CASES)
@pytest.mark.parametrize("sampler", ["shrink", "reject"])
def test_screened_equals_double(gs, name, n_chains, n_steps, sampler):
    """The single-precision screen only ever takes decisions its error margin guarantees, everything else is
    decided and computed in double precision: the screened kernel and the all-double kernel give the SAME chains
    -- states bit for bit, tries and rejections exactly -- over ~10^8 proposals."""
    wide = name.startswith("wide:")
    if wide:
        pdf, d = _wide_target(gs, name)
    else:
        z = golden(f"traj_{name}.npz")
        pdf = product_target(z)
        d = len(z["x0"])
    x0 = gs.sample_sphere_device(d - 1, n_chains, seed=77).T
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    if sampler == "reject":
        n_chains, n_steps = n_chains // 10, n_steps // 2
        x0 = x0[:n_chains]
    out = {}
    probe = cls(pdf, x0[:1], seed=5, mode="fast", placement="packed")
    spec = probe._lib.gsss_kernel_name(probe._target_dev.handle, 1, 0, 1).decode().startswith("curvespec_kernel")
    # "verify": the default kernel with an infinite margin -- every try decided in double precision by the kernel's own
    # arithmetic -- must give the screened run's bits (the group kernels; round 4: the lane kernels of d = 11 .. 16, which have no
    # all-double lane sibling; at d <= 10 the variant is not built in and the run is the default one)
    for screen in (True, False, "verify"):
        s = cls(pdf, x0, seed=5, mode="fast", placement="packed", screen=screen)
        name_k = s._lib.gsss_kernel_name(s._target_dev.handle, 1, {True: 0, False: 100, "verify": 101}[screen], 1).decode()
        want_k = ("coopfast_kernel" if wide else "fast_kernel") if screen is False else ("curvespec_kernel" if spec else "screened_kernel")
        assert name_k.startswith(want_k), name_k
        s.advance(n_steps // 2)
        s.advance(n_steps - n_steps // 2)          # the split exercises the per-launch state hand-over
        assert int((s._err != 0).sum().item()) == 0
        out[screen] = (s.state_device.clone(), s._n_tries.clone(), s._n_reject.clone())
    import torch
    assert torch.equal(out[True][1], out[False][1])
    assert torch.equal(out[True][2], out[False][2])
    if spec or wide:
        # The group-speculative curve kernel (d >= 4) arranges its sums differently from the all-double lane kernel, and the
        # all-double kernel of d = 11 .. 16 is the cooperative one (four lanes per chain): same decisions (integer outputs
        # above), states to rounding.
        assert float((out[True][0] - out[False][0]).abs().max().item()) < 1e-11
    else:
        assert torch.equal(out[True][0], out[False][0])
    for i in range(3):   # bit for bit against ITSELF with the screen's verdicts ignored
        assert torch.equal(out[True][i], out["verify"][i])
    assert int(out[True][1].sum().item()) > (3 if wide else 4) * n_chains * n_steps * (0.9 if sampler == "shrink" else 1.0)


def test_auto_mode_warns_when_it_falls_back_to_the_exact_kernels(gs):
    import warnings
    from geosss_amd import mcmc
    mcmc._warned_shapes.clear()
    big = gs.Bingham(np.diag(np.arange(200.0)))
    with pytest.warns(RuntimeWarning, match="no fast-mode kernel"):
        s = gs.ShrinkageSphericalSliceSampler(big, np.eye(200)[0], 1)
    assert s.mode == "exact"
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # once per shape
        gs.ShrinkageSphericalSliceSampler(big, np.eye(200)[0], 2)
        z = golden("traj_vmfmix_readme.npz")
        assert gs.ShrinkageSphericalSliceSampler(product_target(z), z["x0"], 1).mode == "fast"   # and never for a built shape


@pytest.mark.parametrize("mode,placement", [("exact", "packed"), ("fast", "packed"), ("fast", "spread")])
def test_replay_stream_too_short_is_flagged(gs, mode, placement):
    """A replay stream that runs out sets GSSS_CHAIN_REPLAY_EXHAUSTED on that chain (and only there); the wrapper raises."""
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    n = 50
    need = int(z["step_draw_offset"][n])
    full = np.stack([z["draws"][:need], z["draws"][:need]])
    short = full.copy()
    short[1, need - 3:] = 0.5                                # same length on the device; chain 1 gets a truncated stride below
    s = gs.ShrinkageSphericalSliceSampler(pdf, np.stack([z["x0"], z["x0"]]), 1, mode=mode, placement=placement)
    s.advance(n, replay=full)
    assert np.all(s.errors == 0) and np.max(np.abs(s.state - z["states"][n])) < TOL
    t = gs.ShrinkageSphericalSliceSampler(pdf, z["x0"], 1, mode=mode, placement=placement)
    t.advance(n, replay=z["draws"][None, : need - 3])
    assert t.errors[0] & 4
    with pytest.raises(gs._lib.GsssError):
        t._check_errors()


# ------------------------------------------------------------------ round 3: sliced launches


@pytest.mark.parametrize("name,n_chains", [("curve_d10_kappa800", 60_000), ("curve_d24_kappa800", 60_000), ("curve_d50_kappa800", 40_000),
                                           ("curve_d100_kappa800", 20_000), ("curve_d200_kappa800", 10_000),
                                           ("bingham_d50_vmax300", 40_000)])   # the cooperative kernel (diagonal A, eight-lane groups)
@pytest.mark.parametrize("sampler", ["shrink", "reject"])
def test_sliced_launch_equals_one_workgroup_per_chunk(gs, name, n_chains, sampler, monkeypatch):
    """Ensembles whose chunks do not fit the chip at once are launched SLICED (one workgroup per (chunk, step slice), tickets,
    hand-over of the chunk's state through HBM: SliceSched, gsss_device.h).  Slice boundaries sit where the kernels refresh their
    carried quantities anyway, so every output -- states, retained rows, tries, rejections, error flags -- equals the unsliced
    launch BIT FOR BIT, for any slice length, any step offset and any split of the steps over launches; a chain that stops
    with an error flag in one slice stays stopped in the next."""
    import torch
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    d = len(z["x0"])
    x0 = gs.sample_sphere_device(d - 1, n_chains, seed=41).T
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    steps = (150, 75) if sampler == "shrink" else (130,)
    # shrink: some chains run into max_tries (kappa = 800 needs ~7 tries a step, the d = 50 Bingham target ~6.4)
    max_tries = (16 if name.startswith("bingham") else 20) if sampler == "shrink" else 1 << 20
    out = {}
    for label, env in (("whole", "0"), ("s64", "64"), ("s128", "128")):
        monkeypatch.setenv("GSSS_SLICE_STEPS", env)
        s = cls(pdf, x0, seed=5, mode="fast", placement="packed", step_offset=37, max_tries=max_tries)
        assert _packed_kernel(s).startswith(("curvespec_kernel", "coopfast_kernel"))
        kept = [s.advance(m, thin=7) for m in steps]
        out[label] = (s.state_device.clone(), torch.cat(kept), s._n_tries.clone(), s._n_reject.clone(), s._err.clone())
    err = out["whole"][4]
    ok = err == 0                                # (a stopped chain writes no further rows: those slots of the buffer are unspecified)
    for label in ("s64", "s128"):
        for i in (0, 2, 3, 4):
            assert torch.equal(out["whole"][i], out[label][i]), (label, i)
        assert torch.equal(out["whole"][1][:, :, ok], out[label][1][:, :, ok]), label
    if sampler == "shrink":
        assert 0 < int((err != 0).sum()) < n_chains               # stopped chains and healthy ones, both kinds in every slice
    else:
        assert int((err != 0).sum()) == 0


@pytest.mark.parametrize("name,resident", [("vmfmix_readme", (1280, 1280)), ("vmfmix_k10_kappa500", (768, 768)), ("bingham_d10_vmax30", (768, 768)),
                                           # (resident workgroups with two chains per lane, with one: without the LDS of parked
                                           # chains the d = 10 mixture kernel fits three workgroups per CU instead of two)
                                           ("vmfmix_d10_k5_kappa100", (512, 768))])
@pytest.mark.parametrize("sampler", ["shrink", "reject"])
@pytest.mark.parametrize("per_lane", [2, 1])
def test_sliced_partial_round_of_the_lane_kernels(gs, name, resident, sampler, per_lane, monkeypatch):
    """The lane kernels (two chains per lane) cut only a SMALL last round of workgroups into step slices
    (plan_partial_round, gsss_device.h): an ensemble of k x resident + a few workgroups gives the same bits -- states,
    retained rows, tries, rejections, error flags -- sliced or not, and a chain that stops in one slice stays stopped."""
    import torch
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    d = len(z["x0"])
    # one full round of workgroups + 37 more, the last one ragged; per_lane 1: an ensemble of half the size, which the library
    # packs one chain per lane (256-chain workgroups)
    resident = resident[2 - per_lane]
    monkeypatch.setenv("GSSS_RESIDENT_PER_CU", str(resident // 256))   # the plan of the box the counts above were read on, on every box
    n_chains = (resident + 37) * 256 * per_lane - 100
    if per_lane == 2:
        monkeypatch.setenv("GSSS_ONE_PER_LANE", "0")            # (an ensemble of this size would run one per lane as well)
    x0 = gs.sample_sphere_device(d - 1, n_chains, seed=43).T
    out = {}
    for label, env in (("whole", "0"), ("sliced", "128")):
        monkeypatch.setenv("GSSS_SLICE_STEPS", env)
        cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
        s = cls(pdf, x0, seed=6, mode="fast", placement="packed", step_offset=91, max_tries=24 if sampler == "shrink" else 1 << 20)
        assert _packed_kernel(s).startswith("screened_kernel")
        kept = [s.advance(m, thin=13) for m in ((650, 520) if sampler == "shrink" else (300,))]
        import ctypes as C
        grid, steps = C.c_int64(0), C.c_int32(0)
        frac = C.c_double(-1.0)
        s._lib.gsss_last_launch(C.byref(grid), C.byref(steps), C.byref(frac))
        assert (0.0 < frac.value < 0.75) if steps.value else frac.value == 0.0   # only a small last round is sliced
        out[label] = (s.state_device.clone(), torch.cat(kept), s._n_tries.clone(), s._n_reject.clone(), s._err.clone(), int(steps.value))
    assert out["whole"][5] == 0 and out["sliced"][5] == 128
    err = out["whole"][4]
    ok = err == 0
    for i in (0, 2, 3, 4):
        assert torch.equal(out["whole"][i], out["sliced"][i]), i
    assert torch.equal(out["whole"][1][:, :, ok], out["sliced"][1][:, :, ok])
    if sampler == "shrink":
        assert 0 < int((err != 0).sum()) < n_chains
    else:
        assert int((err != 0).sum()) == 0


def test_sliced_launch_long_hand_over_chain(gs, monkeypatch):
    """The bench's own launch shape -- 10^5 curve chains x 1000 steps, 1563 chunks x 16 slices of 64 steps, every chunk handed from
    CU to CU (and XCD to XCD) fifteen times -- against the unsliced launch: every state bit, every counter.  A stale line
    anywhere along a hand-over chain would show here."""
    import torch
    z = golden("traj_curve_d10_kappa800.npz")
    pdf = product_target(z)
    n = 100_000
    x0 = gs.sample_sphere_device(9, n, seed=47).T
    out = {}
    for label, env in (("whole", "0"), ("s64", "64"), ("s128", "128")):
        monkeypatch.setenv("GSSS_SLICE_STEPS", env)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=8, mode="fast", placement="packed")
        kept = s.advance(1000, thin=100)
        s.advance(1000)
        out[label] = (s.state_device.clone(), kept.clone(), s._n_tries.clone(), s._n_reject.clone(), s._err.clone())
    for label in ("s64", "s128"):
        for i in range(5):
            assert torch.equal(out["whole"][i], out[label][i]), (label, i)
    assert int((out["whole"][4] != 0).sum()) == 0


def test_sliced_partial_round_long_hand_over_chain(gs, monkeypatch):
    """The bench's headline launch -- 10^6 README chains x 1000 steps, the last partial round of workgroups cut into sixteen
    64-step slices (eight of 128), each chunk's state written THROUGH from XCD to XCD (SliceSched::hand_over: no release, only
    the consumer's acquire) -- against the unsliced launch: every state bit, every retained row, every counter."""
    import torch
    z = golden("traj_vmfmix_readme.npz")
    pdf = product_target(z)
    n = 1_000_000
    x0 = gs.sample_sphere_device(2, n, seed=49).T
    monkeypatch.setenv("GSSS_RESIDENT_PER_CU", "5")          # five workgroups per CU: the headline's plan (1.53 rounds, 674 chunks sliced)
    out, sliced = {}, {}
    for label, env in (("whole", "0"), ("s64", "64"), ("s128", "128")):
        monkeypatch.setenv("GSSS_SLICE_STEPS", env)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=9, mode="fast", placement="packed")
        assert _packed_kernel(s).startswith("screened_kernel")
        kept = s.advance(1000, thin=100)
        s.advance(1000)
        import ctypes as C
        steps = C.c_int32(0)
        s._lib.gsss_last_launch(None, C.byref(steps), None)
        sliced[label] = int(steps.value)
        out[label] = (s.state_device.clone(), kept.clone(), s._n_tries.clone(), s._n_reject.clone(), s._err.clone())
    assert sliced == {"whole": 0, "s64": 64, "s128": 128}
    for label in ("s64", "s128"):
        for i in range(5):
            assert torch.equal(out["whole"][i], out[label][i]), (label, i)
    assert int((out["whole"][4] != 0).sum()) == 0


@pytest.mark.parametrize("name", ["vmfmix_readme", "vmfmix_k10_kappa500", "bingham_d10_vmax30"])
def test_one_chain_per_lane_equals_two(gs, name, monkeypatch):
    """Small and mid-size ensembles run the lane kernels with ONE chain per lane (256-chain workgroups, RunBlock::one_per_lane);
    large ones park a second chain per lane.  Chains are keyed by their id: the first 40 000 chains of a 700 000-chain ensemble
    run two per lane (GSSS_ONE_PER_LANE=0) equal the same 40 000 run on their own (one per lane), bit for bit."""
    import torch
    z = golden(f"traj_{name}.npz")
    pdf = product_target(z)
    d = len(z["x0"])
    n_big, n_small = 700_000, 40_000
    x0 = gs.sample_sphere_device(d - 1, n_big, seed=53).T
    out = {}
    for label, n in (("big", n_big), ("small", n_small)):
        if label == "big":
            monkeypatch.setenv("GSSS_ONE_PER_LANE", "0")
        else:
            monkeypatch.delenv("GSSS_ONE_PER_LANE")
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0[:n].contiguous(), seed=10, mode="fast", placement="packed")
        assert _packed_kernel(s).startswith("screened_kernel")
        kept = s.advance(120, thin=40)
        import ctypes as C
        grid = C.c_int64(0)
        s._lib.gsss_last_launch(C.byref(grid), None, None)
        out[label] = (s.state_device[:, :n_small].clone(), kept[:, :, :n_small].clone(), s._n_tries[:n_small].clone(), int(grid.value))
    assert out["small"][3] == (n_small + 255) // 256          # one chain per lane: 256-chain workgroups
    assert out["big"][3] >= (n_big + 511) // 512 and out["big"][3] < (n_big + 255) // 256
    for i in range(3):
        assert torch.equal(out["big"][i], out["small"][i]), i


def test_long_launch_gets_longer_slices(gs, monkeypatch):
    """A launch is cut into at most 64 slices per chunk (slice_length, gsss_device.h): a C-ABI caller that asks for 20 000 steps
    in ONE gsss_run gets 320-step slices instead of 157 of 128 steps waiting on each other in a row -- and, as for any slice
    length, the same bits as the unsliced launch.  (The Python classes cap a launch at 4096 steps; the cap is lifted here.)"""
    import ctypes as C
    import torch
    from geosss_amd import mcmc
    monkeypatch.setattr(mcmc, "_MAX_STEPS_PER_LAUNCH", 1 << 30)
    z = golden("traj_curve_d10_kappa800.npz")
    pdf = product_target(z)
    n, n_steps = 52_000, 20_000                                      # 813 chunks of 64 chains: more than the chip holds at once
    monkeypatch.setenv("GSSS_RESIDENT_PER_CU", "3")                  # (three workgroups per CU, 768 resident -- planned so on every box)
    x0 = gs.sample_sphere_device(9, n, seed=59).T
    out, slice_steps = {}, {}
    for label, env in (("whole", "0"), ("default", None)):
        if env is None:
            monkeypatch.delenv("GSSS_SLICE_STEPS", raising=False)
        else:
            monkeypatch.setenv("GSSS_SLICE_STEPS", env)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=12, mode="fast", placement="packed", step_offset=5)
        kept = s.advance(n_steps, thin=5000)
        steps = C.c_int32(0)
        s._lib.gsss_last_launch(None, C.byref(steps), None)
        slice_steps[label] = int(steps.value)
        out[label] = (s.state_device.clone(), kept.clone(), s._n_tries.clone(), s._n_reject.clone(), s._err.clone())
    assert slice_steps["whole"] == 0
    assert slice_steps["default"] == 320                              # ceil(20 000 / 64) rounded up to a multiple of 64
    for i in range(5):
        assert torch.equal(out["whole"][i], out["default"][i]), i
    assert int((out["whole"][4] != 0).sum()) == 0


@pytest.mark.parametrize("name", ["bingham_d10_vmax30", "bingham_d5_dense", "binghamfisher_d5", "binghamfisher_d6", "vmfmix_readme",
                                  "wide:bingham_diag:3", "wide:bingham:7", "wide:bingham_diag:9", "wide:bingham:9", "wide:bingham_diag:8"])
def test_rows_held_back_in_lds_land_where_they_belong(gs, name, monkeypatch):
    """One chain per lane, (chains, draws, dims) output, Bingham targets at d <= 10 (the mixture runs the plain kernel either way:
    the control): rows of 8 d bytes that do not end on a 32-byte sector are held back in LDS
    until their run does (screened_kernel<.., STAGE>, RunBlock::stage_rows) -- fewer partial sectors, the SAME array: with the staging
    switched off (GSSS_STAGE_ROWS=0) every retained row of every chain is bit-identical, for odd and even row offsets (sample()
    starts at row 1), thinning, runs of rows split over launches (the Python classes cap a launch at 4096 steps) and chains that
    stop early (max_tries)."""
    import torch
    if name.startswith("wide:"):                             # (every row size modulo a sector: d = 3, 7, 9 wait for four rows, d = 8 for none)
        pdf, d = _wide_target(gs, name)
    else:
        z = golden(f"traj_{name}.npz")
        pdf = product_target(z)
        d = len(z["x0"])
    n = 30_000                                               # one chain per lane at this size
    x0 = gs.sample_sphere_device(d - 1, n, seed=61).T
    out = {}
    for label, env in (("staged", None), ("plain", "0")):
        if env is None:
            monkeypatch.delenv("GSSS_STAGE_ROWS", raising=False)
        else:
            monkeypatch.setenv("GSSS_STAGE_ROWS", env)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=14, mode="fast", placement="packed")
        assert _packed_kernel(s).startswith("screened_kernel")
        a = s.sample(23, burnin=3, thin=7, as_tensor=True)                # rows 1 .. 22 of a 23-row run, one launch
        b = s.sample(12, thin=500, as_tensor=True)                        # 5500 steps: two launches, 8 + 3 rows
        buf = torch.zeros((n, 9, d), dtype=torch.float64, device="cuda")  # an odd run length: chains start on either sector phase
        s.advance(8 * 3 + 2, thin=3, out=buf, chain_major=True, row0=1)   # ... and the launch goes on for two steps behind its last row
        buf2 = torch.zeros((n, 3, d), dtype=torch.float64, device="cuda")
        s.advance(19, thin=5, out=buf2, chain_major=True)                 # rows 0 .. 2, the last one held back when the launch ends 4 steps later
        t = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=14, mode="fast", placement="packed", max_tries=14)
        c = torch.zeros((n, 40, d), dtype=torch.float64, device="cuda")
        t.advance(40, thin=1, out=c, chain_major=True)                    # some chains stop early: their held-back rows still go out
        out[label] = (a.clone(), b.clone(), buf.clone(), c.clone(), t._err.clone(), s.state_device.clone(), buf2.clone())
    for i in range(7):
        assert torch.equal(out["staged"][i], out["plain"][i]), i
    assert float(out["staged"][6].abs().min()) > 0.0
    stopped = int((out["plain"][4] != 0).sum())
    assert stopped < n and (stopped > 0 or name.startswith("wide:"))   # (the reference's targets: some chains do stop at 14 tries)
    assert float(out["staged"][2][:, 0].abs().max()) == 0.0 and float(out["staged"][2][:, 1:].abs().min()) > 0.0  # row 0 untouched, rows 1 .. 8 written

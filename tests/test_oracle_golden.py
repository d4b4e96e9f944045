"""Pins the CPU oracle (oracle/gsss_oracle.c) to the reference.

Every fixture under tests/golden/ was produced by tests/golden/make_golden.py RUNNING the
reference (geosss @ /root/reference) in the build container; nothing here reads the reference.
Tolerance: 1e-12 absolute per state component / log-density (the north-star bar is 1e-10).
"""
import numpy as np
import pytest

from conftest import golden, trajectory_names

TOL = 1e-12


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kats:
        got = oracle.philox4x32_10(ctr, key)
        assert [int(v) for v in got] == want


def test_stream_uniform_range_and_determinism(oracle):
    u = np.array([oracle.stream_block(3521, c, s, b) for c in range(4) for s in range(4) for b in range(4)])
    assert np.all((u >= 0) & (u < 1))
    assert len(np.unique(u)) == u.size
    assert np.array_equal(oracle.stream_block(3521, 2, 3, 1), oracle.stream_block(3521, 2, 3, 1))
    # chain / step ids above 2^32 are distinct streams
    assert not np.array_equal(oracle.stream_block(1, 5, 0, 0), oracle.stream_block(1, 5 + 2**32, 0, 0))
    assert not np.array_equal(oracle.stream_block(1, 5, 7, 0), oracle.stream_block(1, 5, 7 + 2**32, 0))


def test_logsumexp_matches_scipy(oracle):
    from scipy.special import logsumexp
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10):
        for _ in range(50):
            a = rng.normal(scale=30, size=n)
            assert abs(oracle.logsumexp(a) - logsumexp(a)) <= 1e-13 * max(1, abs(logsumexp(a)))
    a = np.array([1.5, 1.5, -3.0])  # tie at the maximum
    assert abs(oracle.logsumexp(a) - logsumexp(a)) < 1e-15


def test_geometry_kat(oracle):
    z = golden("geometry_kat.npz")
    for d in (3, 10, 50):
        X, Z = z[f"d{d}_x"], z[f"d{d}_z"]
        for i in range(len(X)):
            assert np.max(np.abs(oracle.radial_projection(X[i]) - z[f"d{d}_radial"][i])) < TOL
            assert np.max(np.abs(oracle.orthogonal_projection(Z[i], X[i]) - z[f"d{d}_ortho"][i])) < TOL
            assert np.max(np.abs(oracle.spherical_projection(Z[i], X[i]) - z[f"d{d}_spherical"][i])) < TOL
        a, b, q = z[f"d{d}_slerp_a"], z[f"d{d}_slerp_b"], z[f"d{d}_slerp_q"]
        for i in range(len(q)):
            dist, near = oracle.distance_slerp(q[i], a[i], b[i])
            # acos near 1 amplifies rounding: compare cos(dist) tightly, dist loosely
            assert abs(np.cos(dist) - np.cos(z[f"d{d}_slerp_dist"][i])) < 1e-12
            assert np.max(np.abs(near - z[f"d{d}_slerp_near"][i])) < TOL
            assert np.max(np.abs(oracle.find_nearest(z[f"d{d}_knots"], q[i]) - z[f"d{d}_nearest"][i])) < TOL


def _kat_cases():
    z = golden("logprob_kat.npz")
    return sorted({k.split("__")[0] for k in z.files})


@pytest.mark.parametrize("name", _kat_cases())
def test_logprob_kat(oracle, name):
    z = golden("logprob_kat.npz")
    tgt = oracle.Target.from_fixture(z, prefix=f"{name}__target_")
    X, want = z[f"{name}__X"], z[f"{name}__logp"]
    got = tgt.log_prob(X)
    scale = np.maximum(1.0, np.abs(want))
    assert np.max(np.abs(got - want) / scale) < TOL
    assert abs(tgt.log_prob(X[0]) - want[0]) / scale[0] < TOL
    assert np.max(np.abs(got - z[f"{name}__logp_batched"]) / scale) < 1e-11


@pytest.mark.parametrize("name", trajectory_names("shrink") + trajectory_names("reject"))
def test_trajectory_replay(oracle, name):
    """Replaying the reference's recorded draws reproduces the reference chain: every state,
    every threshold, the number of tries of every step and n_reject."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    sampler = oracle.REJECT if str(z["sampler"]) == "reject" else oracle.SHRINK
    n_steps = len(z["states"]) - 1
    out = oracle.run(tgt, z["x0"], n_steps, sampler=sampler, replay=z["draws"], trace_threshold=True)
    assert out["err"][0] == 0
    assert np.max(np.abs(out["samples"][0] - z["states"][1:])) < TOL
    assert np.max(np.abs(out["threshold"][0] - z["threshold"])) < 1e-11
    assert out["n_reject"][0] == int(z["n_reject"])
    assert out["n_tries"][0] == int(z["tries"].sum())
    # the recorded margins say no accept decision sits within rounding of its threshold
    assert float(z["min_margin"]) > 1e-8


@pytest.mark.parametrize("name", trajectory_names("shrink"))
def test_trajectory_teacher_forced(oracle, name):
    """Each step on its own: start from the reference state i with the draws of step i."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    off, draws, states = z["step_draw_offset"], z["draws"], z["states"]
    n = len(states) - 1
    width = int(np.max(np.diff(off)))
    replay = np.full((n, width), 0.5)
    for i in range(n):
        replay[i, : off[i + 1] - off[i]] = draws[off[i]: off[i + 1]]
    out = oracle.run(tgt, states[:-1], 1, replay=replay)
    assert np.all(out["err"] == 0)
    assert np.max(np.abs(out["state"] - states[1:])) < TOL
    assert np.array_equal(out["n_tries"], z["tries"])


def test_readme_call_shape_and_counts(oracle):
    """README.md:44-64: sample(1000, 100) keeps the initial state as row 0, runs 1099 steps,
    returns rows [100:]; log_prob.num_calls = steps + tries."""
    z = golden("readme_sample_call.npz")
    t = golden("traj_vmfmix_readme.npz")
    assert np.array_equal(z["samples"], t["states"][100:1100])
    assert int(z["n_reject"]) == int(t["n_reject"])
    assert int(z["num_calls"]) == 1099 + int(t["tries"].sum())


def test_philox_mode_partition_invariance(oracle):
    t = golden("traj_vmfmix_readme.npz")
    tgt = oracle.Target.from_fixture(t)
    x0 = oracle.sample_sphere(7, 32, 3)
    assert np.allclose(np.linalg.norm(x0, axis=1), 1.0)
    full = oracle.run(tgt, x0, 40, seed=99)
    # chains split over two "ranks"
    a = oracle.run(tgt, x0[:20], 40, seed=99, chain_offset=0)
    b = oracle.run(tgt, x0[20:], 40, seed=99, chain_offset=20)
    assert np.array_equal(np.concatenate([a["samples"], b["samples"]]), full["samples"])
    # steps split over two calls (resume)
    h1 = oracle.run(tgt, x0, 25, seed=99)
    h2 = oracle.run(tgt, h1["state"], 15, seed=99, step_offset=25)
    assert np.array_equal(h2["state"], full["state"])
    assert np.array_equal(h1["n_reject"] + h2["n_reject"], full["n_reject"])
    # OpenMP over chains changes nothing
    par = oracle.run(tgt, x0, 40, seed=99, n_threads=4)
    assert np.array_equal(par["samples"], full["samples"])


def test_oracle_statistics_match_reference(oracle):
    """Philox-driven oracle chains reproduce the reference's rejections/step and mode
    occupancy (stats_*.npz: 8 reference chains) within Monte-Carlo error."""
    for name, n_chains, n_steps in (("vmfmix_readme", 256, 400), ("bingham_d10_vmax30", 256, 400)):
        s = golden(f"stats_{name}.npz")
        t = golden(f"traj_{name}.npz")
        tgt = oracle.Target.from_fixture(t)
        x0 = np.repeat(t["x0"][None], n_chains, axis=0)
        out = oracle.run(tgt, x0, n_steps, seed=5, thin=1, n_threads=4)
        burn = 100
        rej = out["n_reject"].sum() / (n_chains * n_steps)
        ref = s["rej_per_step"].mean()
        assert abs(rej - ref) / ref < 0.03, (name, rej, ref)
        if "occupancy" in s.files:
            X = out["samples"][:, burn:].reshape(-1, 3)
            mu = t["target_mu"]
            modes = mu / np.linalg.norm(mu, axis=1, keepdims=True)
            occ = np.bincount(np.argmax(X @ modes.T, axis=1), minlength=len(mu)) / len(X)
            assert np.max(np.abs(occ - s["occupancy"].mean(0))) < 0.05, (occ, s["occupancy"].mean(0))


def test_numpy_port_reproduces_reference_chain():
    """oracle/numpy_port.py (the reference-like CPU baseline bench.py times) seeded like the
    reference's README run reproduces the golden chain exactly (same numpy stream, same arithmetic)."""
    from oracle import numpy_port
    t = golden("traj_vmfmix_readme.npz")
    out, rej = numpy_port.run_chain(numpy_port.VmfMixture(t["target_mu"], t["target_weights"]), t["x0"], 300, 3521)
    assert np.max(np.abs(out - t["states"][1:301])) < 1e-12
    assert rej == int((t["tries"][:300] - 1).sum())


@pytest.mark.parametrize("name", ["traj_cpd_protein", "traj_gmm_protein_k10", "traj_cpd_cube_3d2d"])
def test_registration_log_prob_kat(oracle, name):
    """CoherentPointDrift / GaussianMixtureModel.log_prob (geosss/registration.py) at 48 quaternions, unit and not: the
    oracle's brute-force k nearest neighbours against the reference's KDTree."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    got = tgt.log_prob(z["kat_q"])
    assert np.max(np.abs(got - z["kat_logp"]) / np.maximum(1.0, np.abs(z["kat_logp"]))) < TOL

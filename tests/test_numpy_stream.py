"""numpy's own random stream restated (oracle: PCG64 XSL-RR + Generator.random/uniform/
standard_normal with the ziggurat tables read out of the installed numpy) matches numpy bit for bit,
and with it the oracle reproduces every golden reference chain FROM ITS SEED ALONE."""
import numpy as np
import pytest

from conftest import golden, trajectory_names


@pytest.mark.parametrize("seed", [0, 3521, 2**63 + 12345, np.random.SeedSequence(48385).spawn(3)[2]])
def test_stream_matches_numpy_bitwise(oracle, seed):
    g = np.random.default_rng(seed)
    want_z = g.standard_normal(300000)      # ~2000 draws take the ziggurat's slow paths, ~80 its tail
    want_u = g.random(1000)
    z, u, words = oracle.npy_fill(oracle.pcg64_words(seed)[0], 300000, 1000)
    assert np.array_equal(z, want_z)
    assert np.array_equal(u, want_u)
    # the advanced state equals numpy's advanced state
    st = g.bit_generator.state["state"]
    assert (int(words[0]) << 64 | int(words[1])) == st["state"]
    # interleaving normals / uniforms as the sampler does
    g = np.random.default_rng(seed)
    w = oracle.pcg64_words(seed)[0]
    for _ in range(200):
        z, u, w = oracle.npy_fill(w, 3, 5)
        assert np.array_equal(z, g.standard_normal(3))
        assert np.array_equal(u, np.array([g.random() for _ in range(5)]))


@pytest.mark.parametrize("name", trajectory_names("shrink") + trajectory_names("reject"))
def test_reference_chain_from_seed(oracle, name):
    """cls(pdf, x0, seed) of the reference, reproduced from (pdf, x0, seed): all states at 1e-12."""
    z = golden(name + ".npz")
    tgt = oracle.Target.from_fixture(z)
    sampler = oracle.REJECT if str(z["sampler"]) == "reject" else oracle.SHRINK
    n_steps = len(z["states"]) - 1
    out = oracle.run(tgt, z["x0"], n_steps, sampler=sampler, numpy_seed=int(z["seed"]))
    assert out["err"][0] == 0
    assert np.max(np.abs(out["samples"][0] - z["states"][1:])) < 1e-12
    assert out["n_reject"][0] == int(z["n_reject"])

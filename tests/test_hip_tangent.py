"""The set-up of a step on S^2 as the bench times it, against the REFERENCE.

On the library's Philox stream a d = 3 step does not draw three normals and project them (mcmc.py:387,
sphere.py:29-33): it draws the unit tangent directly, u = cos(phi) b1 + sin(phi) b2 with phi = 2 pi w / 2^32
from one 32-bit word, in a fixed orthonormal basis of the tangent plane at x / |x| (gsss_device.h `tangent3`).
That is the same law iff (b1, b2, x / |x|) is orthonormal, phi is the angle of u in that basis, and the
reference's own tangents -- tests/golden/tangent_kat.npz, 10^5 pairs (x, spherical_projection(z, x)) written by
running the reference -- have uniformly distributed angles in the very same basis.  `gsss_tangent_s2` evaluates
what the sampler kernels evaluate (same functions, both sincos variants).
"""

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def _device_tangent(gs, x, w, table_driven):
    import torch
    lib = gs._lib.load()
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
    wd = torch.from_numpy(np.ascontiguousarray(w, dtype=np.uint32).view(np.int32)).cuda()
    out = torch.empty((len(x), 12), dtype=torch.float64, device="cuda")
    gs._lib.check(lib.gsss_tangent_s2(xd.data_ptr(), wd.data_ptr(), len(x), int(table_driven), out.data_ptr(), 0, None))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[:, 0:3], o[:, 3:6], o[:, 6:9], o[:, 9:12]


def _fixture_points():
    z = golden("tangent_kat.npz")
    n_pole = int(z["n_pole"])
    x = np.concatenate([z["x"], np.tile([0.0, 0.0, 1.0], (n_pole, 1)), np.tile([0.0, 0.0, -1.0], (n_pole, 1))])
    groups = {"unit": slice(0, int(z["n_unit"])), "short": slice(int(z["n_unit"]), int(z["n_unit"]) + int(z["n_short"])),
              "north pole": slice(len(z["x"]), len(z["x"]) + n_pole), "south pole": slice(len(z["x"]) + n_pole, None)}
    return x, z["u"], groups


@pytest.mark.parametrize("table_driven", [1, 0], ids=["throughput kernels", "exact kernels"])
def test_direct_tangent_is_the_reference_law(gs, table_driven):
    from scipy import stats
    x, u_ref, groups = _fixture_points()
    assert len(x) == 100_000 and abs(np.linalg.norm(x[groups["short"]][0]) - 0.998) < 1e-12
    rng = np.random.default_rng(7)
    w = rng.integers(0, 2**32, len(x), dtype=np.uint64).astype(np.uint32)
    w[:8] = [0, 1, 2**30, 2**31 - 1, 2**31, 2**31 + 1, 3 * 2**30, 2**32 - 1]  # the quadrant edges of the table-driven sincos
    nrm, b1, b2, u = _device_tangent(gs, x, w, table_driven)

    # (1) an orthonormal frame at every x -- unit, |x| = 0.998, both poles
    assert np.max(np.abs(nrm - x / np.linalg.norm(x, axis=1)[:, None])) < 1e-15
    for a, b in ((nrm, b1), (nrm, b2), (b1, b2)):
        assert np.max(np.abs(np.sum(a * b, axis=1))) < 1e-15
    for a in (nrm, b1, b2):
        assert np.max(np.abs(np.sum(a * a, axis=1) - 1.0)) < 1e-15

    # (2) the tangent for word w sits at angle exactly 2 pi w / 2^32 in that frame
    phi = 2.0 * np.pi * (w.astype(np.float64) / 2.0**32)
    want = np.cos(phi)[:, None] * b1 + np.sin(phi)[:, None] * b2
    assert np.max(np.abs(u - want)) < 1e-15
    ang = np.arctan2(np.sum(u * b2, axis=1), np.sum(u * b1, axis=1))
    diff = np.angle(np.exp(1j * (ang - phi)))
    assert np.max(np.abs(diff)) < 2e-15
    assert np.max(np.abs(np.sum(u * nrm, axis=1))) < 1e-15 and np.max(np.abs(np.sum(u * u, axis=1) - 1.0)) < 1e-15

    # (3) the reference's tangents u = spherical_projection(z, x) lie in that plane and their angles in the DEVICE's frame are
    # uniform: the one-angle draw and the projected normals are the same law (Kolmogorov-Smirnov per group of x and overall;
    # fixed data, so these are fixed numbers: p = 0.10 .. 0.72 with a numpy restatement of the frame)
    c1, c2 = np.sum(u_ref * b1, axis=1), np.sum(u_ref * b2, axis=1)
    assert np.max(np.abs(u_ref - c1[:, None] * b1 - c2[:, None] * b2)) < 1e-12
    v = np.arctan2(c2, c1) / (2.0 * np.pi) + 0.5
    for name, sl in list(groups.items()) + [("all", slice(None))]:
        ks = stats.kstest(v[sl], "uniform")
        assert ks.pvalue > 0.01, (name, ks)
    # ... and independent of where on the sphere x is: the angle does not correlate with the frame's own coordinates
    for j in range(3):
        r = np.corrcoef(np.cos(2.0 * np.pi * v[groups["unit"]]), x[groups["unit"], j])[0, 1]
        assert abs(r) < 4.0 / np.sqrt(60000), (j, r)


def test_words_of_the_stream_reach_the_tangent(gs, oracle):
    """The word the kernels feed to the tangent is word 3 of block 0 of the step: one step of the headline kernel from a
    known state moves along cos(theta) x + sin(theta) u with u the device tangent for that word (u is recovered from the
    move and checked against gsss_tangent_s2 at the stream's word, taken from the oracle's Philox restatement)."""
    orc = oracle
    z = golden("traj_vmfmix_readme.npz")
    from helpers import product_target
    pdf = product_target(z)
    n = 4096
    x0 = orc.sample_sphere(5, n, 3)
    seed, off, step = 991, 12345, 17
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, chain_offset=off, step_offset=step, placement="packed")
    s.advance(1)
    x1 = s.state
    # counter = (block, step_lo, chain_lo, chain_hi16 | step_hi16 << 16), key = seed
    w = np.array([orc.philox4x32_10([0, step, off + c, 0], [seed, 0])[3] for c in range(n)], dtype=np.uint32)
    _, _, _, u = _device_tangent(gs, x0, w, 1)
    # x1 = cos(theta) x0 + sin(theta) u  =>  the component of x1 orthogonal to x0 is parallel to u
    t = x1 - np.sum(x1 * x0, axis=1)[:, None] * x0
    sin_t = np.sum(t * u, axis=1)
    moved = np.abs(sin_t) > 1e-6
    assert moved.mean() > 0.99
    assert np.max(np.abs(t[moved] - sin_t[moved, None] * u[moved])) < 1e-13

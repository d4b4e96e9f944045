"""The reference's helper functions either side of the sampler (geosss/sphere.py coordinate maps and great-circle
helpers, the host-side densities of geosss/distributions.py, the clipped elementary functions of geosss/utils.py) against
values the reference itself produced (tests/golden/helpers_kat.npz, written by make_golden.py `helpers`)."""
import numpy as np
import pytest
import torch

from conftest import golden

TOL = dict(rtol=0, atol=1e-13)


def _sphere_checks(to, back):
    """`to`: numpy -> the container under test; `back`: its result -> numpy"""
    from geosss_amd import sphere as S
    z = golden("helpers_kat.npz")
    X3 = to(z["X3"])
    assert np.allclose(back(S.cartesian2polar(X3)), z["polar"], **TOL)
    phi, theta = S.cartesian2spherical(X3)
    assert np.allclose(back(phi), z["sph_phi"], **TOL) and np.allclose(back(theta), z["sph_theta"], **TOL)
    assert np.allclose(back(S.polar2cartesian(to(z["angles"]))), z["polar2cart"], **TOL)
    assert np.allclose(back(S.spherical2cartesian(to(z["angles"]), to(z["theta"]))), z["sph2cart"], **TOL)
    for d in (4, 10):
        v, u, x = to(z[f"d{d}_pole"]), to(z[f"d{d}_u"]), to(z[f"d{d}_x"])
        assert np.allclose(back(S.sample_subsphere(v, seed=3)), z[f"d{d}_subsphere"], **TOL)
        assert np.allclose(back(S.wrap(to(z[f"d{d}_tangent"]), u, v)), z[f"d{d}_wrap"], **TOL)
        arc, rot = S.slerp(v, x), S.givens(u, v, x)
        for i, p in enumerate(z[f"d{d}_phis"]):
            assert np.allclose(back(arc(float(p))), z[f"d{d}_slerp"][i], **TOL)
            assert np.allclose(back(rot(float(p))), z[f"d{d}_givens"][i], **TOL)
        stack = (torch.stack if isinstance(v, torch.Tensor) else np.stack)
        got = S.distance(stack([u, v, x]), stack([x, x, x]))
        # arccos at 1 turns the last-bit difference of two summation orders into ~1e-8 for the (x, x) pair
        assert np.allclose(back(got)[:2], z[f"d{d}_distance"][:2], **TOL) and abs(back(got)[2]) < 1e-7
    # the projections: batch and single point, non-unit poles (the fixtures of the geometry KAT)
    g = golden("geometry_kat.npz")
    for d in (3, 10, 50):
        X, Z = g[f"d{d}_x"], g[f"d{d}_z"]
        assert np.allclose(back(S.radial_projection(to(X))), g[f"d{d}_radial"], **TOL)
        assert np.allclose(back(S.radial_projection(to(X[0]))), g[f"d{d}_radial"][0], **TOL)
        for i in range(4):
            assert np.allclose(back(S.orthogonal_projection(to(Z[i]), to(X[i]))), g[f"d{d}_ortho"][i], **TOL)
            assert np.allclose(back(S.spherical_projection(to(Z[i]), to(X[i]))), g[f"d{d}_spherical"][i], **TOL)
        many = S.spherical_projection(to(Z), to(X[0]))                # a batch against one pole
        assert many.shape == Z.shape and np.allclose(back(many)[0], g[f"d{d}_spherical"][0], **TOL)
        assert np.max(np.abs(back(many) @ X[0])) < 1e-12


def test_sphere_helpers_numpy():
    from geosss_amd import sphere as S
    _sphere_checks(lambda a: np.array(a), lambda a: np.asarray(a))
    z = golden("helpers_kat.npz")
    for d in (4, 10):
        assert np.array_equal(S.sample_marginal(d, size=8, seed=5), z[f"d{d}_marginal"])
    assert isinstance(S.radial_projection(np.array([3.0, 4.0])), np.ndarray)


def test_sphere_helpers_torch_cpu():
    _sphere_checks(lambda a: torch.as_tensor(np.array(a)), lambda a: a.numpy())


@pytest.mark.gpu
def test_sphere_helpers_device_tensors():
    out = []
    _sphere_checks(lambda a: torch.as_tensor(np.array(a)).cuda(), lambda a: (out.append(a.is_cuda), a.cpu().numpy())[1])
    assert all(out)                                                  # results stayed on the device


def test_host_densities_against_reference_values():
    """MarginalVonMisesFisher (and the mixture of marginals the reference's diagnostics plot over the coordinate
    histograms, scripts/vMF_diagnostics.py:106-108), MultivariateNormal, ACG, Uniform."""
    import geosss_amd as gs
    z = golden("helpers_kat.npz")
    grid = z["grid"]
    for d in (3, 10):
        mus = z[f"marg_d{d}_mus"]
        for k, mu in enumerate(mus):
            for i in range(d):
                m = gs.MarginalVonMisesFisher(i, mu)
                assert np.allclose(m.prob(grid), z[f"marg_d{d}_prob"][k, i], rtol=1e-12, atol=0)
                assert np.allclose(m.log_prob(grid), z[f"marg_d{d}_log_prob"][k, i], rtol=1e-12, atol=1e-12)
        for i in range(d):
            mix = gs.MixtureModel([gs.MarginalVonMisesFisher(i, mu) for mu in mus])
            assert np.allclose(mix.log_prob(grid), z[f"marg_d{d}_mixture"][i], rtol=1e-12, atol=1e-12)
        # a density on [-1, 1]: integrates to one, and is no target for the samplers
        fine = np.linspace(-1, 1, 20001)[1:-1]
        assert abs(np.trapezoid(gs.MarginalVonMisesFisher(0, mus[0]).prob(fine), fine) - 1.0) < 1e-3
        with pytest.raises(TypeError):
            gs.MarginalVonMisesFisher(0, mus[0])._pack()
        with pytest.raises(TypeError):
            gs.MixtureModel([gs.MarginalVonMisesFisher(0, mu) for mu in mus])._pack()
        with pytest.raises(TypeError):
            gs.MixtureModel([gs.MarginalVonMisesFisher(0, mus[0]), gs.VonMisesFisher(mus[1])])
    mvn = gs.MultivariateNormal(z["mvn_mu"], z["mvn_C"])
    assert np.allclose(mvn.log_prob(z["mvn_Y"]), z["mvn_log_prob"], rtol=1e-12)
    acg = gs.ACG(z["mvn_C"])
    assert np.allclose(acg.log_prob(z["mvn_Y"]), z["acg_log_prob"], rtol=1e-12)
    assert np.allclose(acg.log_prob(z["mvn_Y"][0]), z["acg_log_prob"][0], rtol=1e-12)
    with pytest.raises(TypeError):
        acg._pack()
    u = gs.Uniform()
    assert u.log_prob(z["mvn_Y"]) == 0.0 and not np.any(u.gradient(z["mvn_Y"][0]))
    with pytest.raises(TypeError):
        u._pack()
    assert gs.Uniform(5).d == 5 and gs.Uniform(5)._pack()[0] == gs._lib.BINGHAM


def test_utils_helpers_against_reference_values():
    from geosss_amd import utils as U
    import geosss_amd as gs
    z = golden("helpers_kat.npz")
    assert np.array_equal(U.exp(z["clip_in"]), z["clip_exp"])
    assert np.array_equal(U.log(z["clip_log_in"]), z["clip_log"])
    assert abs(U.relative_entropy(z["kl_p"], z["kl_q"]) - float(z["kl"])) < 1e-12
    assert [U.format_time(t) for t in z["format_time_in"]] == list(z["format_time"])
    assert len(gs.colors) == 5 and all(len(c) == 3 for c in gs.colors)


@pytest.mark.gpu
def test_uniform_target_on_the_device():
    """Uniform(d) runs on the samplers as the zero Bingham target: every proposal of the shrinkage sampler is accepted at
    the first try, and the draws are uniform on the sphere (mean ~ 0, second moment ~ I / d)."""
    import geosss_amd as gs
    d, n = 6, 4096
    x0 = gs.sample_sphere(d - 1, n, seed=3)
    s = gs.ShrinkageSphericalSliceSampler(gs.Uniform(d), x0, seed=5)
    last = s.sample(40, as_tensor=True)[:, -1, :].cpu()
    assert s.n_reject == 0
    assert float(last.mean(0).abs().max()) < 5.0 / np.sqrt(n)
    assert float((last.T @ last / n - torch.eye(d, dtype=torch.float64) / d).abs().max()) < 5.0 / np.sqrt(n)


def test_curve_helpers_against_reference_values():
    """spherical_curve.py's host helpers: constrained_brownian_curve (same seed -> the reference's knots), SlerpCurve's
    arc-length tables and evaluation, distance_slerp and find_nearest (the geometry KAT's clipped-end and on-arc cases)."""
    from geosss_amd.spherical_curve import SlerpCurve, constrained_brownian_curve, distance_slerp
    z = golden("helpers_kat.npz")
    for d in (3, 7):
        knots = constrained_brownian_curve(n_points=12, dimension=d, step_size=0.3, seed=77 + d)
        assert np.allclose(knots, z[f"cbc_d{d}_knots"], rtol=0, atol=1e-13)
        curve = SlerpCurve(z[f"cbc_d{d}_knots"])
        for name in ("theta", "bins", "widths"):
            assert np.allclose(getattr(curve, name), z[f"cbc_d{d}_{name}"], rtol=0, atol=1e-13)
        assert np.allclose(curve(z[f"cbc_d{d}_t"]), z[f"cbc_d{d}_points"], rtol=0, atol=1e-12)
    g = golden("geometry_kat.npz")
    for d in (3, 10, 50):
        for q, a, b, dist, near in zip(g[f"d{d}_slerp_q"], g[f"d{d}_slerp_a"], g[f"d{d}_slerp_b"], g[f"d{d}_slerp_dist"],
                                       g[f"d{d}_slerp_near"]):
            got_d, got_y = distance_slerp(q, a, b)
            assert np.allclose(got_y, near, rtol=0, atol=1e-13) and abs(got_d - dist) < 1e-7
        curve = SlerpCurve(g[f"d{d}_knots"])
        for q, want in zip(g[f"d{d}_slerp_q"], g[f"d{d}_nearest"]):
            assert np.allclose(curve.find_nearest(q), want, rtol=0, atol=1e-13)


def test_result_files_round_trip_with_the_reference(tmp_path):
    """geosss_amd.io reads the pickles the reference's scripts write (tests/golden/ref_dump.pkl[.gz], written by
    geosss.io.dump) and writes the same format; tensors are stored as numpy arrays; the lock directory behaves as the
    reference's."""
    import os
    import pickle
    from conftest import GOLDEN as GOLDEN_DIR
    from geosss_amd import io
    z = golden("helpers_kat.npz")
    runs = io.load(os.path.join(GOLDEN_DIR, "ref_dump.pkl"))
    assert sorted(runs) == ["hmc", "rwmh", "sss-reject", "sss-shrink"]
    for m, v in runs.items():
        assert np.array_equal(v, z["dump_" + m])
    table = io.load(os.path.join(GOLDEN_DIR, "ref_dump.pkl.gz"), gzip=True)
    assert table["n"] == 7 and table["ess"][1:] == [3.5, "text"] and np.array_equal(table["ess"][0], z["dump_rwmh"][0])
    # what we write: byte-for-byte the reference's plain file for the same object, and readable compressed
    out = str(tmp_path / "runs.pkl")
    io.dump(runs, out)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN_DIR, "ref_dump.pkl"), "rb").read()
    io.dump({"x": torch.as_tensor(z["dump_hmc"]), "nested": [torch.ones(2), (torch.zeros(1), 4)]}, out + ".gz", gzip=True,
            lock=True)
    back = io.load(out + ".gz", gzip=True, lock=True, timeout=1.0)
    assert isinstance(back["x"], np.ndarray) and np.array_equal(back["x"], z["dump_hmc"])
    assert isinstance(back["nested"][0], np.ndarray) and isinstance(back["nested"][1][0], np.ndarray)
    assert not os.path.exists(out + ".gz.lock")
    with open(out + ".gz", "rb") as f:                         # plain pickle of numpy objects: no torch needed to read it
        assert b"torch" not in __import__("gzip").decompress(f.read())
    # a held lock: the second access times out with IOError; a truncated file is an IOError too
    os.mkdir(out + ".lock")
    with pytest.raises(IOError):
        io.load(out, lock=True, timeout=0.05)
    os.rmdir(out + ".lock")
    with open(out, "rb") as f:
        head = f.read(40)
    with open(out, "wb") as f:
        f.write(head)
    with pytest.raises(IOError):
        io.load(out)
    assert pickle.HIGHEST_PROTOCOL >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["shrink", "reject"])
def test_bingham_with_zeroed_matrix_is_uniform(cls):
    """The reference's tests/test_bingham.py:66-79 (`pdf = gs.random_bingham(d); pdf.A *= 0.0`, then both slice samplers):
    the edited target is re-uploaded, and every coordinate follows the marginal (1 - t^2)^((d-3)/2) of the uniform
    distribution -- held here through its second and fourth moments, 1/d and 3/(d (d + 2))."""
    import geosss_amd as gs
    d, n = 10, 20000
    pdf = gs.random_bingham(d)
    x0 = gs.sample_sphere(d - 1, n, seed=1)
    warm = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=2)
    warm.sample(2)                                            # the device copy of the non-zero matrix exists now
    pdf.A *= 0.0
    Sampler = gs.ShrinkageSphericalSliceSampler if cls == "shrink" else gs.RejectionSphericalSliceSampler
    s = Sampler(pdf, x0, seed=3)
    x = s.sample(30, as_tensor=True)[:, -1, :].cpu().numpy()
    assert s.n_reject == 0
    m2, m4 = (x ** 2).mean(0), (x ** 4).mean(0)
    assert np.max(np.abs(m2 - 1 / d)) < 5 * np.sqrt(2 / (d * d * (d + 2)) / n) + 1e-3
    assert np.max(np.abs(m4 - 3 / (d * (d + 2)))) < 2e-3


@pytest.mark.gpu
def test_demo_example_runs():
    """examples/demo.py -- the reference's demo.ipynb with the import swapped -- runs end to end and reports balanced mode
    occupancy for the many-chain slice samplers."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "demo.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l for l in r.stdout.splitlines() if "chain-steps/s" in l]
    assert len(rows) == 2
    for l in rows:
        occ = [float(v) for v in re.search(r"mode occupancy \[([^\]]*)\]", l).group(1).split()]
        assert max(abs(o - 1 / 3) for o in occ) < 0.01


def test_sphere_helper_identities():
    """Size-independent properties of the coordinate maps and great-circle helpers (random inputs, several dimensions)."""
    from geosss_amd import sphere as S
    rng = np.random.default_rng(99)
    X = S.sample_sphere(2, 500, seed=4)
    phi, theta = S.cartesian2spherical(X)
    assert np.allclose(S.spherical2cartesian(phi, theta), X, atol=1e-12)            # the maps invert each other on S^2
    ang = rng.uniform(0, 2 * np.pi, 200)
    assert np.allclose(S.cartesian2polar(S.polar2cartesian(ang)), ang, atol=1e-12)
    for d in (3, 7, 40):
        v = S.sample_sphere(d - 1, seed=d)
        u = S.sample_subsphere(v, seed=d + 1)
        x = S.sample_sphere(d - 1, seed=d + 2)
        assert abs(u @ v) < 1e-14 and abs(np.linalg.norm(u) - 1) < 1e-14             # a unit vector of the great subsphere
        rot, arc = S.givens(u, v, x), S.slerp(v, x)
        omega = np.arccos(v @ x)
        assert np.allclose(rot(0.0), x, atol=1e-15) and np.allclose(rot(2 * np.pi), x, atol=1e-13)
        assert np.allclose(arc(0.0), v, atol=1e-14) and np.allclose(arc(omega), x, atol=1e-13)
        for t in rng.uniform(0, 2 * np.pi, 5):
            y = rot(t)
            assert abs(np.linalg.norm(y) - 1) < 1e-13                                # rotations keep the sphere
            assert np.allclose(S.givens(u, v, y)(-t), x, atol=1e-13)                 # and compose
            z = arc(t * omega / (2 * np.pi))
            assert abs(np.linalg.norm(z) - 1) < 1e-13
            assert abs(S.distance(v, z) + S.distance(z, x) - omega) < 1e-7           # on the shortest arc between the two
        w = S.wrap(rng.standard_normal(d), u, v)
        assert abs(np.linalg.norm(w) - 1) < 1e-14
        P = S.orthogonal_projection(rng.standard_normal((20, d)), 3.7 * v)           # the pole need not be a unit vector
        assert np.max(np.abs(P @ v)) < 1e-13
    t = S.sample_marginal(10, size=200_000, seed=1)
    assert abs(np.mean(t)) < 5e-3 and abs(np.mean(t * t) - 0.1) < 2e-3               # a coordinate of the uniform law on S^9


@pytest.mark.gpu
def test_paper_experiments_example_runs():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "paper_experiments.py"), "--chains", "16", "--draws", "400"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("ESS(x_1)") == 16 and "KL on the spiral grid" in r.stdout


HMC_FIXTURES = ["mh_hmc_bingham_d10_vmax30", "mh_hmc_bingham_d5_dense", "mh_hmc_binghamfisher_d5", "mh_hmc_cpd_cube_3d2d",
                "mh_hmc_cpd_protein", "mh_hmc_curve_d10_kappa800", "mh_hmc_curve_d50_kappa800", "mh_hmc_gmm_protein_k10",
                "mh_hmc_vmfmix_d10_k5_kappa100", "mh_hmc_vmfmix_readme"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", HMC_FIXTURES)
def test_device_gradient_known_answers(name):
    """Distribution.gradient on the device (gsss_gradient: the functions the spherical HMC kernel evaluates) against the
    gradients the reference returned at the same points (`grad_X` -> `grad` of the HMC fixtures): one row, a batch, a device
    tensor.  BinghamFisher: the reference's class inherits Bingham's 2 A x -- its linear term does not enter."""
    from helpers import product_target
    z = golden(name + ".npz")
    X, want = z["grad_X"], z["grad"]
    assert len(X) > 0
    pdf = product_target(z)
    scale = np.maximum(1.0, np.abs(want))
    got = pdf.gradient(X)
    assert got.shape == want.shape and np.max(np.abs(got - want) / scale) < 1e-9
    assert np.max(np.abs(pdf.gradient(X[0]) - want[0]) / scale[0]) < 1e-9
    on_dev = pdf.gradient(torch.as_tensor(X).cuda())
    assert on_dev.is_cuda and np.array_equal(on_dev.cpu().numpy(), got)
    with pytest.raises(ValueError):
        pdf.gradient(np.zeros(pdf.d + 1))


@pytest.mark.gpu
def test_registration_with_translation_after_reference_test_cpd():
    """The reference's tests/test_cpd.py on the device: the cube, a z-y-z rotation, a translation, noise and outliers;
    log_prob(R, t) and gradient(q, t) equal the reference's values (helpers_kat.npz), the true pose scores above the identity,
    and the gradient agrees with finite differences of log_prob."""
    from scipy.optimize import approx_fprime
    from geosss_amd.pointcloud import PointCloud, RotationMatrix, RotationProjection, matrix2quat
    from geosss_amd.registration import CoherentPointDrift
    z = golden("helpers_kat.npz")
    cube = np.array([[x, y, zz] for zz in (-1, 1) for y in (-1, 1) for x in (-1, 1)], dtype=float)
    R_true = RotationMatrix().rotation3d(np.array([0.2, 0.3, 0.1]))
    assert np.allclose(R_true, z["cpdt_R"], atol=1e-15) and np.allclose(matrix2quat(R_true), z["cpdt_q"], atol=1e-15)
    assert np.allclose(RotationMatrix(degree=True).create(z["cpdt_euler_deg"]), z["cpdt_R_deg"], atol=1e-15)
    assert np.allclose(RotationMatrix().rotation2d(0.7), z["cpdt_R_2d"], atol=1e-15)
    qs = z["cpdt_qs"]
    for tag, src in (("3d", PointCloud(cube)), ("2d", RotationProjection(cube))):
        t_true = z[f"cpdt_{tag}_t"]
        for w in (0.0, 0.2, 0.4):
            key = f"cpdt_{tag}_w{int(10 * w)}"
            cpd = CoherentPointDrift(PointCloud(z[f"cpdt_{tag}_target"]), src, sigma=0.5, k=8, beta=1.0, omega=w)
            lp_true, lp_id = cpd.log_prob(R_true, t_true), cpd.log_prob(np.eye(3), np.zeros(len(t_true)))
            assert abs(lp_true - float(z[key + "_logp_true"])) < 1e-9 and abs(lp_id - float(z[key + "_logp_identity"])) < 1e-9
            assert lp_true > lp_id
            assert np.allclose(cpd.log_prob(qs, t_true), z[key + "_logp_qs"], rtol=0, atol=1e-9)
            assert np.allclose(cpd.gradient(qs, t_true), z[key + "_grad_qs"], rtol=1e-9, atol=1e-9)
            assert np.allclose(cpd.gradient(qs), z[key + "_grad_qs_not"], rtol=1e-9, atol=1e-9)
            q_true = matrix2quat(R_true)
            numeric = approx_fprime(q_true, lambda q: cpd.log_prob(q, t_true), epsilon=1e-6)
            assert np.max(np.abs(cpd.gradient(q_true, t_true) - numeric)) < 1e-3 * max(1.0, np.max(np.abs(numeric)))
            with pytest.raises(ValueError):
                cpd.log_prob(q_true, np.zeros(len(t_true) + 1))

"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every
symbol include/gsss.h declares; the product refuses to run without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gsss.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsss_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from geosss_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gsss.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.gsss_abi_version() == _lib.ABI_VERSION == 10


def test_struct_layouts_match_header():
    """Field order of the ctypes structs = field order in gsss.h."""
    from geosss_amd import _lib
    text = open(os.path.join(ROOT, "include", "gsss.h")).read()
    for cname, cls in (("gsss_target_desc", _lib.TargetDesc), ("gsss_run_args", _lib.RunArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = re.findall(r"\b\*?\s*([a-z_0-9A-Z]+)\s*;", body)
        assert fields == [f[0] for f in cls._fields_], cname


def test_integration_doc_structs_match_header():
    """INTEGRATION.md section B shows the ctypes binding a geosss maintainer would add: its two struct snippets list the
    header's fields, in the header's order, under the header's ABI version (round 2's drifted: ABI 7, a truncated _Desc)."""
    from geosss_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for cls_name, cls in (("_Desc", _lib.TargetDesc), ("_Run", _lib.RunArgs)):
        start = doc.index(f"class {cls_name}(C.Structure):")
        block = doc[start: doc.index("\n\n", start)]                      # the class statement up to the next blank line
        fields = re.findall(r'\("([a-z_0-9A-Z]+)", C\.(c_[a-z0-9_]+)\)', block)
        assert [f for f, _ in fields] == [f[0] for f in cls._fields_], cls_name
        import ctypes as C
        assert [getattr(C, t) for _, t in fields] == [f[1] for f in cls._fields_], cls_name   # (c_int32 IS c_int here)
    assert f"ABI version {_lib.ABI_VERSION}" in doc and f"(ABI {_lib.ABI_VERSION})" in doc
    assert not re.search(r"ABI version (?!%d)\d" % _lib.ABI_VERSION, doc)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import geosss_amd as gs
    from geosss_amd._lib import GsssError
    pdf = gs.MixtureModel([gs.VonMisesFisher([0.0, 0.0, 5.0])])
    with pytest.raises(GsssError):
        pdf.log_prob(np.array([0.0, 0.0, 1.0]))
    with pytest.raises(GsssError):
        gs.ShrinkageSphericalSliceSampler(pdf, np.array([0.0, 0.0, 1.0]), 1)
    with pytest.raises(GsssError):
        gs.sample_sphere(2, 4, seed=0, rng="philox")


def test_target_construction_matches_reference_recipes():
    """random_bingham / brownian_curve rebuild the paper's targets from the same seeds
    (golden fixtures hold the reference's matrices and knots)."""
    import geosss_amd as gs
    from conftest import golden
    z = golden("traj_bingham_d10_vmax30.npz")
    b = gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982)
    assert np.allclose(b.A, z["target_A"], rtol=0, atol=1e-12)
    assert np.allclose(np.abs(b.mode), np.abs(z["x0"]))
    z = golden("traj_bingham_d5_dense.npz")
    b = gs.random_bingham(d=5, vmax=20.0, vmin=-3.0, eigensystem=False, seed=11)
    assert np.allclose(b.A, z["target_A"], rtol=0, atol=1e-12)
    z = golden("traj_curve_d10_kappa800.npz")
    k = gs.brownian_curve(n_points=10, dimension=10, step_size=0.5, seed=4562)
    assert np.allclose(k, z["target_knots"], rtol=0, atol=1e-14)


def test_burnin_and_seed_rules():
    import geosss_amd as gs
    from geosss_amd.mcmc import seed_to_key
    assert gs.determine_burnin(1000, 100) == 100
    assert gs.determine_burnin(1000, 0.2) == 200
    with pytest.raises(AssertionError):
        gs.determine_burnin(10, -1)
    with pytest.raises(AssertionError):
        gs.determine_burnin(10, 1.5)
    assert seed_to_key(3521) == 3521
    ss = np.random.SeedSequence(48385).spawn(3)
    keys = {seed_to_key(s) for s in ss}
    assert len(keys) == 3 and all(0 <= k < 2**64 for k in keys)
    assert seed_to_key(None) != seed_to_key(None)


def test_call_counter_protocol():
    import geosss_amd as gs
    f = gs.MixtureModel.log_prob
    assert hasattr(f, "num_calls") and callable(f.reset_counters)
    f.reset_counters()
    assert gs.MixtureModel.log_prob.num_calls == 0
    pdf = gs.MixtureModel([gs.VonMisesFisher([0.0, 0.0, 5.0])])
    assert pdf.log_prob.num_calls == 0  # reachable through the bound method, as scripts do
    # separate counters per class, like the reference's per-class decoration
    assert gs.Bingham.log_prob is not gs.MixtureModel.log_prob


def test_sample_sphere_is_the_reference_recipe():
    """gs.sample_sphere(d, size, seed) = radial_projection(default_rng(seed).standard_normal(...)) as in
    geosss/sphere.py:39-50: the reference's target recipes build the reference's arrays (fixtures hold them)."""
    import geosss_amd as gs
    from conftest import golden
    z = golden("traj_vmfmix_k10_kappa500.npz")          # modes = sample_sphere(2, 10, seed=1234) * 500
    assert np.array_equal(500.0 * gs.sample_sphere(2, 10, seed=1234), z["target_mu"])
    assert np.array_equal(gs.sample_sphere(2, seed=1345), z["x0"])
    for name, d in (("traj_curve_d10_kappa800.npz", 10), ("traj_curve_d50_kappa800.npz", 50),
                    ("traj_curve_d200_kappa800.npz", 200)):
        assert np.array_equal(gs.sample_sphere(d - 1, seed=1345), golden(name)["x0"])   # scripts/curve_vMF.py:577-589
    x = gs.sample_sphere(4, 7, seed=3)
    g = np.random.default_rng(3).standard_normal((7, 5))
    assert np.array_equal(x, g / (np.linalg.norm(g, axis=-1) + 1e-100)[:, None])
    with pytest.raises(ValueError):
        gs.sample_sphere(2, 4, seed=0, rng="pcg")


def test_seed_argument_handling():
    """Seeds are never silently reinterpreted: a list of seeds is refused for the Philox stream; a Generator handed over as
    seed keeps its stream position, and every sampler built from it gets a key of its own (a spawned child of its seed
    sequence) -- two samplers from one generator are independent, as in the reference, which shares the generator (mcmc.py:45)."""
    import geosss_amd as gs
    from geosss_amd.mcmc import seed_to_key
    g = np.random.default_rng(5)
    before = g.bit_generator.state
    k1, k2 = seed_to_key(g), seed_to_key(g)
    assert k1 != k2 and g.bit_generator.state == before
    assert g.bit_generator.seed_seq.n_children_spawned == 0               # the caller's seed sequence is not spawned from either
    assert [c.spawn_key for c in g.bit_generator.seed_seq.spawn(2)] == [(0,), (1,)]  # ... so the caller's own children are what they would have been
    h = np.random.default_rng(5)
    assert (seed_to_key(h), seed_to_key(h)) == (k1, k2)                  # reproducible: same generator history, same keys
    assert seed_to_key(h.bit_generator) not in (k1, k2)                  # a BitGenerator spawns from the same sequence
    # the count of a LIVE generator never restarts, however many other generators pass by (the table used to evict
    # the generator seen longest ago at 1024 entries, alive or not, and hand its next sampler the key of its first)
    from geosss_amd import mcmc
    live = [np.random.default_rng(100 + i) for i in range(700)]
    for x in live:
        seed_to_key(x)
    for i in range(1500):
        seed_to_key(np.random.default_rng(10_000 + i))                   # dropped at once: forgotten when the table fills
    assert seed_to_key(g) not in (k1, k2)
    assert all(mcmc._children_built[id(x.bit_generator)][1] == 1 for x in live) and len(mcmc._children_built) < 1800
    pdf = gs.MixtureModel([gs.VonMisesFisher([0.0, 0.0, 5.0])])
    with pytest.raises(ValueError):
        gs.ShrinkageSphericalSliceSampler(pdf, np.eye(3), [1, 2, 3])
    with pytest.raises(ValueError):
        gs.ShrinkageSphericalSliceSampler(pdf, np.eye(3), 1, rng="mt19937")


def test_script_conveniences(caplog):
    """What the reference's scripts reach for next to the samplers: `gs.take_time` (utils.py:42-48; wall time around a
    synchronised block here) and `gs.sphere.distance` (sphere.py:64-68)."""
    import logging
    import geosss_amd as gs
    with caplog.at_level(logging.INFO):
        with gs.take_time("demo"):
            pass
        with gs.take_time("quiet", mute=True):
            pass
    assert any(r.getMessage().startswith("demo took ") for r in caplog.records)
    assert not any("quiet" in r.getMessage() for r in caplog.records)
    d = gs.sphere.distance(np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0]]))
    assert np.allclose(d, [np.pi / 2, 0.0])


def test_custom_python_distribution_is_refused_with_a_clear_message():
    """The reference's samplers take any object with a Python log_prob; the kernels cannot call Python: a subclass without a
    device parameter block is refused by name, before anything is launched."""
    import pytest
    import geosss_amd as gs

    class Mine(gs.Distribution):
        d = 3

        def log_prob(self, x):
            return 0.0

    with pytest.raises(TypeError, match="no device parameter block"):
        Mine()._pack()


def test_graft_entry_build_passes():
    """`__graft_entry__.build()` -- the driver's "does it build" check -- compiles, loads and checks the ABI it was built with."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=root, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-1500:]


def test_group_kernels_do_not_spill():
    """The curve group kernels the bench times are built for three (d <= 32) / two wavefronts per SIMD without scratch beyond a measured budget: their
    register budget rests on an internal LLVM option (geosss_amd/build.py probes it) and on the kernel's own structure, and a
    spilling build would send gigabytes of scratch traffic per launch to HBM (gsss_curvespec.h).  Checked from the compiler's
    own resource remarks, device code only (hipcc cross-compiles here)."""
    from geosss_amd import build
    ru = build.resource_usage("gsss_fast_curvespec.hip")
    want = {"ILi4ELi1ELi10ELb0ELb0ELi0E": 3, "ILi16ELi1ELi10ELb0ELb0ELi0E": 3, "ILi4ELi2ELi10ELb0ELb0ELi0E": 3, "ILi4ELi3ELi10ELb0ELb0ELi0E": 3,
            "ILi8ELi3ELi10ELb0ELb0ELi0E": 3, "ILi16ELi3ELi10ELb0ELb0ELi0E": 3,
            "ILi4ELi1ELi17ELb0ELb0ELi0E": 3, "ILi4ELi2ELi17ELb0ELb0ELi0E": 3, "ILi16ELi1ELi17ELb0ELb0ELi0E": 3, "ILi16ELi2ELi17ELb0ELb0ELi0E": 3,
            "ILi4ELi4ELi10ELb0ELb0ELi0E": 2, "ILi8ELi4ELi10ELb0ELb0ELi0E": 2, "ILi16ELi4ELi10ELb0ELb0ELi0E": 2,
            # round 5: three quads + one tail component per lane (d = 49 .. 52, 97 .. 104, 193 .. 208; cfg4's d = 50 / 200)
            "ILi4ELi3ELi10ELb0ELb0ELi1E": 3, "ILi8ELi3ELi10ELb0ELb0ELi1E": 3, "ILi16ELi3ELi10ELb0ELb0ELi1E": 3,
            # ... one behind one and two quads, and two tail components per lane (d = 17 .. 24 -- cfg4's overlap point d = 24 --, 33 .. 40, 53 .. 56, 105 .. 112)
            "ILi4ELi1ELi10ELb0ELb0ELi1E": 3, "ILi4ELi1ELi10ELb0ELb0ELi2E": 3, "ILi4ELi2ELi10ELb0ELb0ELi1E": 3, "ILi4ELi2ELi10ELb0ELb0ELi2E": 3,
            "ILi4ELi3ELi10ELb0ELb0ELi2E": 3, "ILi8ELi3ELi10ELb0ELb0ELi2E": 3}  # <L, Q, NK, replay, stats, tail> -> waves per SIMD
    # scratch a build may hold (bytes a lane).  Two and three component quads per lane are the MEASURED exceptions to "no
    # scratch": three wavefronts with 128 .. 224 bytes spilled are 16-22 % faster than two without (gsss_curvespec.h,
    # profiles/r04_ab_q2_three_waves.log; the resident wavefronts' scratch stays in or near the L2), which four quads are not
    # (d = 50: 320 bytes, 15 % slower): those hold nothing in scratch
    budget = {"ILi4ELi1ELi10ELb0ELb0ELi0E": 0, "ILi16ELi1ELi10ELb0ELb0ELi0E": 0, "ILi4ELi2ELi10ELb0ELb0ELi0E": 104, "ILi4ELi3ELi10ELb0ELb0ELi0E": 124,
              "ILi8ELi3ELi10ELb0ELb0ELi0E": 116, "ILi16ELi3ELi10ELb0ELb0ELi0E": 108,
              "ILi4ELi1ELi17ELb0ELb0ELi0E": 160, "ILi4ELi2ELi17ELb0ELb0ELi0E": 184, "ILi16ELi1ELi17ELb0ELb0ELi0E": 152, "ILi16ELi2ELi17ELb0ELb0ELi0E": 180,
              # (round 5: the try uniforms rest in LDS as 32-bit words -- every build lost 12 .. 48 B of scratch; the three-quad builds run
              # without the knot-row pipeline: 176 / 140 / 184 -> 124 / 116 / 108 B, the uneven ones 216 / 184 / 240 -> 176 / 168 / 160 B, so that
              # the resident wavefronts' scratch stays under an XCD's 4 MB of L2: profiles/r05_ab_knot_pipe_q3.log)
              "ILi4ELi3ELi10ELb0ELb0ELi1E": 176, "ILi8ELi3ELi10ELb0ELb0ELi1E": 168, "ILi16ELi3ELi10ELb0ELb0ELi1E": 160,
              "ILi4ELi1ELi10ELb0ELb0ELi1E": 56, "ILi4ELi1ELi10ELb0ELb0ELi2E": 64, "ILi4ELi2ELi10ELb0ELb0ELi1E": 160, "ILi4ELi2ELi10ELb0ELb0ELi2E": 188,
              "ILi4ELi3ELi10ELb0ELb0ELi2E": 212, "ILi8ELi3ELi10ELb0ELb0ELi2E": 196}
    seen = 0
    for name, r in ru.items():
        for key, waves in want.items():
            if "curvespec_kernel" + key in name:
                seen += 1
                assert r["scratch"] <= budget.get(key, 0), (name, r)
                assert r["occupancy"] >= waves, (name, r)
    assert seen == len(want), sorted(ru)


def test_wide_lane_kernels_do_not_spill():
    """The one-chain-per-lane screened kernels of d = 11 .. 16 (round 4) hold 4 d registers of state per lane: built for two or
    three wavefronts per SIMD without scratch (a spilling lane kernel is what rounds 2-3 measured 1.1-4 x slower)."""
    from geosss_amd import build
    seen = 0
    for src, dims in (("gsss_fast_vmf_d12.hip", (12,)), ("gsss_fast_vmf_d16.hip", (16,)), ("gsss_fast_bingham_wide_a.hip", (11, 12, 13)),
                      ("gsss_fast_bingham_wide_b.hip", (14, 15, 16))):
        for name, r in build.resource_usage(src).items():
            if "screened_kernel" in name and "Lb0ELb0ELb0ELb0E" in name:          # <.., replay, stats, stage, numpy: all false>
                seen += 1
                assert r["scratch"] == 0 and r["occupancy"] >= 2, (name, r)
    assert seen == 2 * 3 + 2 * 6

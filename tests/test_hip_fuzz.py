"""Randomised differential test of the lane kernels' launch machinery.

One chain per lane or two, the one-chain-per-lane BUILD (`screened_kernel<.., STAGE>`), rows held back in LDS until a run of them
ends on a sector, a sliced last round of workgroups, chain-major or component-major retained rows, row offsets, launches that end
between two kept rows, chains that stop at max_tries: each is a choice the library makes from the launch's shape, and none of them
may change a bit.  Every case draws a target, an ensemble size around the chip's resident workgroups, step counts, thinning and an
output layout at random (seeded), runs it as the library would and again with every one of those choices switched off
(GSSS_ONE_PER_LANE=0, GSSS_STAGE_ROWS=0, GSSS_SLICE_STEPS=0: two chains per lane, plain stores, one workgroup per chunk for the
whole launch), and compares states, retained rows, tries, rejections and error flags bitwise.  The plain configuration is the
one the oracle tests of test_hip_parity.py hold to the reference's chains.  (geosss/mcmc.py:55-77, 382-401: `sample` keeps every
thin-th state of every chain, whatever the launch shape.)
"""
import ctypes as C
import os

import numpy as np
import pytest

# GSSS_FUZZ_SCALE=k runs k times the cases (a soak run on the GPU box; the default suite stays at a few seconds)
SCALE = max(1, int(os.environ.get("GSSS_FUZZ_SCALE", "1")))

pytestmark = pytest.mark.gpu

OFF = {"GSSS_ONE_PER_LANE": "0", "GSSS_STAGE_ROWS": "0", "GSSS_SLICE_STEPS": "0"}
SEEN = {"cases": 0, "sliced": 0, "chain_major": 0, "bingham_chain_major_open_end": 0, "stopped_chains": 0}


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def _target(gs, rng):
    kind = rng.choice(["vmf", "bingham_diag", "bingham"])
    d = int(rng.integers(3, 11))
    if kind == "vmf":
        k = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 13, 16]))
        mu = float(rng.uniform(20.0, 90.0)) * gs.sample_sphere(d - 1, k, seed=int(rng.integers(1 << 30)))
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in mu], rng.uniform(0.5, 2.0, k)), d, f"vmf d={d} K={k}"
    pdf = gs.random_bingham(d=d, vmax=float(rng.uniform(10.0, 40.0)), vmin=0.0, eigensystem=kind == "bingham_diag", seed=int(rng.integers(1 << 30)))
    return pdf, d, f"{kind} d={d}"


def _run(gs, torch, pdf, d, x0, cfg, env, monkeypatch):
    for k in OFF:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cls = gs.RejectionSphericalSliceSampler if cfg["sampler"] == "reject" else gs.ShrinkageSphericalSliceSampler
    s = cls(pdf, x0, seed=cfg["seed"], mode="fast", placement="packed", step_offset=cfg["step_offset"], max_tries=cfg["max_tries"])
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()
    assert name.startswith("screened_kernel"), name
    n = x0.shape[0]
    outs, sliced = [], False
    for n_steps, thin, row0 in cfg["launches"]:
        rows = n_steps // thin
        if cfg["chain_major"]:
            buf = torch.zeros((n, rows + row0 + 1, d), dtype=torch.float64, device="cuda")
            s.advance(n_steps, thin=thin, out=buf, chain_major=True, row0=row0)
        else:
            buf = torch.zeros((rows, d, n), dtype=torch.float64, device="cuda")
            if rows > 0:
                s.advance(n_steps, thin=thin, out=buf)
            else:
                s.advance(n_steps)
        steps = C.c_int32(0)
        s._lib.gsss_last_launch(None, C.byref(steps), None)
        sliced = sliced or steps.value != 0
        outs.append(buf)
    torch.cuda.synchronize()
    return dict(rows=outs, state=s.state_device.clone(), tries=s._n_tries.clone(), rej=s._n_reject.clone(), err=s._err.clone(), sliced=sliced, kernel=name)


@pytest.mark.parametrize("case", range(96 * SCALE))
def test_launch_choices_do_not_change_a_bit(gs, case, monkeypatch):
    import torch
    rng = np.random.default_rng(9000 + case)
    pdf, d, what = _target(gs, rng)
    # ensembles: small (one chain per lane by the size rule), around one round of resident workgroups + a few (a sliced partial
    # round: 768 .. 1280 workgroups are resident, 256 or 512 chains each), ragged last chunks
    size = rng.choice(["small", "mid", "round"])
    n = {"small": int(rng.integers(300, 40_000)), "mid": int(rng.integers(100_000, 260_000)),
         "round": int(rng.choice([512, 768, 1024, 1280])) * int(rng.choice([256, 512])) + int(rng.integers(1, 60)) * 256 - int(rng.integers(0, 255))}[size]
    sampler = "reject" if rng.random() < 0.25 else "shrink"
    launches = []
    for _ in range(int(rng.integers(1, 4))):
        thin = int(rng.choice([1, 2, 3, 7, 10, 50, 100, 129]))
        n_steps = int(rng.integers(1, 5)) * 128 + int(rng.integers(0, 128)) if rng.random() < 0.6 else int(rng.integers(1, 200))
        thin = max(thin, -(-n_steps // 40))                              # (at most 40 rows a launch: the buffers of 6 x 10^5 chains stay small)
        launches.append((n_steps, thin, int(rng.integers(0, 3))))
    cfg = dict(sampler=sampler, seed=int(rng.integers(1 << 30)), step_offset=int(rng.integers(0, 1000)), launches=launches,
               chain_major=bool(rng.random() < 0.6), max_tries=(1 << 20) if sampler == "reject" or rng.random() < 0.6 else int(rng.integers(8, 30)))
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    got = _run(gs, torch, pdf, d, x0, cfg, {}, monkeypatch)
    ref = _run(gs, torch, pdf, d, x0, cfg, OFF, monkeypatch)
    info = (what, n, cfg, got["kernel"], "sliced" if got["sliced"] else "unsliced")
    assert not ref["sliced"], info
    for key in ("state", "tries", "rej", "err"):
        assert torch.equal(got[key], ref[key]), (key, info)
    for i, (a, b) in enumerate(zip(got["rows"], ref["rows"])):
        assert torch.equal(a, b), (f"rows of launch {i}", info)
    SEEN["cases"] += 1
    SEEN["sliced"] += int(got["sliced"])
    SEEN["chain_major"] += int(cfg["chain_major"])
    SEEN["bingham_chain_major_open_end"] += int(cfg["chain_major"] and what.startswith("bingham") and any(m % t for m, t, _ in launches))
    SEEN["stopped_chains"] += int(bool((ref["err"] != 0).any()))


def test_the_cases_covered_the_choices():
    """(runs behind the cases above) the draw of cases did reach what this file is about"""
    if SEEN["cases"] < 96 * SCALE:
        pytest.skip("not every case ran in this process")
    assert SEEN["sliced"] >= 10 and SEEN["chain_major"] >= 30 and SEEN["bingham_chain_major_open_end"] >= 10 and SEEN["stopped_chains"] >= 10, SEEN


CURVE_SEEN = {"cases": 0, "sliced": 0}


@pytest.mark.parametrize("case", range(40 * SCALE))
def test_group_kernel_slicing_does_not_change_a_bit(gs, case, monkeypatch):
    """The group-speculative curve kernels cut EVERY chunk of a launch into step slices once the chunks outnumber the resident
    workgroups: any dimension 4 .. 256 (lane groups of 4, 8, 16; one to four component quads a lane), 2 .. 17 knots, ragged
    ensembles, several launches (slices restart at the launch's global step), either layout of the retained rows -- against one
    workgroup per chunk (GSSS_SLICE_STEPS=0), bitwise."""
    import torch
    rng = np.random.default_rng(7000 + case)
    d = int(rng.choice([4, 5, 8, 10, 13, 16, 17, 24, 33, 49, 50, 52, 64, 65, 97, 100, 104, 128, 129, 193, 200, 208, 256]))   # (49 .. 52, 97 .. 104, 193 .. 208: the uneven layouts)
    k = int(rng.choice([2, 3, 7, 10, 10, 10, 12, 16 if d <= 64 else 17])) if d <= 48 or d > 64 else int(rng.choice([2, 5, 10, 10]))
    knots = gs.brownian_curve(k, d, 0.5, seed=int(rng.integers(1 << 30)))
    pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), float(rng.choice([100.0, 300.0, 800.0])))
    per = 64 if d <= 64 else (32 if d <= 128 and k <= 10 else 16)
    n = int(rng.integers(600, 1400)) * per - int(rng.integers(0, per)) if rng.random() < 0.7 else int(rng.integers(2000, 20_000))
    sampler = "reject" if rng.random() < 0.2 else "shrink"
    launches = [(int(rng.integers(2, 5)) * 128 + int(rng.integers(0, 128)), int(rng.choice([1, 5, 64, 100])), 0) for _ in range(int(rng.integers(1, 3)))]
    launches = [(m, max(t, -(-m // 12)), r) for m, t, r in launches]
    cfg = dict(sampler=sampler, seed=int(rng.integers(1 << 30)), step_offset=int(rng.integers(0, 1000)), launches=launches,
               chain_major=bool(rng.random() < 0.4), max_tries=(1 << 20) if sampler == "reject" or rng.random() < 0.7 else int(rng.integers(12, 40)))
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T

    def run(env):
        monkeypatch.delenv("GSSS_SLICE_STEPS", raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
        s = cls(pdf, x0, seed=cfg["seed"], mode="fast", placement="packed", step_offset=cfg["step_offset"], max_tries=cfg["max_tries"])
        name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()
        assert name.startswith("curvespec_kernel"), name
        outs, sliced = [], False
        for n_steps, thin, row0 in launches:
            rows = n_steps // thin
            if cfg["chain_major"]:
                buf = torch.zeros((n, rows + 1, d), dtype=torch.float64, device="cuda")
                s.advance(n_steps, thin=thin, out=buf, chain_major=True, row0=row0)
            else:                                             # (zeroed: a chain that stops keeps no further rows, and advance() hands out torch.empty)
                buf = torch.zeros((rows, d, n), dtype=torch.float64, device="cuda")
                s.advance(n_steps, thin=thin, out=buf)
            steps = C.c_int32(0)
            s._lib.gsss_last_launch(None, C.byref(steps), None)
            sliced = sliced or steps.value != 0
            outs.append(buf.clone())
        torch.cuda.synchronize()
        return outs, s.state_device.clone(), s._n_tries.clone(), s._n_reject.clone(), s._err.clone(), sliced, name

    got, ref = run({}), run({"GSSS_SLICE_STEPS": "0"})
    info = (d, k, n, cfg, got[6], "sliced" if got[5] else "unsliced")
    assert not ref[5], info
    for i in range(1, 5):
        assert torch.equal(got[i], ref[i]), (i, info)
    for a, b in zip(got[0], ref[0]):
        assert torch.equal(a, b), ("rows", info)
    CURVE_SEEN["cases"] += 1
    CURVE_SEEN["sliced"] += int(got[5])


def test_the_curve_cases_were_sliced():
    if CURVE_SEEN["cases"] < 40 * SCALE:
        pytest.skip("not every case ran in this process")
    assert CURVE_SEEN["sliced"] >= 12, CURVE_SEEN


def _any_target(gs, rng, dims):
    kind = rng.choice(["vmf", "bingham_diag", "bingham"])
    d = int(rng.choice(dims))
    if kind == "vmf":
        k = int(rng.choice([1, 2, 3, 5, 6, 9, 10]))
        mu = float(rng.uniform(20.0, 90.0)) * gs.sample_sphere(d - 1, k, seed=int(rng.integers(1 << 30)))
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in mu], rng.uniform(0.5, 2.0, k)), d, f"vmf d={d} K={k}"
    pdf = gs.random_bingham(d=d, vmax=float(rng.uniform(10.0, 40.0)), vmin=0.0, eigensystem=kind == "bingham_diag", seed=int(rng.integers(1 << 30)))
    return pdf, d, f"{kind} d={d}"


@pytest.mark.parametrize("case", range(40 * SCALE))
def test_screen_takes_only_the_decisions_it_can_guarantee(gs, case):
    """Random targets of every lane-kernel shape (d = 3 .. 16; mixtures of up to 10 components, eigenbasis and dense Bingham), both
    samplers, the library stream and numpy's: the single-precision screen must not change a decision.  d <= 10: the screened
    kernel against the all-double lane kernel, every state bit; d = 11 .. 16: against itself with every try left to the
    double-precision decision (GSSS_VARIANT_FAST_VERIFY) bit for bit, and against the all-double cooperative kernel in every
    integer output.  (geosss/mcmc.py:389, 397: the acceptance test is the reference's double-precision comparison.)"""
    import torch
    rng = np.random.default_rng(5000 + case)
    numpy_stream = case % 4 == 3
    pdf, d, what = _any_target(gs, rng, range(3, 11) if numpy_stream else range(3, 17))
    n = int(rng.integers(2_000, 60_000))
    n_steps = int(rng.integers(10, 60))
    sampler = "reject" if rng.random() < 0.25 else "shrink"
    if sampler == "reject":
        n, n_steps = max(500, n // 8), max(4, n_steps // 3)
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    seed = int(rng.integers(1 << 30))
    split = int(rng.integers(1, n_steps))
    out = {}
    for screen in (True, False) + (("verify",) if d > 10 else ()):
        kw = dict(rng="numpy") if numpy_stream else {}
        s = cls(pdf, x0, np.random.SeedSequence(seed) if numpy_stream else seed, mode="fast", placement="packed", screen=screen, **kw)
        a = s.advance(split, thin=1)
        b = s.advance(n_steps - split, thin=1)
        assert int((s._err != 0).sum().item()) == 0
        out[screen] = (torch.cat([a, b]), s.state_device.clone(), s._n_tries.clone(), s._n_reject.clone()) + ((s._rng_state.clone(),) if numpy_stream else ())
    info = (what, n, n_steps, sampler, "numpy" if numpy_stream else "philox")
    for i in (2, 3) + ((4,) if numpy_stream else ()):
        assert torch.equal(out[True][i], out[False][i]), (i, info)
    if d <= 10:
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1]), info
    else:
        assert float((out[True][0] - out[False][0]).abs().max().item()) < 1e-11, info
        for i in range(4):
            assert torch.equal(out[True][i], out["verify"][i]), (i, info)


@pytest.mark.parametrize("case", range(24 * SCALE))
def test_splitting_a_run_over_launches_does_not_change_a_bit(gs, case):
    """A chain's draws are keyed by (seed, chain, global step): n steps in one launch or in any split give the same bits -- lane
    kernels (d <= 16) and the four- / eight-lane group kernels, which form every coefficient from x at every step; chain blocks
    launched separately with their chain_offset (a shard of an ensemble, ensemble.py) included."""
    import torch
    rng = np.random.default_rng(3000 + case)
    if case % 3 == 2:
        d = int(rng.choice([4, 7, 10, 16, 24, 40, 50, 64, 100, 128]))
        knots = gs.brownian_curve(int(rng.choice([3, 10, 10])), d, 0.5, seed=int(rng.integers(1 << 30)))
        pdf, what = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), float(rng.choice([300.0, 800.0]))), f"curve d={d}"
    else:
        pdf, d, what = _any_target(gs, rng, range(3, 17))
    n = int(rng.integers(3_000, 50_000))
    n_steps = int(rng.integers(20, 300))
    thin = int(rng.choice([1, 3, 10]))
    n_steps -= n_steps % thin
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    seed = int(rng.integers(1 << 30))
    whole = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, mode="fast", placement="packed")
    rows = whole.advance(n_steps, thin=thin)
    cuts = sorted(set(int(c) * thin for c in rng.integers(1, max(2, n_steps // thin), size=3)))
    parts = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, mode="fast", placement="packed")
    got, done = [], 0
    for c in cuts + [n_steps]:
        if c > done:
            got.append(parts.advance(c - done, thin=thin))
            done = c
    info = (what, n, n_steps, thin, cuts)
    assert torch.equal(torch.cat(got), rows) and torch.equal(parts.state_device, whole.state_device) and torch.equal(parts._n_tries, whole._n_tries), info
    # the second half of the chains as a shard of its own
    lo = n // 2 + int(rng.integers(0, 100))
    shard = gs.ShrinkageSphericalSliceSampler(pdf, x0[lo:], seed=seed, mode="fast", placement="packed", chain_offset=lo)
    srows = shard.advance(n_steps, thin=thin)
    assert torch.equal(srows, rows[:, :, lo:]) and torch.equal(shard._n_tries, whole._n_tries[lo:]), info


@pytest.mark.parametrize("case", range(24 * SCALE))
def test_running_statistics_do_not_depend_on_the_launch_choices(gs, case, monkeypatch):
    """The running statistics (gsss_run_args.stats_dev: moments, lag sums, hops, mode counts; utils.py:96-134 on a series that is
    never stored) are sums over a chain's kept states in time order: packing, slicing and the split over launches must leave every
    accumulator bit as it is -- lane kernels and curve group kernels, with and without stored rows beside them."""
    import torch
    rng = np.random.default_rng(1000 + case)
    curve = case % 3 == 2
    if curve:
        d = int(rng.choice([5, 10, 24, 50, 100, 200]))
        knots = gs.brownian_curve(10, d, 0.5, seed=int(rng.integers(1 << 30)))
        pdf, what = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), 800.0), f"curve d={d}"
        per = 64 if d <= 64 else (32 if d <= 128 else 16)
        n = int(rng.integers(780, 1100)) * per - int(rng.integers(0, per))
    else:
        pdf, d, what = _any_target(gs, rng, range(3, 11))
        n = int(rng.choice([768, 1024, 1280])) * int(rng.choice([256, 512])) + int(rng.integers(1, 50)) * 256 - int(rng.integers(0, 255))
    thin = int(rng.choice([1, 4, 25]))
    launches = [int(rng.integers(2, 5)) * 128 + int(rng.integers(0, 128)) for _ in range(2)]
    keep = bool(rng.random() < 0.5)
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    seed = int(rng.integers(1 << 30))
    out = {}
    for label, env in (("default", {}), ("plain", OFF)):
        for k in OFF:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, mode="fast", placement="packed").enable_stats(lags=8)
        sliced = False
        for m in launches:
            m -= m % thin
            if keep:
                s.advance(m, thin=max(thin, -(-m // 8)))               # (stored rows beside the statistics, a few of them)
            else:
                s.advance(m, thin=thin, keep=False)
            steps = C.c_int32(0)
            s._lib.gsss_last_launch(None, C.byref(steps), None)
            sliced = sliced or steps.value != 0
        out[label] = (s._stats["acc"].clone(), s.state_device.clone(), sliced)
    info = (what, n, thin, launches, keep, "sliced" if out["default"][2] else "unsliced")
    assert not out["plain"][2], info
    assert torch.equal(out["default"][1], out["plain"][1]), info
    assert torch.equal(out["default"][0], out["plain"][0]), info
    assert float(out["default"][0].abs().sum().item()) > 0.0


@pytest.mark.parametrize("case", range(24 * SCALE))
def test_baseline_samplers_split_and_shard_invariance(gs, case):
    """RWMH / HMC / independence / mixture kernels (geosss/mcmc.py:118-332) on the counter-based stream: a run split over launches
    at random, and the upper part of the ensemble run as a shard of its own (chain_offset), give the bits of the whole -- states,
    accept counts, adapted stepsizes (the adaptation window `reset(burn)` counts a chain's own proposals)."""
    import torch
    rng = np.random.default_rng(2000 + case)
    if case % 4 == 3:
        d = int(rng.choice([5, 10, 24, 50]))
        knots = gs.brownian_curve(10, d, 0.5, seed=int(rng.integers(1 << 30)))
        pdf, what = gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), 300.0), f"curve d={d}"
    else:
        pdf, d, what = _any_target(gs, rng, [3, 4, 5, 8, 10, 12, 20])
    kind = ["rwmh", "hmc", "indep", "mix"][case % 4 if case < 16 else int(rng.integers(0, 4))]
    n = int(rng.integers(200, 5000))
    n_steps, burn = int(rng.integers(10, 60)), int(rng.integers(0, 20))
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    seed = int(rng.integers(1 << 30))

    def build(x, **kw):
        if kind == "rwmh":
            return gs.MetropolisHastings(pdf, x, seed, stepsize=0.3, **kw)
        if kind == "hmc":
            return gs.SphericalHMC(pdf, x, seed, stepsize=0.02, n_steps=int(5), **kw)
        if kind == "indep":
            return gs.IndependenceSampler(pdf, x, seed, **kw)
        return gs.MixtureRWMHIndependenceSampler(pdf, x, seed, stepsize=0.3, mixing_probability=0.3, **kw)

    def grab(s):
        return (s.state_device.clone(), s._n_accept.clone(), s._stepsize.clone())

    whole = build(x0)
    whole.reset(burn)
    whole.advance(n_steps)
    parts = build(x0)
    parts.reset(burn)
    done = 0
    for c in sorted(set(int(v) for v in rng.integers(1, n_steps, size=3))) + [n_steps]:
        parts.advance(c - done)
        done = c
    lo = n // 2 + int(rng.integers(0, 50))
    shard = build(x0[lo:], chain_offset=lo)
    shard.reset(burn)
    shard.advance(n_steps)
    info = (what, kind, n, n_steps, burn)
    for a, b in zip(grab(whole), grab(parts)):
        assert torch.equal(a, b), info
    w, sh = grab(whole), grab(shard)
    assert torch.equal(w[0][:, lo:], sh[0]) and torch.equal(w[1][lo:], sh[1]) and torch.equal(w[2][lo:], sh[2]), info
    assert 0 < int(w[1].sum().item()) <= n * n_steps, info


@pytest.mark.parametrize("case", range(20 * SCALE))
def test_sample_api_against_plain_launches(gs, case, monkeypatch):
    """`Sampler.sample(n_samples, burnin, thin)` (geosss/mcmc.py:55-77: the state after every thin-th of burnin + n_samples thin...
    steps, burn-in dropped) through the library's own row handling -- rows written in place in (chains, draws, dims) order, launches
    capped at 4096 steps, rows held back, sliced rounds -- equals the rows picked by hand from unit launches of the plain
    configuration, for random n_samples / burnin / thin and repeated calls."""
    import torch
    rng = np.random.default_rng(4000 + case)
    pdf, d, what = _any_target(gs, rng, range(3, 11))
    n = int(rng.choice([1, 7, 300, 5000, 20_000]))
    n_samples, thin = int(rng.integers(1, 40)), int(rng.choice([1, 1, 2, 5, 13]))
    burnin = int(rng.integers(0, 30)) if rng.random() < 0.7 else float(rng.choice([0.1, 0.25]))
    x0 = gs.sample_sphere_device(d - 1, n, seed=int(rng.integers(1 << 30))).T
    seed = int(rng.integers(1 << 30))
    for k in OFF:
        monkeypatch.delenv(k, raising=False)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, placement="packed", mode="fast")
    first = s.sample(n_samples, burnin=burnin, thin=thin, as_tensor=True)
    second = s.sample(5, thin=thin, as_tensor=True)
    for k, v in OFF.items():
        monkeypatch.setenv(k, v)
    r = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=seed, placement="packed", mode="fast")
    n_burn = gs.mcmc.determine_burnin(n_samples, burnin)
    info = (what, n, n_samples, burnin, thin)

    def by_hand(n_rows, skip):
        if skip:
            r.advance(skip)
        rows = [r.state_device.T.clone()[:, None, :]]                           # row 0: the state the call starts from (mcmc.py:64)
        if n_rows > 1:
            rows.append(r.advance((n_rows - 1) * thin, thin=thin).permute(2, 0, 1))
        return torch.cat(rows, dim=1)

    assert torch.equal(first.reshape(n, n_samples, d), by_hand(n_samples, n_burn)), info
    assert torch.equal(second.reshape(n, 5, d), by_hand(5, 0)), info


@pytest.fixture(scope="module")
def oracle_mod():
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import oracle
    oracle.build()
    return oracle


def _edge_target(gs, oracle, rng, d):
    """Targets at the edges of their families: the reference accepts them all (distributions.py:84-86, 156-157, 209-227, 272-275)"""
    kind = rng.choice(["vmf_flat", "vmf_sharp", "vmf_twins", "vmf_zero_weight", "bingham_flat", "bingham_repeated", "bingham_sharp",
                       "curve_two_knots", "curve_short_segments", "curve_sharp"])
    seed = int(rng.integers(1 << 30))
    if kind.startswith("vmf"):
        k = int(rng.choice([1, 2, 3, 6, 10]))
        dirs = gs.sample_sphere(d - 1, k, seed=seed)
        # (kappa < 714: the reference's log(i0(kappa)) overflows beyond, distributions.py:157 -- the oracle follows it, the device does not)
        kappa = {"vmf_flat": rng.uniform(0.0, 1e-3, k), "vmf_sharp": rng.uniform(300.0, 700.0, k)}.get(kind, rng.uniform(10.0, 100.0, k))
        if kind == "vmf_twins" and k >= 2:
            dirs[1] = dirs[0]                                                  # two components on the same mean
        w = rng.uniform(0.5, 2.0, k)
        if kind == "vmf_zero_weight" and k >= 2:
            w[int(rng.integers(0, k))] = 0.0
        mu = kappa[:, None] * dirs
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in mu], w), oracle.Target.vmf_mixture(mu, w), f"{kind} d={d} K={k}"
    if kind.startswith("bingham"):
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        lam = {"bingham_flat": np.zeros(d), "bingham_repeated": np.repeat(rng.uniform(0.0, 30.0, (d + 1) // 2), 2)[:d],
               "bingham_sharp": np.sort(rng.uniform(0.0, 400.0, d))}[kind]
        if rng.random() < 0.5:
            q = np.eye(d)                                                      # the eigenbasis kernels
        A = (q * lam) @ q.T
        A = 0.5 * (A + A.T)
        return gs.Bingham(A), oracle.Target.bingham(A), f"{kind} d={d} {'diag' if q[0, 0] == 1.0 and np.count_nonzero(q) == d else 'dense'}"
    if kind == "curve_two_knots":
        knots = gs.brownian_curve(2, d, 0.5, seed=seed)
    elif kind == "curve_short_segments":
        knots = gs.brownian_curve(10, d, 0.02, seed=seed)                      # sin(theta_g) ~ 0.02: the screen's margin grows with 1 / sin
    else:
        knots = gs.brownian_curve(int(rng.choice([5, 10])), d, 0.5, seed=seed)
    kappa = 5000.0 if kind == "curve_sharp" else float(rng.choice([50.0, 300.0]))
    return gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), kappa), oracle.Target.curve_vmf(knots, kappa), f"{kind} d={d} kappa={kappa}"


@pytest.mark.parametrize("case", range(48 * SCALE))
def test_edge_targets_match_oracle(gs, oracle_mod, case):
    """Targets at the edges of their families -- flat and very sharp mixtures, coincident components, a zero weight, Bingham with
    all-equal / repeated / widely spread eigenvalues, curves of two knots, of very short segments, at kappa = 5000 -- in every
    dimension range, both samplers: packed throughput kernels against the CPU oracle on the same Philox stream, states at 1e-10,
    tries exact, no error flag; where a screened kernel runs, its decisions are the all-double ones."""
    import torch
    oracle = oracle_mod
    rng = np.random.default_rng(6000 + case)
    d = int(rng.choice([3, 3, 4, 6, 8, 10, 12, 16, 24, 50, 130, 300, 700]))
    pdf, tgt, what = _edge_target(gs, oracle, rng, d)
    n, n_steps = (1500, 60) if d <= 16 else (300, 30)
    sampler = "reject" if rng.random() < 0.2 and "sharp" not in what else "shrink"
    if sampler == "reject":
        n_steps = 8
    cls = gs.RejectionSphericalSliceSampler if sampler == "reject" else gs.ShrinkageSphericalSliceSampler
    kind = oracle.REJECT if sampler == "reject" else oracle.SHRINK
    x0 = oracle.sample_sphere(int(rng.integers(1 << 20)), n, d)
    seed = int(rng.integers(1 << 30))
    want = oracle.run(tgt, x0, n_steps, seed=seed, sampler=kind, n_threads=16)
    s = cls(pdf, x0, seed=seed, placement="packed")
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode() if s.mode == "fast" else "exact"
    kept = s.advance(n_steps, thin=1).permute(2, 0, 1).cpu().numpy()
    info = (what, sampler, s.mode, name)
    assert np.all(want["err"] == 0) and np.all(s.errors == 0), info
    assert np.array_equal(s.n_tries_per_chain, want["n_tries"]), info
    assert np.max(np.abs(kept - want["samples"])) < 1e-10, info


def test_concurrent_host_threads_and_streams(gs):
    """Four host threads, each with its own HIP stream and samplers of its own (lane kernel with a sliced last round, group kernel
    with every chunk sliced, exact kernel, RWMH), launching at the same time: the library's host side keeps per-thread error text
    and launch records and one mutex-guarded cache of occupancy queries -- every thread gets the bits of its serial run."""
    import threading
    import torch

    def work(i, out):
        rng = np.random.default_rng(800 + i)
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            res = []
            for rep in range(3):
                pdf, d, _ = _any_target(gs, rng, [3, 5, 10])
                n = 1280 * 256 + 37 * 256 - 11 * i
                x0 = gs.sample_sphere_device(d - 1, n, seed=10 * i + rep).T
                s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=i, mode="fast", placement="packed")
                rows = s.advance(300, thin=100)
                knots = gs.brownian_curve(10, 10 + 7 * i, 0.5, seed=i)
                c = gs.ShrinkageSphericalSliceSampler(gs.CurvedVonMisesFisher(gs.SlerpCurve(knots), 300.0),
                                                     gs.sample_sphere_device(9 + 7 * i, 60_000, seed=i).T, seed=i, mode="fast", placement="packed")
                c.advance(260)
                e = gs.ShrinkageSphericalSliceSampler(pdf, x0[:2000], seed=i, mode="exact")
                e.advance(20)
                m = gs.MetropolisHastings(pdf, x0[:3000], i, stepsize=0.3)
                m.advance(30)
                torch.cuda.current_stream().synchronize()
                res.append([t.clone() for t in (rows, s.state_device, s._n_tries, c.state_device, c._n_tries, e.state_device, m.state_device, m._n_accept)])
            out[i] = res

    serial, threaded = {}, {}
    for i in range(4):
        work(i, serial)
    threads = [threading.Thread(target=work, args=(i, threaded)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(threaded) == [0, 1, 2, 3]
    for i in range(4):
        for a, b in zip(serial[i], threaded[i]):
            for x, y in zip(a, b):
                assert torch.equal(x, y), i

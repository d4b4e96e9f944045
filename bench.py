#!/usr/bin/env python3
"""Benchmark of the many-chain geodesic shrinkage slice sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W

`--gpus N > 1` without WORLD_SIZE in the environment launches N fresh ranks itself (one process per GPU
under `python -m torch.distributed.run`, the analogue of the reference's own fan-out,
scripts/curve_vMF.py:205-267); the parent process never touches the GPU and only relays rank 0's line
and the exit code.  Under an external launcher (RANK / LOCAL_RANK / WORLD_SIZE set) it is a rank.

Workload (BASELINE.json configs[1]): the README 3-component vMF mixture on
S^2 (kappa = 80), 10^6 independent chains PER GPU (weak scaling; chains of rank r have ids r*10^6 ...),
shrinkage sampler.  One bench "step" = one launch of the sampler kernel advancing every chain by
`--inner` (default 1000) MCMC transitions, keeping every 100th state (cfg2's thinned store); chain
states are resident in HBM before the timed region starts.  `value` = MCMC chain-steps per second over
all GPUs.  After the headline timing, rank 0 of a single-GPU run also times <= ~1 s of each other
BASELINE config (Bingham d=10, curve-vMF d=10/50/200, vMF mixture K=10 kappa=500) -> "configs".
Rank 0 prints ONE compact JSON line (< 6 KB: numbers and short names; `compact_line`) as the LAST line of stdout and writes the
complete record -- every field with its prose, per-config rooflines, ESS windows -- to bench_full.json (path in the line's
`full_record`; DESIGN.md "Measurement" explains every field).  On a multi-GPU run every rank also times cfg5's own workload
(vMF mixture K = 10 kappa = 500, 10^6 chains per GPU, chain ids rank * 10^6 ..., final gather) -> configs[{name: "..._sharded"}].
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # vendor vector-FP64 figure
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # PMC HBM bytes per launch, written by tools/pmc_traffic.py

README_MUS = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])

# (workload, chains per GPU): the BASELINE configs other than the headline one; cfg4 as the survey
# wrote it: kappa = 800 at d = 10 / 50 / 200 (sh/submit_job_curve_varying_ndim.sh:5), kappa = 500 at d = 10 (scripts/curve_vMF.py:495
# default) and d = 24, the overlap with the reference's own sweep (sh/submit_job_curve_varying_ndim.sh:11)
EXTRA_CONFIGS = [("bingham_d10", 1_000_000), ("curve_d10", 100_000), ("curve_d50", 100_000), ("curve_d200", 100_000),
                 ("curve_d10_kappa500", 100_000), ("curve_d24", 100_000), ("vmfmix_k10_kappa500", 1_000_000)]
# ... and the headline target on NUMPY'S OWN STREAM (rng="numpy": PCG64 + ziggurat per chain, the reference's arithmetic and
# draw order, mcmc.py:382-401 from the seed) -- the path that reproduces the reference's chains from (pdf, x0, seed) at 1e-10
NUMPY_STREAM_CONFIG = ("vmfmix_readme", 1_000_000)
# ... and, on a multi-GPU run, cfg5's own workload on every rank (10^6 chains per GPU, chain_offset = rank * 10^6, final gather)
SHARDED_CONFIG = ("vmfmix_k10_kappa500", 1_000_000)


def make_target(gs, name):
    if name == "vmfmix_readme":
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in README_MUS]), 3
    if name == "vmfmix_k10_kappa500":  # cfg5: the reference's recipe, scripts/mixture_vMF.py:406-411
        modes = gs.sample_sphere(2, 10, seed=1234, rng="numpy")
        return gs.MixtureModel([gs.VonMisesFisher(500.0 * m) for m in modes]), 3
    if name == "bingham_d10":
        return gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 10
    if name == "bingham_d50":  # scripts/bingham.py ind=1
        return gs.random_bingham(d=50, vmax=300.0, vmin=0.0, eigensystem=True, seed=6982), 50
    if name == "bingham_d50_dense":
        return gs.random_bingham(d=50, vmax=300.0, vmin=0.0, eigensystem=False, seed=6982), 50
    if name.startswith("curve_d"):  # curve_d<d>[_kappa<k>]
        parts = name[len("curve_d"):].split("_kappa")
        d, kappa = int(parts[0]), float(parts[1]) if len(parts) > 1 else 800.0
        return gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, d, 0.5, seed=4562)), kappa), d
    raise ValueError(name)


def algorithmic_flops(name, d, tries_per_step, rng="philox"):
    """FP64 flops (FMA = 2) of the restricted-form algorithm per chain-step, counted from the kernels'
    arithmetic (DESIGN.md "Roofline"): per step the d normals (one Box-Muller pair = 80: log 40, sincos 36,
    sqrt + scalings), the projection (3 dots, 2 axpys, 2 rsqrt-scalings = 10 d + 20), the coefficients of the
    great circle, the level of x and log U; per try sincos (36) + the level of y(theta) + bracket update.
    On S^2 (d = 3) the library stream draws the unit tangent directly: x / |x| (13), one sincos (36),
    the orthonormal basis of the tangent plane (22) and the combination (9) instead of two pairs and the projection."""
    pairs = (d + 1) // 2
    setup = 80.0 if d == 3 else 80.0 * pairs + 10.0 * d + 20.0
    if rng == "numpy":  # the reference's set-up: d ziggurat normals (common path: a product and a compare, ~4 each) and the projection
        setup = 4.0 * d + 10.0 * d + 20.0
    if name.startswith("vmfmix"):
        k = 3 if name.startswith("vmfmix_readme") else 10
        setup += 4.0 * k * d + 34.0 * k              # K dots with x and u; K exps for the level of x
        if k >= 5:  # screened accept test: one double exp of the largest term, K single-precision 2^x bounds
            per_try = 36.0 + 4.0 * k + k + 34.0 + 4.0 * k + 6.0
        else:
            per_try = 36.0 + 4.0 * k + 34.0 * k + 6.0  # K a_k(theta), K exps, sum, bracket
    elif name.startswith("bingham"):
        diag = not name.endswith("dense")
        setup += (10.0 * d if diag else 4.0 * d * d + 8.0 * d) + 40.0  # q-coefficients; log U
        per_try = 36.0 + 12.0 + 6.0
    else:  # curve: 10 knots, 9 segments
        setup += 4.0 * 10 * d + 9 * 28.0 + 40.0      # knot dots with x and u; level of x; log U
        per_try = 36.0 + 10 * 3.0 + 9 * 28.0 + 6.0
    return setup + tries_per_step * per_try


def oracle_target(orc, gs, name):
    """The same target for the CPU oracle, from the product object's plain arrays."""
    pdf, _ = make_target(gs, name)
    if isinstance(pdf, gs.MixtureModel):
        return orc.Target.vmf_mixture([p.mu for p in pdf.pdfs], pdf.weights)
    if isinstance(pdf, gs.Bingham):
        return orc.Target.bingham(pdf.A)
    return orc.Target.curve_vmf(pdf.curve.knots, pdf.kappa)


REFERENCE_NAMES = {"vmfmix_readme": "vmfmix_readme", "vmfmix_k10_kappa500": "vmfmix_k10_kappa500",
                   "bingham_d10": "bingham_d10_vmax30", "curve_d10": "curve_d10_kappa800",
                   "curve_d200": "curve_d200_kappa800"}
STREAM_NUMPY = ("numpy PCG64 + ziggurat, one generator per chain (SeedSequence(seed).spawn(n_chains)): the reference's arithmetic and "
                "draw order -- d normals projected onto the tangent plane, then the uniforms (mcmc.py:387-395)")


# What the library's counter-based stream draws per step: said in the bench line so that nobody reads
# "Philox" as "the reference's arithmetic on another generator"
STREAM_S2 = ("philox-v3: Philox4x32-10 keyed by (seed, chain, step); S^2: the unit tangent drawn as ONE angle (32 bit) instead of "
             "three projected normals, theta0 32 bit, threshold uniform 53 bit, a try's uniform ONE 32-bit word (four tries a block)")
STREAM_D = ("philox-v3: Philox4x32-10 keyed by (seed, chain, step); d normals per step by single-precision Box-Muller from 32-bit "
            "words (they only set the tangent's direction), threshold / theta0 uniforms 53 bit, a try's uniform ONE 32-bit word "
            "(four tries a block)")


def stream_description(d):
    return STREAM_S2 if d == 3 else STREAM_D


_digest = None


def source_digest():
    """sha256 of the kernel sources the LOADED libgsss_hip.so was built from (gsss_source_digest: embedded at build time).  The
    library ships prebuilt, so the files on disk prove nothing about the binary that is timed; profiles are bound to the binary."""
    global _digest
    if _digest is None:
        from geosss_amd import _lib
        _digest = _lib.load().gsss_source_digest().decode()
    return _digest


def tree_digest():
    """The same sha256 over the files lying on disk (geosss_amd.build.source_digest); differs from source_digest() when the
    library was built from other sources than the tree it sits in."""
    from geosss_amd import build
    return build.source_digest()


def profile_record(workload, n, S, thin, mode, layout):
    """The committed rocprofv3 PMC record of a workload (profiles/traffic.json, written by tools/pmc_traffic.py) -- or why it
    cannot be quoted: (record, None) when this run's launch has the profiled shape AND the kernel sources are the ones the
    profile was taken of (sha256 over geosss_amd/csrc + include/gsss.h), else (None, "absent" | "other launch shape" | "stale")."""
    try:
        rec = json.load(open(TRAFFIC_FILE))[workload]
    except (OSError, KeyError, ValueError):
        return None, "absent"
    shape = rec.get("launch", {})
    if (shape.get("chains"), shape.get("steps"), shape.get("thin"), shape.get("mode"), shape.get("layout", "components")) != (n, S, thin, mode, layout):
        return None, "other launch shape"
    if rec.get("csrc_sha256") != source_digest():
        return None, "stale"
    return rec, None


def issue_counters(workload, n, S, thin, mode, layout="chains"):
    """Issue-side counters of the dominant kernel from the committed rocprofv3 PMC passes, quoted only when this run's launch
    has the profiled shape and sources: how busy the vector pipes were, how many wavefronts were resident, and the FP64 flops
    the kernel actually ISSUED against the FP64 peak."""
    rec, why = profile_record(workload, n, S, thin, mode, layout)
    if rec is None:
        return {"counters": why}
    out = dict(rec.get("issue", {}))
    if out:
        out["valu_busy_meaning"] = ("4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): VALU-active wavefront cycles per SIMD "
                                    "cycle; instructions of different wavefronts overlap in the pipeline, so five resident wavefronts can pass 1")
    return out


def roofline_valu(name, d, tps, n, S, thin, mode, kern_ms, layout="chains"):  # name: workload[__numpy_stream][__all_double]
    """Delivered-work figure: the FP64 flops of the ALL-DOUBLE restricted-form algorithm whose decisions the kernel reproduces
    (algorithmic_flops) over the kernel time and the FP64 vector peak.  NOT a hardware utilisation: most per-try flops are
    executed in single precision; `fp64_issued_frac` and `valu_busy` are the hardware's own counters."""
    flops = algorithmic_flops(name, d, tps, rng="numpy" if "__numpy_stream" in name else "philox")
    out = {"bound": "fp64_valu", "achieved": flops * n * S / (kern_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
           "frac": flops * n * S / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, "flops_per_chain_step": flops,
           "meaning": "algorithmic FP64 flops of the all-double algorithm / kernel time / FP64 vector peak: delivered work, not "
                      "hardware utilisation (see fp64_issued_frac, valu_busy)"}
    iss = issue_counters(name, n, S, thin, mode, layout)
    for k in ("fp64_issued_frac", "valu_busy", "valu_busy_meaning", "resident_waves_per_simd", "valu_insts_per_chain_step", "lane_activity"):
        out[k] = iss.get(k)
    out["counters_source"] = iss.get("source")
    out["counters"] = iss.get("counters", "profiles/traffic.json, same launch shape and kernel sources")
    return out


def reference_timing(workload):
    """What the reference itself ran at in the build container (tests/golden/cpu_reference_timing.json,
    written by tests/golden/make_golden.py timing); it cannot run on the GPU box."""
    try:
        t = json.load(open(os.path.join(ROOT, "tests", "golden", "cpu_reference_timing.json")))
        r = t["targets"][REFERENCE_NAMES[workload]]
        return {"steps_per_s_1_core": r["steps_per_s_1_process"], "steps_per_s_all_cores": r["steps_per_s_aggregate_all_cores"],
                "cores": t["host"]["cores"], "cpu": t["host"]["cpu"], "where": "build container, geosss itself",
                "measured": "build container, not this run (the reference's files cannot travel to the GPU box)"}
    except (OSError, KeyError, ValueError):
        return None


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(gs, workload, d, budget_s=12.0):
    """The CPU oracle (C restatement of the reference loop, oracle/gsss_oracle.c) timed on the
    host cores on a bounded sample of the same workload (same target, same sampler, Philox stream).
    Runs BEFORE this process initialises the GPU (the numpy port forks worker processes)."""
    from oracle import oracle as orc
    tgt = oracle_target(orc, gs, workload)
    cores = host_cores()
    x0 = orc.sample_sphere(0, 1024 * cores, d)
    t0 = time.perf_counter()
    orc.run(tgt, x0, 10, seed=3521, keep_samples=False, n_threads=cores)
    rate = len(x0) * 10 / (time.perf_counter() - t0)
    n_chains = 4096 * cores
    n_steps = int(max(50, 0.5 * budget_s * rate / n_chains))
    x0 = orc.sample_sphere(0, n_chains, d)
    dts = []
    for rep in range(2):  # two samples, the better one quoted: the box's cores are shared, and one unlucky reading was 4 x low
        t0 = time.perf_counter()
        out = orc.run(tgt, x0, n_steps, seed=3521 + rep, keep_samples=False, n_threads=cores)
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    return {"value": n_chains * n_steps / dt, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_chains} chains x {n_steps} steps of the same target and sampler, C oracle with OpenMP over "
                      f"chains on {cores} threads, {dt:.1f} s (the better of two samples: {dts[0]:.1f}, {dts[1]:.1f} s)",
            "tries_per_step": float(out["n_tries"].sum() / (n_chains * n_steps)),
            "reference": reference_timing(workload), "numpy_port": numpy_port_baseline(workload, cores)}


def numpy_port_baseline(workload, cores, n_steps=400):
    """oracle/numpy_port.py -- a per-chain NumPy loop structured like the reference (one chain per
    process) -- timed on this box: 1 process and one process per available core."""
    if workload != "vmfmix_readme":
        return None
    from oracle import numpy_port
    x0 = np.array([-0.86, 0.19, -0.47])
    one = numpy_port.time_chains(README_MUS, x0, n_steps, 1)
    allc = numpy_port.time_chains(README_MUS, x0, n_steps, cores)
    return {"steps_per_s_1_core": one, "steps_per_s_all_cores": allc, "cores": cores,
            "sample": f"{n_steps} steps per chain, one chain per process"}


def ess_per_sec(gs, sampler, pdf, steps_per_sec_total, n_steps=4000, thin=4, lags=64):
    """Effective samples per second (secondary metric) of the WHOLE ensemble of the timed run: the sampler kernels
    accumulate the lag sums of the first coordinate per chain (gsss_run_args.stats_dev; no draws are stored -- at
    10^6 chains x 4000 steps they would be 96 GB) and the reference's own estimator (utils.acf + the IAT heuristic,
    geosss/utils.py:96-134, on the series thinned by `thin`) gives n_eff per chain; plus the mode occupancy."""
    from geosss_amd import diagnostics as dg
    sampler.enable_stats(lags=lags, second_moment=False)
    sampler.advance(n_steps, thin=thin, keep=False)
    r = sampler.stats()
    rel = float((r["n_eff"] / r["n"]).mean().item()) / thin        # effective draws per chain-step
    trunc = float(r["iat_truncated"].double().mean().item())
    # the same heuristic on the autocorrelation AVERAGED over the chains (no per-chain noise, no Jensen gap of mean(1 / IAT))
    iat_pooled = float(dg.iat_from_acf(r["acf"].mean(0, keepdim=True))[0].item()) * thin
    # and with no window at all: the chains are independent and (after the timed launches) stationary, so the spread of their
    # means measures tau directly (diagnostics.ess_between_chains)
    bc = dg.ess_between_chains(r["proj_mean"], r["n"], r["proj_var"])
    tau_steps = bc["tau"] * thin
    out = {"ess_per_step": rel, "ess_per_sec": rel * steps_per_sec_total,
           "iat_truncated_frac": trunc,  # chains whose pair sums never went negative within the lags: their IAT is a lower bound
           "ess_per_step_is": "estimate" if trunc < 0.02 else "upper bound: the lag window cuts the autocorrelation of a share of the chains (between_chains is the estimate)",
           "iat_steps_pooled_acf": iat_pooled,
           "between_chains": {"tau_steps": tau_steps, "ess_per_step": 1.0 / tau_steps, "ess_per_sec": steps_per_sec_total / tau_steps,
                              "rel_se": bc["rel_se"], "chains": bc["chains"], "steps_per_chain": n_steps,
                              "estimator": "tau = n Var_chains(chain mean) / Var(x) over all chains of the launch (independent, stationary): "
                                           "no lag window; finite-length bias ~ -tau / n"},
           "estimator": f"geosss IAT heuristic on the running autocorrelation of the first coordinate (lags <= {lags} x {thin} "
                        f"steps), all {sampler.n_chains} chains x {n_steps} steps, no stored draws"}
    if "mode_occupancy" in r:
        out["mode_occupancy"] = [float(v) for v in r["mode_occupancy"].mean(0).cpu()]
    return out


def ess_between_chains_doubling(gs, torch, sampler, pdf, workload, steps_per_sec_total, budget_s=2.0, sub_chains=16384):
    """Slowly mixing targets (the curve-vMF family: the chain creeps along the curve, tau of the first coordinate is 10^4 .. 10^6
    steps): no window of a bench-sized launch holds the autocorrelation, and a between-chain estimate is only as good as
    its window is long (n >> tau) and its chains are stationary (burn-in >> tau).  So: doubling windows on a sub-ensemble of the
    timed chains (every window is the burn-in of the next), within a time budget; the last window's tau with its length over
    tau, `converged` when that ratio is >= 20 and tau moved < 10 % against the window before -- otherwise tau is a LOWER bound,
    and the figure of the run-out measurement (tools/ess_convergence.py, profiles/r04_ess_convergence.json) is quoted beside it."""
    from geosss_amd import diagnostics as dg
    m = min(sub_chains, sampler.n_chains)
    sub = gs.ShrinkageSphericalSliceSampler(pdf, sampler.state_device[:, :m].T.contiguous(), seed=3521, chain_offset=sampler.chain_offset,
                                            step_offset=sampler._step, placement="packed")
    steps, used, prev, hist = 8192, 0.0, None, []
    rate = None
    while True:
        if rate is not None and used + steps / rate > budget_s:
            break
        thin = max(1, steps // 512)
        sub.enable_stats(lags=4, second_moment=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sub.advance(steps, thin=thin, keep=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        used += dt
        rate = steps / dt
        r = sub.stats()
        bc = dg.ess_between_chains(r["proj_mean"], r["n"], r["proj_var"])
        tau = bc["tau"] * thin
        hist.append({"window_steps": steps, "tau_steps": tau})
        change = abs(tau / prev - 1.0) if prev else None
        prev = tau
        steps *= 2
    last = hist[-1]
    ratio = last["window_steps"] / last["tau_steps"]
    converged = bool(ratio >= 20.0 and change is not None and change < 0.10)
    out = {"tau_steps": last["tau_steps"], "tau_is": "estimate" if converged else "lower bound (window / tau too small or still moving)",
           "window_steps": last["window_steps"], "window_over_tau": ratio, "change_vs_previous_window": change, "converged": converged,
           "ess_per_step": 1.0 / last["tau_steps"], "ess_per_sec": steps_per_sec_total / last["tau_steps"],
           "ess_is": "estimate" if converged else "upper bound", "chains": m, "rel_se": bc["rel_se"], "windows": hist,
           "estimator": "tau = n Var_chains(chain mean) / Var(x) on doubling windows of a sub-ensemble of the timed chains (every window "
                        "is the next one's burn-in); no lag window"}
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "r04_ess_convergence.json")))[workload]
        out["run_out"] = dict(ref, ess_per_step=1.0 / ref["tau_steps"], ess_per_sec=steps_per_sec_total / ref["tau_steps"],
                              source="profiles/r04_ess_convergence.json (tools/ess_convergence.py: the same estimator run until the "
                                     "window is >= 16 tau; a property of target and sampler, not of the kernel build)")
    except (OSError, KeyError, ValueError):
        pass
    return out


def measured_traffic(workload, n, S, thin, mode, layout="chains"):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); only quoted when this
    run's launch has the shape the profile was taken with."""
    rec, why = profile_record(workload, n, S, thin, mode, layout)
    if rec is None:
        return None, f"counters: {why}"
    return rec["bytes_per_launch"], rec["source"]


def hbm_bytes_per_step(d, thin, S):
    """ALGORITHMIC HBM bytes per chain-step, SURVEY.md section 8(d): the retained sample 8 d / thin plus the state load and
    store and the counter words amortised over the S steps of a launch, (16 d + 16) / S."""
    return 8.0 * d / thin + (16.0 * d + 16.0) / S


def handover_bytes_per_step(d, S, slice_steps=0, sliced_fraction=0.0):
    """What the library's own scheduling adds: the SLICED share of a launch hands state (16 d), the two
    64-bit counters (read and written: 32) over through HBM once per slice instead of once per launch.  Traffic,
    not algorithm: reported beside the algorithmic bytes, never inside them."""
    if slice_steps <= 0 or sliced_fraction <= 0.0:
        return 0.0
    return sliced_fraction * (16.0 * d + 32.0) * (1.0 / min(S, slice_steps) - 1.0 / S)


def hbm_roofline(d, thin, S, n, kern_ms, slice_steps, sliced_frac, traffic, traffic_src, note=None):
    """The `roofline` object of a bench line: `achieved` / `frac` from the ALGORITHMIC bytes of SURVEY.md section 8(d) alone;
    the hand-over of sliced launches and the counters' total beside them."""
    alg = hbm_bytes_per_step(d, thin, S) * n * S
    hand = handover_bytes_per_step(d, S, slice_steps, sliced_frac) * n * S
    achieved = alg / (kern_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "algorithmic_bytes": alg, "handover_bytes": hand, "traffic": traffic,
           "traffic_ratio": (traffic / alg) if traffic else None, "traffic_source": traffic_src,
           "bytes_meaning": "algorithmic_bytes = (8 d / thin + (16 d + 16) / S) x chain-steps of a launch (SURVEY.md section 8(d)); "
                            "handover_bytes = state and counters of the sliced share of the launch passed from slice to slice "
                            "through HBM (self-inflicted, not in `achieved`); traffic = 2 x FETCH_SIZE + WRITE_SIZE of the PMC passes"}
    if note:
        out["note"] = note
    return out


def last_slice_steps(gs):
    """(slice length, share of the chains that ran sliced) of this thread's last sampler launch ((0, 0.0): unsliced),
    gsss_last_launch."""
    import ctypes as C
    grid, steps, frac = C.c_int64(0), C.c_int32(0), C.c_double(0.0)
    gs._lib.load().gsss_last_launch(C.byref(grid), C.byref(steps), C.byref(frac))
    return int(steps.value), float(frac.value)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """N fresh child ranks under torch.distributed.run.  This parent has not imported torch and never
    touches the GPU; rank 0's JSON line reaches stdout through the inherited pipe."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def pick_layout(layout, kernel_name, d, thin=100):
    """--layout auto: the layout that moves the fewest bytes for the kernel family (measured, tools/layout_experiment.sh; both are
    what the C ABI offers, gsss_run_args.samples_chain_rows).  The group kernels step a wavefront's chains together, so a kept
    row leaves as whole 512-byte runs per component: component-major.  The lane kernels keep a chain's row when ITS step count
    says so: 8-byte stores into eight chains' shared 64-byte lines at different times (WRITE_SIZE 2.6 GB for the 0.8 GB of
    kept rows of Bingham d = 10); chain-major, a chain's 8 d bytes are contiguous and leave together (1.2 GB) -- unless a row is
    smaller than a line AND rows are far apart in time (d = 3, thin = 100: 24 bytes; chain-major 0.55 GB against 0.29 GB).  When
    nearly every state is kept (thin <= 16) a chain's consecutive rows fill its lines within a few steps: chain-major again
    (d = 3, thin = 1, 10^6 chains x 200 steps: 6.6 against 9.1 ms per launch, 3.0e10 against 2.2e10 chain-steps/s)."""
    if layout != "auto":
        return layout
    lane = kernel_name.startswith(("screened_kernel", "fast_kernel"))
    return "chains" if lane and (8 * d >= 64 or thin <= 16) else "components"


def kept_buffer(torch, layout, n, S, thin, d):
    """Where the retained rows of one launch go.  "chains": (chains, draws, dims), the reference's own order (what
    `sampler.sample()` returns; a chain appends its 8 d bytes per kept row to its own run) -- "components": [row][d][chain], the
    kernels' component-major state layout."""
    if layout == "chains":
        return torch.empty((n, S // thin, d), dtype=torch.float64, device="cuda"), dict(chain_major=True, row0=0)
    return torch.empty((S // thin, d, n), dtype=torch.float64, device="cuda"), dict()


def numpy_stream_kernel_name(gs, sampler, packed_name):
    """The kernel a packed launch on numpy's stream runs (gsss_kernel_name names the library stream's): the screened lane kernel's
    NUMPY build for mixtures of K <= 10 and Bingham targets at d <= 10 (gsss_fast_vmf_lane.h, gsss_fast_bingham.hip), else the
    all-double lane kernel's."""
    import re
    m = re.match(r"screened_kernel<(\d+), Screen(Vmf<\d+, (\d+)>|Bingham\w*<\d+>)>", packed_name)
    if m and int(m.group(1)) <= 10 and (m.group(3) is None or int(m.group(3)) <= 10):
        return packed_name[:-1] + ", NUMPY>"
    lib = gs._lib.load()
    name = lib.gsss_kernel_name(sampler._target_dev.handle, gs._lib.MODE_FAST, gs._lib.VARIANT_FAST_DOUBLE, 1).decode()
    return name[:-1] + ", NUMPY>" if name.startswith("fast_kernel") else name


def time_config(gs, torch, name, n, S, seed=3521, ess=True, layout="auto", rng="philox", screen=True):
    """<= ~1 s of one of the other BASELINE configs: same launch shape as the headline workload.  rng="numpy": the same launch
    on numpy's own stream, one generator per chain, packed (fast_kernel<..., NUMPY>).  screen=False: the all-double kernel
    (GSSS_VARIANT_FAST_DOUBLE: every try decided in double precision, no single-precision screen)."""
    pdf, d = make_target(gs, name)
    x0 = gs.sample_sphere_device(d - 1, n, seed=0)
    kw_s = dict(rng="numpy", placement="packed") if rng == "numpy" else {}
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=seed, screen=screen, **kw_s)
    thin = 100
    lib = gs._lib.load()
    mode_id = gs._lib.MODE_FAST if s.mode == "fast" else gs._lib.MODE_EXACT
    kernel = lib.gsss_kernel_name(s._target_dev.handle, mode_id, 0 if screen else gs._lib.VARIANT_FAST_DOUBLE, 1).decode()
    if rng == "numpy":
        kernel = numpy_stream_kernel_name(gs, s, kernel)
    layout = pick_layout(layout, kernel, d, thin)
    kept, kw = kept_buffer(torch, layout, n, S, thin, d)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    s.advance(100)                                   # warm-up: 100 transitions (cfg: "warm-up 100 steps")
    b.record()
    torch.cuda.synchronize()
    reps = int(max(1, min(10, 1.0 / max(1e-4, a.elapsed_time(b) * 1e-3 * S / 100))))
    tries0 = int(s._n_tries.sum().item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        s.advance(S, thin=thin, out=kept, **kw)
        b.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    tps = (int(s._n_tries.sum().item()) - tries0) / (n * S * reps)
    slice_steps, sliced_frac = last_slice_steps(gs)
    wl_key = name + ("__numpy_stream" if rng == "numpy" else "") + ("" if screen else "__all_double")
    traffic, src = measured_traffic(wl_key, n, S, thin, s.mode, layout)
    value = n * S * reps / dt
    # ESS / s of the whole ensemble from the running lag sums: thin so that ~64 lags span the autocorrelation (slow targets: Bingham)
    # (the curve targets mix over a thousand steps and more: thin 128, so that the 64 lags span 8192 steps)
    ess_steps, ess_thin = (32768, 128) if name.startswith("curve") else ((2000, 8) if name.startswith("bingham") else (2000, 4))
    ess_out = ess_per_sec(gs, s, pdf, value, n_steps=ess_steps, thin=ess_thin) if ess and rng != "numpy" else None
    if ess_out is not None and name.startswith("curve"):
        # (the one-window figure above took the timed launches as burn-in: ~10^4 steps against tau = 10^4 .. 10^6 -- not stationary,
        # not long enough; it stays in the line as `one_window`, the doubling windows replace it)
        ess_out["between_chains_one_window"] = ess_out.pop("between_chains")
        ess_out["between_chains"] = ess_between_chains_doubling(gs, torch, s, pdf, name, value)
    return {"name": wl_key,
            "workload": f"{name}: shrinkage slice sampler, {n} chains x {S} transitions per launch, thin={thin}"
                        + (", rng=numpy (one PCG64 generator per chain), packed" if rng == "numpy" else "")
                        + ("" if screen else ", screen=False (all-double kernel)"),
            "stream": STREAM_NUMPY if rng == "numpy" else stream_description(d), "slice_steps": slice_steps,
            "sliced_fraction": round(sliced_frac, 4),
            "value": value, "unit": "chain-steps/s", "launches": reps, "mode": s.mode, "ess": ess_out,
            "kernel": kernel,
            "kernel_ms": kern_ms, "tries_per_step": tps, "chains_in_error": int((s._err != 0).sum().item()),
            "roofline": hbm_roofline(d, thin, S, n, kern_ms, slice_steps, sliced_frac, traffic, src),
            "kept_rows_layout": layout,
            "roofline_valu": roofline_valu(wl_key, d, tps, n, S, thin, s.mode, kern_ms, layout)}


def host_api_rate(gs, torch, pdf, d, n, draws=100):
    """PCIe-INCLUSIVE rate of the reference-shaped call (never `value`): `sampler.sample(draws)` returns every draw of every
    chain as an ndarray (geosss/mcmc.py:55-77; scripts/curve_vMF.py:119-120 reads it on the host) -- page-locked memory, blocks of
    chains copied under the next block's kernel (geosss_amd/mcmc.py _sample_to_host).  cold: the first call of its size (locks the
    pages); warm: the previous array was dropped and its pages are reused."""
    from geosss_amd import _pinned
    x0 = gs.sample_sphere_device(d - 1, n, seed=0).T.contiguous()
    out = {"call": f"sample({draws}) -> ndarray, {n} chains", "array_gb": 8e-9 * n * draws * d, "chain_steps": n * (draws - 1)}
    _pinned.trim()
    for label in ("cold", "warm"):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=3521)
        s.advance(50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        arr = s.sample(draws)
        dt = time.perf_counter() - t0
        out[label] = {"s": dt, "chain_steps_per_s": n * (draws - 1) / dt, "gb_per_s": arr.nbytes / dt / 1e9,
                      "blocks": s._plan_blocks(0, draws, 1, None)}
        del arr, s
    _pinned.trim()
    return out


def time_sharded_config(gs, torch, dist, name, n, S, rank, world, barrier, gather_states, seed=3521, thin=100):
    """Another BASELINE config timed on EVERY rank of a multi-GPU run (cfg5: vMF mixture K = 10 kappa = 500, 10^6 chains per GPU,
    chain ids rank * 10^6 ..., final states gathered: SURVEY.md section 8(d)/(e); the reference's fan-out of independent
    processes, scripts/mixture_vMF.py:137-149).  Same bracketing as the headline: barrier + synchronize on both sides, the gather
    inside the timed region, max over ranks."""
    n = int(os.environ.get("GSSS_BENCH_SHARDED_CHAINS", n))          # (tests rehearse the path with a smaller ensemble)
    pdf, d = make_target(gs, name)
    x0 = gs.sample_sphere_device(d - 1, n, seed=0, chain_offset=rank * n)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=seed, chain_offset=rank * n)
    lib = gs._lib.load()
    kernel = lib.gsss_kernel_name(s._target_dev.handle, gs._lib.MODE_FAST if s.mode == "fast" else gs._lib.MODE_EXACT, 0, 1).decode()
    layout = pick_layout("auto", kernel, d, thin)
    kept, kw = kept_buffer(torch, layout, n, S, thin, d)
    counts = [n] * world
    s.advance(100)                                   # warm-up: 100 transitions
    gather_states(s.state_device, counts=counts)
    reps = 5
    tries0 = int(s._n_tries.sum().item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        s.advance(S, thin=thin, out=kept, **kw)
        b.record()
    g0.record()
    final = gather_states(s.state_device, counts=counts)
    g1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    ones = torch.ones(1, dtype=torch.int64, device="cuda")
    agg = torch.tensor([int(s._n_tries.sum().item()) - tries0, int((s._err != 0).sum().item())], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ones)
        dist.all_reduce(agg)
    assert final.shape[1] == n * world
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    total = n * S * reps * world
    digest = None
    if os.environ.get("GSSS_BENCH_CHECKSUM"):  # (tests: the gathered [d, world * n] states, to be compared with a one-process run)
        import hashlib
        digest = hashlib.sha256(final.contiguous().cpu().numpy().tobytes()).hexdigest()
    return {"name": name + "_sharded", "final_sha256": digest, "chains_per_gpu": n, "workload": f"{name}: {n} chains per GPU x {S} transitions per launch, thin={thin}, chain ids "
                                                   f"rank * {n} ..., final states gathered, {world} ranks",
            "value": total / float(tmax.item()), "unit": "chain-steps/s", "launches": reps, "kernel": kernel, "kernel_ms": kern_ms,
            "gather_ms": float(g0.elapsed_time(g1)), "ranks_seen": int(ones.item()), "tries_per_step": int(agg[0].item()) / total,
            "chains_in_error": int(agg[1].item()), "kept_rows_layout": layout,
            "roofline": hbm_roofline(d, thin, S, n, kern_ms, *last_slice_steps(gs), None, "counters: not collected on a multi-rank run")}


def _num(v, digits=6):
    """Floats of the printed line at 6 significant digits (the full record keeps every digit)."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if v == v and abs(v) != float("inf") else None   # (strict JSON: no NaN / Infinity in the line)
    if isinstance(v, dict):
        return {k: _num(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


LINE_LIMIT = 6000  # bytes: the driver reads the LAST stdout line out of an 8 KB tail (round 4's 34.7 KB line was not parsed)


def compact_line(full, full_path=None):
    """The ONE JSON line bench.py prints: numbers and short names only, every prose field (the `*_meaning`, `note`,
    `estimator`, stream descriptions) and the per-config detail stay in the full record (bench_full.json; DESIGN.md section 6/7
    explains each field).  configs[] = [{name, value, kernel_ms, hbm_frac, valu_frac, traffic_ratio}]."""
    cfg = full["config"]
    out = _pick(full, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"])
    out["config"] = _pick(cfg, ["workload", "target", "d", "chains_per_gpu", "transitions_per_step", "thin", "mode", "kernel",
                                "kept_rows_layout", "slice_steps", "sliced_fraction", "csrc_sha256", "library_matches_tree"])
    out["config"]["stream"] = cfg["stream_short"]
    by_name = {c["name"]: c for c in full.get("configs", [])}
    # the same target, launch shape and sampler (a) on numpy's own stream = the reference's arithmetic and draw order,
    # mcmc.py:382-401 from the seed; (b) in the all-double kernel (no single-precision screen)
    out["value_numpy_stream"] = by_name.get(cfg["target"] + "__numpy_stream", {}).get("value")
    out["value_all_double"] = by_name.get(cfg["target"] + "__all_double", {}).get("value")
    out.update(_pick(full, ["tries_per_step", "chains_in_error", "kernel_ms"]))
    out["roofline"] = _pick(full["roofline"], ["bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes", "handover_bytes",
                                               "traffic", "traffic_ratio"])
    src = full["roofline"].get("traffic_source") or ""
    out["roofline"]["traffic_source"] = src.split(" (")[0][:80]
    out["roofline_valu"] = _pick(full["roofline_valu"], ["bound", "achieved", "peak", "unit", "frac", "flops_per_chain_step",
                                                         "fp64_issued_frac", "valu_busy", "resident_waves_per_simd",
                                                         "valu_insts_per_chain_step", "lane_activity"])
    out["roofline_valu"]["counters"] = "profiles/traffic.json" if full["roofline_valu"].get("valu_busy") is not None else \
        str(full["roofline_valu"].get("counters"))[:40]
    if full.get("cpu_baseline"):
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = _pick(cb, ["value", "unit", "cores", "kind"])
        out["cpu_baseline"]["sample"] = cb["sample"].split(" of the same")[0] + f", C oracle, {cb['cores']} threads"
        if cb.get("numpy_port"):
            out["cpu_baseline"]["numpy_port_1_core"] = cb["numpy_port"]["steps_per_s_1_core"]
        if cb.get("reference"):
            out["cpu_baseline"]["reference_1_core_build_container"] = cb["reference"]["steps_per_s_1_core"]
    if full.get("ess"):
        e = full["ess"]
        out["ess"] = {"ess_per_step": e["ess_per_step"], "ess_per_sec": e["ess_per_sec"],
                      "tau_steps_between_chains": e["between_chains"]["tau_steps"]}
    r = full.get("rccl") or {}
    out["rccl"] = _pick(r, ["ranks_seen", "backend", "gather_ms"])
    if "kernel_ms_per_rank" in r:
        out["rccl"]["kernel_ms_min_max"] = [r["kernel_ms_per_rank"]["min"], r["kernel_ms_per_rank"]["max"]]
        out["rccl"]["wall_ms_min_max"] = [r["wall_ms_per_rank"]["min"], r["wall_ms_per_rank"]["max"]]
    rows = []
    for c in full.get("configs", []):
        row = {"name": c["name"], "value": c["value"], "kernel_ms": c["kernel_ms"], "hbm_frac": c["roofline"]["frac"],
               "valu_frac": (c.get("roofline_valu") or {}).get("frac"), "traffic_ratio": c["roofline"].get("traffic_ratio")}
        row.update(_pick(c, ["gather_ms", "ranks_seen"]))
        rows.append(row)
    out["configs"] = rows
    if full.get("host_api"):  # PCIe-inclusive, never `value`: sample(100) -> ndarray end to end
        h = full["host_api"]
        out["host_api"] = {"chain_steps_per_s": h["warm"]["chain_steps_per_s"], "gb_per_s": h["warm"]["gb_per_s"],
                           "cold_chain_steps_per_s": h["cold"]["chain_steps_per_s"]}
    if full_path:
        out["full_record"] = full_path
    line = json.dumps(_num(out), separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def write_full_record(full):
    """The complete record (every field of rounds 1-4's line and more) beside the printed line: ./bench_full.json, and under
    gpurun_out/ too when that directory exists (it is what travels back from a GPU box)."""
    path = None
    for cand in (os.path.join(ROOT, "gpurun_out", "bench_full.json"), os.path.join(ROOT, "bench_full.json")):
        try:
            if os.path.isdir(os.path.dirname(cand)):
                with open(cand, "w") as f:
                    json.dump(full, f, indent=1)
                path = path or os.path.relpath(cand, ROOT)
        except OSError:
            pass
    return path



def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=1_000_000, help="chains per GPU")
    ap.add_argument("--inner", type=int, default=1000, help="MCMC transitions per launch (= per bench step)")
    ap.add_argument("--workload", default="vmfmix_readme")
    ap.add_argument("--thin", type=int, default=100, help="keep every thin-th state (cfg2: every 100th)")
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--layout", default="auto", choices=["auto", "chains", "components"],
                    help="retained rows: (chains, draws, dims) as the reference returns them, the kernels' [row][d][chain], or "
                         "(auto) whichever moves fewer bytes for the kernel family, see pick_layout")
    ap.add_argument("--rng", default="philox", choices=["philox", "numpy"],
                    help="numpy: the workload on numpy's own stream (one PCG64 generator per chain, packed) -- what configs[] times "
                         "as <workload>__numpy_stream; for profiling that entry on its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs")
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("need --gpus >= 1, --steps >= 1, --warmup >= 0")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return self_launch(args, argv)              # nothing below has run: no torch, no HIP in this process
    world = int(env_world or "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: RCCL needs it on this driver; set before torch loads HIP)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}\n")
        return 2
    backend = os.environ.get("GSSS_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl":
        import torch  # (device_count() does not initialise the GPU)
        if torch.cuda.device_count() < world:
            if rank == 0:
                sys.stderr.write(f"bench.py: --gpus {world} over RCCL needs {world} visible GPUs, this node shows "
                                 f"{torch.cuda.device_count()} (one rank per GPU; GSSS_BENCH_BACKEND=gloo rehearses the ranks on fewer)\n")
            return 3

    import geosss_amd as gs                          # imports torch; does not initialise the GPU
    pdf, d = make_target(gs, args.workload)
    # CPU baseline first: rank 0 of a single-GPU run, before this process holds a HIP context
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(gs, args.workload, d)

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # one process per GPU; GSSS_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the
    # multi-rank code path on a single-GPU box (RCCL refuses two ranks on one device)
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from geosss_amd.ensemble import gather_states

    n = args.chains
    chain_offset = rank * n
    x0 = gs.sample_sphere_device(d - 1, n, seed=0, chain_offset=chain_offset)  # [d, n] on device
    kw_rng = dict(rng="numpy", placement="packed") if args.rng == "numpy" else {}
    sampler = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521, chain_offset=chain_offset, mode=args.mode,
                                                variant=args.variant, **kw_rng)
    wl_key = args.workload + ("__numpy_stream" if args.rng == "numpy" else "")
    S = args.inner
    thin = min(args.thin, S) if args.thin > 0 else S
    layout = pick_layout(args.layout, gs._lib.load().gsss_kernel_name(
        sampler._target_dev.handle, gs._lib.MODE_FAST if sampler.mode == "fast" else gs._lib.MODE_EXACT, args.variant, 1).decode(), d, thin)
    kept, kw = kept_buffer(torch, layout, n, S, thin, d)
    counts = [n] * world                             # chains per rank are fixed: no size exchange per gather

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sampler.advance(S, thin=thin, out=kept, **kw)
    rccl = {"ranks_seen": 1, "backend": None, "gather_ms": 0.0}
    if world > 1:  # the collective's lazy channel setup must not land inside the timed region
        gather_states(sampler.state_device, counts=counts)
        ones = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(ones)                        # every rank contributes 1: proves N live ranks on the backend
        barrier()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        gather_states(sampler.state_device, counts=counts)
        g1.record()
        torch.cuda.synchronize()
        rccl = {"ranks_seen": int(ones.item()), "backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend,
                "gather_ms": float(g0.elapsed_time(g1)), "gather_bytes_per_rank": 8 * d * n}
    tries0 = int(sampler._n_tries.sum().item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        sampler.advance(S, thin=thin, out=kept, **kw)
        b.record()
    final = gather_states(sampler.state_device, counts=counts) if world > 1 else sampler.state_device
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert final.shape[1] == n * world

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    headline_slices = last_slice_steps(gs)
    if world > 1:  # per-rank kernel time and this rank's own wall clock: imbalance between GPUs shows in the line
        mine = torch.tensor([kern_ms, (time.perf_counter() - t0) * 1e3], dtype=torch.float64, device="cuda")
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = torch.stack(per_rank).cpu().numpy()
        rccl["kernel_ms_per_rank"] = {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max()),
                                      "all": [float(v) for v in per_rank[:, 0]]}
        rccl["wall_ms_per_rank"] = {"min": float(per_rank[:, 1].min()), "max": float(per_rank[:, 1].max())}
    tries = int(sampler._n_tries.sum().item()) - tries0
    bad = int((sampler._err != 0).sum().item())
    chain_steps = n * S * args.steps
    if world > 1:
        agg = torch.tensor([tries, bad], dtype=torch.int64, device="cuda")
        dist.all_reduce(agg)
        tries, bad = int(agg[0].item()), int(agg[1].item())
    total_steps = chain_steps * world
    value = total_steps / elapsed

    sharded = None
    if world > 1 and not args.no_configs and args.workload == "vmfmix_readme":
        del kept                                      # every rank: cfg5's own workload, sharded like the headline
        sharded = time_sharded_config(gs, torch, dist, *SHARDED_CONFIG, S, rank, world, barrier, gather_states)

    if rank == 0:
        slice_steps, sliced_frac = headline_slices
        lib = gs._lib.load()
        mode_id = gs._lib.MODE_FAST if sampler.mode == "fast" else gs._lib.MODE_EXACT
        traffic, traffic_src = measured_traffic(wl_key, n, S, thin, sampler.mode, layout)
        out = {
            "metric": "mcmc_chain_steps_per_sec", "value": value, "unit": "chain-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: shrinkage slice sampler, {n} chains/GPU x {S} transitions per "
                                   "launch, thin=%d" % thin,
                       "stream": STREAM_NUMPY if args.rng == "numpy" else stream_description(d),
                       "stream_short": "numpy PCG64+ziggurat per chain (mcmc.py:387-395 order)" if args.rng == "numpy" else (
                           "philox-v3 (chain,step)-keyed; S2: tangent = one angle; 32-bit tries" if d == 3 else
                           "philox-v3 (chain,step)-keyed; f32 Box-Muller normals; 32-bit tries"),
                       "target": wl_key, "d": d, "chains_per_gpu": n, "transitions_per_step": S, "thin": thin,
                       "slice_steps": slice_steps, "sliced_fraction": round(sliced_frac, 4),
                       "kept_rows_layout": layout,
                       "csrc_sha256": source_digest(),           # of the LOADED library (gsss_source_digest)
                       "library_matches_tree": source_digest() == tree_digest(),
                       "mode": sampler.mode,
                       "kernel": (numpy_stream_kernel_name(gs, sampler, lib.gsss_kernel_name(sampler._target_dev.handle, mode_id, 0, 1).decode())
                                  if args.rng == "numpy" else lib.gsss_kernel_name(sampler._target_dev.handle, mode_id, args.variant, 1).decode()),
                       "sharding": f"{world} x independent chain blocks, final states all-gathered over RCCL"
                       if world > 1 else "single GPU"},
            "tries_per_step": tries / total_steps, "chains_in_error": bad,
            "kernel_ms": kern_ms, "rccl": rccl,
            "roofline": hbm_roofline(d, thin, S, n, kern_ms, slice_steps, sliced_frac, traffic, traffic_src,
                                     note="chain state lives in registers/LDS for the whole launch, so HBM sees only the "
                                          "state load/store, counters and the thinned sample; the kernel is bound by vector "
                                          "issue (see roofline_valu and DESIGN.md)"),
            "roofline_valu": roofline_valu(wl_key, d, tries / total_steps, n, S, thin, sampler.mode, kern_ms, layout),
        }
        if world == 1 and not args.no_ess and args.rng != "numpy":
            out["ess"] = ess_per_sec(gs, sampler, pdf, value)
        if world == 1 and not args.no_configs and args.workload == "vmfmix_readme":
            del sampler, kept
            out["configs"] = [time_config(gs, torch, name, nc, S, layout=args.layout) for name, nc in EXTRA_CONFIGS]
            out["configs"].append(time_config(gs, torch, *NUMPY_STREAM_CONFIG, S, layout=args.layout, rng="numpy"))
            out["configs"].append(time_config(gs, torch, *NUMPY_STREAM_CONFIG, S, layout=args.layout, screen=False, ess=False))
            out["host_api"] = host_api_rate(gs, torch, pdf, d, n)
        if sharded is not None:
            out["configs"] = [sharded]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(compact_line(out, write_full_record(out)), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

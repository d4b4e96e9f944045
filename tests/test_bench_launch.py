"""`python bench.py --gpus N` starts its own N ranks (the analogue of the reference's fan-out,
scripts/curve_vMF.py:205-267): the parent never initialises the GPU, the children rendezvous on 127.0.0.1.

CPU leg: the launch is exercised up to the point where a rank needs a GPU.  GPU leg: two gloo ranks share the
one GPU of the test box and the whole multi-rank path (sharded chain ids, gather, reductions, JSON line) runs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_line_fits():
    """The driver reads the LAST stdout line out of an 8 KB tail: round 4's line had grown to 34.7 KB and was not parsed
    (BENCH_r04.json: parsed null).  The printed line is a compact projection of the full record; here the projection of a
    complete single-GPU record (the committed profiles/r0*_bench_full.json / round 4's 34.7 KB line, all nine configs, every
    optional object present) must stay below 6000 bytes, parse, and carry what the driver and the judge read."""
    import glob
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_full.json"))) or [os.path.join(ROOT, "profiles", "r04_bench_default.json")]
    full = json.load(open(recs[-1]))
    full["config"].setdefault("stream_short", "philox-v3 (chain,step)-keyed; S2: tangent = one angle; 32-bit tries")
    full["config"].setdefault("thin", 100)
    for c in full["configs"]:
        c.setdefault("name", c["workload"].split(":")[0] + ("__numpy_stream" if "rng=numpy" in c["workload"] else ""))
    if not any(c["name"].endswith("__all_double") for c in full["configs"]):
        full["configs"].append(dict(full["configs"][-1], name="vmfmix_readme__all_double"))
    # worst case for the length: a multi-rank record's per-rank fields on top
    full["rccl"] = {"ranks_seen": 8, "backend": "rccl (torch.distributed nccl)", "gather_ms": 1.2345678, "gather_bytes_per_rank": 24000000,
                    "kernel_ms_per_rank": {"min": 24.123456, "max": 24.654321, "all": [24.5] * 8}, "wall_ms_per_rank": {"min": 241.23456, "max": 246.54321}}
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(line) < 6000 and "\n" not in line, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_valu", "cpu_baseline", "configs", "value_numpy_stream", "value_all_double"):
        assert k in out, k
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["value_numpy_stream"] and out["value_all_double"] and len(out["configs"]) >= 9
    assert all(set(c) >= {"name", "value", "kernel_ms", "hbm_frac", "valu_frac", "traffic_ratio"} for c in out["configs"])
    assert "model" not in out["config"] and len(out["config"]["stream"]) <= 80
    assert not any(k.endswith("meaning") or k == "note" for k in (*out["roofline"], *out["roofline_valu"]))


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_rccl_ranks_need_one_gpu_each():
    """Under the RCCL backend a rank per GPU: fewer visible GPUs than --gpus is refused with a one-line reason and a non-zero
    exit before any process group is formed (nothing hangs in a rendezvous)."""
    import torch
    have = torch.cuda.device_count()
    want = have + 2
    r = _run(["--gpus", str(want), "--no-cpu-baseline"], {"WORLD_SIZE": str(want), "RANK": "0", "LOCAL_RANK": "0",
                                                          "GSSS_BENCH_BACKEND": "nccl"}, timeout=120)
    assert r.returncode == 3, r.stderr[-2000:]
    assert f"needs {want} visible GPUs" in r.stderr and not r.stdout.strip()
    assert r.stderr.strip().count("\n") == 0      # one line


def test_self_launch_reaches_the_gpu_check():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the gpu leg runs the launch end to end")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--chains", "256", "--inner", "10", "--no-cpu-baseline"],
             {"GSSS_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    # both child ranks got as far as the device check of the product (no CPU fallback)
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]
    assert not r.stdout.strip()


@pytest.mark.gpu
def test_two_ranks_end_to_end_on_one_gpu():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "20000", "--inner", "100", "--thin", "50",
              "--no-cpu-baseline", "--no-ess"], {"GSSS_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["rccl"]["ranks_seen"] == 2 and out["rccl"]["backend"] == "gloo"
    assert out["chains_in_error"] == 0 and out["value"] > 0
    assert 4.5 < out["tries_per_step"] < 5.6
    # imbalance between ranks is visible in the line: per-rank kernel time (min, max) and wall clock, the gather's time -- every
    # rank's kernel time in the full record
    kmin, kmax = out["rccl"]["kernel_ms_min_max"]
    assert 0 < kmin <= kmax and out["rccl"]["gather_ms"] > 0
    assert 0 < out["rccl"]["wall_ms_min_max"][0] <= out["rccl"]["wall_ms_min_max"][1]
    assert "philox-v3" in out["config"]["stream"] and len(line[0]) < 6000
    full = json.load(open(os.path.join(ROOT, out["full_record"])))
    assert "meaning" in full["roofline_valu"] and full["value"] == pytest.approx(out["value"], rel=1e-5)
    assert len(full["rccl"]["kernel_ms_per_rank"]["all"]) == 2
    # cfg5's own workload timed on every rank of a multi-rank run (a smaller ensemble here: --chains applies to the headline only)
    sh = [c for c in out["configs"] if c["name"] == "vmfmix_k10_kappa500_sharded"]
    assert len(sh) == 1 and sh[0]["ranks_seen"] == 2 and sh[0]["gather_ms"] > 0 and sh[0]["value"] > 0
    # one rank, same flags: the same JSON schema and a comparable per-GPU rate
    one = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--chains", "20000", "--inner", "100", "--thin", "50",
                "--no-cpu-baseline", "--no-ess", "--no-configs"])
    assert one.returncode == 0, one.stderr[-3000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert o1["n_gpus"] == 1 and o1["rccl"]["ranks_seen"] == 1
    assert set(o1) - {"configs"} <= set(out) | {"configs"}


@pytest.mark.gpu
def test_four_ranks_time_cfg5_sharded_and_gather_what_one_process_computes():
    """`bench.py --gpus N` times cfg5's own workload on every rank (vMF mixture K = 10 kappa = 500, chain ids rank * n ..., final
    states gathered; BASELINE.json configs[4], the reference's fan-out scripts/mixture_vMF.py:137-149).  Rehearsed with four gloo
    ranks as separate processes on the one GPU: every rank is seen, and the gathered [d, 4 n] states are, bit for bit, what ONE
    process computes for chains 0 .. 4 n - 1 with the same sequence of launches."""
    import hashlib
    import torch
    import bench
    import geosss_amd as gs
    n, S = 60_000, 200
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "1", "--chains", "20000", "--inner", str(S), "--no-cpu-baseline", "--no-ess"],
             {"GSSS_BENCH_BACKEND": "gloo", "GSSS_BENCH_SHARDED_CHAINS": str(n), "GSSS_BENCH_CHECKSUM": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]                          # the LAST stdout line is the JSON line
    out = json.loads(line)
    assert len(line) < 6000 and out["n_gpus"] == 4 and out["rccl"]["ranks_seen"] == 4
    sh = [c for c in out["configs"] if c["name"] == "vmfmix_k10_kappa500_sharded"][0]
    assert sh["ranks_seen"] == 4 and sh["gather_ms"] > 0 and sh["value"] > 0
    full = json.load(open(os.path.join(ROOT, out["full_record"])))
    rec = [c for c in full["configs"] if c["name"] == "vmfmix_k10_kappa500_sharded"][0]
    assert rec["chains_per_gpu"] == n and rec["chains_in_error"] == 0 and 6.0 < rec["tries_per_step"] < 7.2
    # one process, chains 0 .. 4 n - 1, the launches of time_sharded_config: 100 warm-up transitions, then `launches` x S kept every 100th
    pdf, d = bench.make_target(gs, "vmfmix_k10_kappa500")
    x0 = gs.sample_sphere_device(d - 1, 4 * n, seed=0)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521)
    s.advance(100)
    for _ in range(rec["launches"]):
        s.advance(S, thin=100)
    torch.cuda.synchronize()
    assert hashlib.sha256(s.state_device.contiguous().cpu().numpy().tobytes()).hexdigest() == rec["final_sha256"]

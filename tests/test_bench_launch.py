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
    # imbalance between ranks is visible in the line: per-rank kernel time (min / max / all) and wall clock, the gather's time
    kr = out["rccl"]["kernel_ms_per_rank"]
    assert len(kr["all"]) == 2 and 0 < kr["min"] <= kr["max"] and out["rccl"]["gather_ms"] > 0
    assert 0 < out["rccl"]["wall_ms_per_rank"]["min"] <= out["rccl"]["wall_ms_per_rank"]["max"]
    assert "philox-v2" in out["config"]["stream"] and "meaning" in out["roofline_valu"]
    # one rank, same flags: the same JSON schema and a comparable per-GPU rate
    one = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--chains", "20000", "--inner", "100", "--thin", "50",
                "--no-cpu-baseline", "--no-ess", "--no-configs"])
    assert one.returncode == 0, one.stderr[-3000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert o1["n_gpus"] == 1 and o1["rccl"]["ranks_seen"] == 1
    assert set(o1) - {"configs"} <= set(out) | {"configs"}

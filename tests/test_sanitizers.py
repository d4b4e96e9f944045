"""Sanitizer leg (CPU only; GPU sanitizers are not available on this pool): the C oracle built with
-fsanitize=address,undefined runs its golden-vector checks in a child process with the ASan runtime preloaded."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from oracle import oracle as orc
orc._LIB = {lib!r}
orc._lib = None
from conftest import golden
for name in ("traj_vmfmix_readme", "traj_bingham_d5_dense", "traj_curve_d10_kappa800", "traj_reject_vmfmix_readme"):
    z = golden(name + ".npz")
    tgt = orc.Target.from_fixture(z)
    kind = orc.REJECT if str(z["sampler"]) == "reject" else orc.SHRINK
    n = min(120, len(z["states"]) - 1)
    out = orc.run(tgt, z["x0"], n, sampler=kind, replay=z["draws"][None])
    assert np.max(np.abs(out["samples"][0] - z["states"][1:n + 1])) < 1e-12
    x0 = orc.sample_sphere(1, 50, len(z["x0"]))
    orc.run(tgt, x0, 20, seed=3, n_threads=1)
    orc.run(tgt, x0[:3], 10, numpy_seed=[1, 2, 3])
for name in ("mh_rwmh_vmfmix_readme", "mh_hmc_bingham_d5_dense", "mh_hmc_curve_d10_kappa800"):
    z = golden(name + ".npz")
    tgt = orc.Target.from_fixture(z)
    kind = orc.RWMH if str(z["sampler"]) == "rwmh" else orc.HMC
    out = orc.mh_run(tgt, z["x0"], 100, sampler=kind, stepsize=float(z["stepsize0"]), adapt_steps=int(z["burnin"]),
                     replay=z["draws"][None], trace=True)
    assert np.array_equal(out["accept"][0], z["accept"][:100])
    orc.mh_run(tgt, z["x0"], 50, sampler=kind, numpy_seed=int(z["seed"]))
z = golden("geometry_kat.npz")
for x, a, b in zip(z["d10_slerp_q"][:8], z["d10_slerp_a"], z["d10_slerp_b"]):
    orc.distance_slerp(x, a, b)
print("ASAN-LEG-OK")
"""


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    lib = os.path.join(ROOT, "oracle", "_build", "libgsss_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/libgsss_oracle_asan.so"])
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no libasan runtime next to gcc")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib)], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0 and "ASAN-LEG-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

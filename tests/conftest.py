import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries: build the HIP library (hipcc cross-compiles without a GPU) and
    # the CPU oracle once, before collection imports anything that loads them
    from geosss_amd import build as hip_build
    if not os.path.exists(hip_build.LIB):
        hip_build.build(verbose=False)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def trajectory_names(sampler="shrink"):
    pre = "traj_reject_" if sampler == "reject" else "traj_"
    out = []
    for f in sorted(os.listdir(GOLDEN)):
        if f.startswith(pre) and f.endswith(".npz") and (sampler == "reject" or not f.startswith("traj_reject_")):
            out.append(f[:-4])
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc

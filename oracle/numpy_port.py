"""Per-chain NumPy restatement of the reference's shrinkage sampler loop (geosss/mcmc.py:382-401
with sphere.py:10-33 and distributions.py:156-157, 218-221), structured like the reference: one
chain, one Python-level step at a time, a handful of NumPy/SciPy calls per try.

TEST / BENCH INFRASTRUCTURE ONLY (see oracle/gsss_oracle.c header).  bench.py times it on the GPU
box as the "reference-like" CPU number, because the reference's own files cannot travel there.
tests/test_oracle_golden.py checks it against the golden reference chain.
"""
import numpy as np
from scipy.special import i0, logsumexp


class VmfMixture:
    def __init__(self, mu, weights=None):
        self.mu = np.asarray(mu, dtype=float)
        w = np.ones(len(self.mu)) if weights is None else np.asarray(weights, dtype=float)
        self.logw = np.log(w / w.sum())

    def log_prob(self, x):
        # distributions.py:156-157 per component (norm and i0 re-evaluated per call, as there), :218-221
        p = np.array([x @ m - np.log(2 * np.pi) - np.log(i0(np.linalg.norm(m))) for m in self.mu])
        return logsumexp(p + self.logw, axis=-1)


def spherical_projection(z, v):
    n = v / (np.linalg.norm(v, axis=-1) + 1e-100)   # sphere.py:10-18
    w = z - (z @ n) * n                              # sphere.py:21-26
    return w / (np.linalg.norm(w, axis=-1) + 1e-100)


def shrink_step(pdf, x, rng):
    """One transition; rng needs standard_normal(d), random(), uniform(lo, hi). Returns (y, n_reject)."""
    u = spherical_projection(rng.standard_normal(len(x)), x)
    threshold = pdf.log_prob(x) + np.log(rng.random())
    theta = rng.uniform(0, 2 * np.pi)
    lo, hi = theta - 2 * np.pi, theta
    rej = 0
    while True:
        theta = rng.uniform(lo, hi)
        y = np.cos(theta) * x + np.sin(theta) * u
        if pdf.log_prob(y) > threshold:
            return y, rej
        if theta < 0:
            lo = theta
        else:
            hi = theta
        rej += 1


def run_chain(pdf, x0, n_steps, seed):
    rng = np.random.default_rng(seed)
    x = np.array(x0, dtype=float)
    out = np.empty((n_steps, len(x)))
    rej = 0
    for i in range(n_steps):
        x, r = shrink_step(pdf, x, rng)
        out[i] = x
        rej += r
    return out, rej


def _worker(args):
    mu, x0, n_steps, seed = args
    return run_chain(VmfMixture(mu), x0, n_steps, seed)[1]


def time_chains(mu, x0, n_steps, n_procs):
    """steps/s of n_procs independent chains, one per process (the reference's own scale-out)."""
    import time
    from concurrent.futures import ProcessPoolExecutor
    jobs = [(mu, x0, n_steps, 100 + i) for i in range(n_procs)]
    t0 = time.perf_counter()
    if n_procs == 1:
        _worker(jobs[0])
    else:
        with ProcessPoolExecutor(n_procs) as ex:
            list(ex.map(_worker, jobs))
    return n_procs * n_steps / (time.perf_counter() - t0)

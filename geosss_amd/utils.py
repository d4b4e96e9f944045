"""Launcher + call counter with the reference's names (geosss/utils.py:137-232): the two geodesic slice samplers
and the two baselines the paper compares them with (RWMH, spherical HMC)."""
import contextlib
import logging
import time

from .distributions import counted
from .mcmc import MetropolisHastings, RejectionSphericalSliceSampler, ShrinkageSphericalSliceSampler, SphericalHMC

count_calls = counted

# the reference's plot palette and overflow-safe elementary functions (utils.py:15-88), for scripts that import them
colors = [(0.85, 0.3, 0.1), (0.15, 0.35, 0.6), (0.95, 0.7, 0.1), (0.0, 0.0, 0.0), (0.8, 0.8, 0.8)]
EXP_MIN, EXP_MAX = -308, 709
LOG_MIN, LOG_MAX = 1e-308, 1e308


def format_time(t):
    """Seconds -> "12.3 ms" (utils.py:31-37)."""
    scale, unit = next(((s, u) for s, u in ((1.0, "s"), (1e-3, "ms"), (1e-6, "us"), (1e-9, "ns")) if t > s or t == 0), (1e-9, "ns"))
    return "%.1f %s" % (t / scale, unit)


def exp(x, x_min=EXP_MIN, x_max=EXP_MAX):
    """exp of the argument clipped into [max(x_min, -308), min(x_max, 709)]: never overflows (utils.py:49-67)."""
    import numpy as np
    return np.exp(np.clip(x, max(x_min, EXP_MIN), min(x_max, EXP_MAX)))


def log(x, x_min=LOG_MIN, x_max=LOG_MAX):
    """log of the argument clipped into [max(x_min, 1e-308), min(x_max, 1e308)] (utils.py:70-83)."""
    import numpy as np
    return np.log(np.clip(x, max(x_min, LOG_MIN), min(x_max, LOG_MAX)))


def relative_entropy(p, q):
    """Kullback-Leibler divergence sum_i p_i (log p_i - log q_i) with the clipped log (utils.py:86-88)."""
    return p @ (log(p) - log(q))


@contextlib.contextmanager
def take_time(desc, mute=False):
    """`with take_time("sss-shrink"): ...` as the reference's scripts time their sampler runs (utils.py:42-48, e.g.
    scripts/curve_vMF.py:103).  The reference logs CPU process time; the work here happens on the device, so the block is
    bracketed by device synchronisation and the WALL time is logged (same message format, `logging.info`)."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    yield
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not mute:
        logging.info("%s took %s", desc, format_time(dt))


def counter(method_names):
    """Class decorator: wrap the named methods with the call counter (utils.py:162-185)."""
    names = [method_names] if isinstance(method_names, str) else list(method_names)

    def decorate(cls):
        for name in names:
            if hasattr(cls, name):
                setattr(cls, name, counted(getattr(cls, name)))
        return cls

    return decorate


class SamplerLauncher:
    """`SamplerLauncher(pdf, initial, n_samples, burnin=0.2, seed=None).run("sss-shrink")`
    (utils.py:188-232).  Keeps the sampler objects as `.ssss` / `.rsss` like the reference so
    that callers can read `.n_reject` afterwards (scripts/curve_vMF.py:105-112)."""

    def __init__(self, pdf, initial, n_samples, burnin=0.2, seed=None, **sampler_kwargs):
        self.pdf = pdf
        self.initial = initial
        self.n_samples = n_samples
        self.burnin = burnin
        self.seed = seed
        self.sampler_kwargs = sampler_kwargs

    def run_sss_reject(self):
        self.rsss = RejectionSphericalSliceSampler(self.pdf, self.initial, self.seed, **self.sampler_kwargs)
        return self.rsss.sample(self.n_samples, burnin=self.burnin)

    def run_sss_shrink(self):
        self.ssss = ShrinkageSphericalSliceSampler(self.pdf, self.initial, self.seed, **self.sampler_kwargs)
        return self.ssss.sample(self.n_samples, burnin=self.burnin)

    def run_rwmh(self):
        kw = {k: v for k, v in self.sampler_kwargs.items() if k not in ("mode", "screen", "placement")}
        self.rwmh = MetropolisHastings(self.pdf, self.initial, self.seed, stepsize=1e-1, **kw)  # utils.py:210-214
        return self.rwmh.sample(self.n_samples, burnin=self.burnin)

    def run_hmc(self):
        kw = {k: v for k, v in self.sampler_kwargs.items() if k not in ("mode", "screen", "placement")}
        self.hmc = SphericalHMC(self.pdf, self.initial, self.seed, stepsize=1e-1, **kw)          # utils.py:216-220
        return self.hmc.sample(self.n_samples, burnin=self.burnin)

    def run(self, method):
        if method == "sss-reject":
            return self.run_sss_reject()
        if method == "sss-shrink":
            return self.run_sss_shrink()
        if method == "rwmh":
            return self.run_rwmh()
        if method == "hmc":
            return self.run_hmc()
        raise ValueError(f"method {method} not known")  # utils.py:232

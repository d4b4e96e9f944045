"""Launcher + call counter with the reference's names (geosss/utils.py:137-232): the two geodesic slice samplers
and the two baselines the paper compares them with (RWMH, spherical HMC)."""
import contextlib
import logging
import time

from .distributions import counted
from .mcmc import MetropolisHastings, RejectionSphericalSliceSampler, ShrinkageSphericalSliceSampler, SphericalHMC

count_calls = counted


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
        scale, unit = next(((s, u) for s, u in ((1.0, "s"), (1e-3, "ms"), (1e-6, "us"), (1e-9, "ns")) if dt > s or dt == 0), (1e-9, "ns"))
        logging.info("%s took %.1f %s", desc, dt / scale, unit)


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

"""ctypes front-end of the CPU oracle (oracle/gsss_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (geosss_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np
from scipy.special import i0, ive

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libgsss_oracle.so")

VMF_MIXTURE, BINGHAM, CURVE_VMF, CPD = 1, 2, 3, 4
SHRINK, REJECT, RWMH, HMC, INDEP, MIX = 0, 1, 2, 3, 4, 5
ERR_MAX_TRIES, ERR_NONFINITE, ERR_REPLAY_EXHAUSTED = 1, 2, 4


def build(force=False):
    src = os.path.join(_HERE, "gsss_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class _Target(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("k", C.c_int32),
                ("mu", C.c_void_p), ("lognorm", C.c_void_p), ("logw", C.c_void_p),
                ("A", C.c_void_p), ("b", C.c_void_p), ("knots", C.c_void_p), ("kappa", C.c_double),
                ("src", C.c_void_p), ("src_w", C.c_void_p), ("tgt", C.c_void_p), ("tgt_w", C.c_void_p),
                ("n_target", C.c_int32), ("target_dim", C.c_int32), ("k_nn", C.c_int32), ("outlier", C.c_int32),
                ("sigma", C.c_double), ("beta", C.c_double), ("omega", C.c_double), ("log_volume", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.gor_logprob.restype = C.c_double
        _lib.gor_distance_slerp.restype = C.c_double
        _lib.gor_distance.restype = C.c_double
        _lib.gor_logsumexp.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def log_i0(kappa):
    """log(i0(kappa)) exactly as geosss/distributions.py:157 evaluates it (overflows to inf
    at kappa >~ 713.99, which the reference inherits)."""
    with np.errstate(over="ignore"):
        return np.log(i0(kappa))


class Target:
    """Plain-array description of a target + the C struct the oracle reads."""

    def __init__(self, kind, d, k=0, mu=None, lognorm=None, logw=None, A=None, knots=None, kappa=0.0, b=None):
        self.kind, self.d, self.k = kind, int(d), int(k)
        self.mu = _f64(mu) if mu is not None else None
        self.lognorm = _f64(lognorm) if lognorm is not None else None
        self.logw = _f64(logw) if logw is not None else None
        self.A = _f64(A) if A is not None else None
        self.knots = _f64(knots) if knots is not None else None
        self.b = _f64(b) if b is not None else None
        self.kappa = float(kappa)
        self.c = _Target(kind, self.d, self.k, _p(self.mu), _p(self.lognorm), _p(self.logw), _p(self.A),
                         _p(self.b), _p(self.knots), self.kappa)

    @classmethod
    def vmf_mixture(cls, mu, weights=None):
        mu = _f64(mu)
        k, d = mu.shape
        w = np.ones(k) if weights is None else np.array(weights, dtype=float)
        w = w / w.sum()                                            # distributions.py:213-216
        lognorm = np.log(2 * np.pi) + log_i0(np.linalg.norm(mu, axis=1))  # distributions.py:157
        with np.errstate(divide="ignore"):                          # (a zero weight is log 0 = -inf on purpose, as in the reference)
            logw = np.log(w)
        return cls(VMF_MIXTURE, d, k, mu=mu, lognorm=lognorm, logw=logw)

    @classmethod
    def bingham(cls, A, b=None):
        A = _f64(A)
        return cls(BINGHAM, A.shape[0], A=A, b=b)

    @classmethod
    def curve_vmf(cls, knots, kappa):
        knots = _f64(knots)
        return cls(CURVE_VMF, knots.shape[1], knots.shape[0], knots=knots, kappa=kappa)

    @classmethod
    def cpd(cls, source, source_w, target, target_w, sigma, k_nn, beta=1.0, omega=0.0, outlier=True):
        """CoherentPointDrift / GaussianMixtureModel of geosss/registration.py on unit quaternions."""
        source, target = _f64(source), _f64(target)
        t = cls(CPD, 4, len(source))
        t.src, t.src_w, t.tgt, t.tgt_w = source, _f64(source_w), target, _f64(target_w)
        log_volume = float(np.sum(np.log(np.ptp(target, 0))))                   # registration.py:207-213
        t.c = _Target(CPD, 4, len(source), None, None, None, None, None, None, 0.0, _p(t.src), _p(t.src_w), _p(t.tgt),
                      _p(t.tgt_w), len(target), target.shape[1], int(k_nn), int(bool(outlier)), float(sigma), float(beta),
                      float(omega), log_volume)
        return t

    @classmethod
    def from_fixture(cls, z, prefix="target_"):
        kind = str(z[prefix + "kind"])
        if kind == "cpd":
            return cls.cpd(z[prefix + "source"], z[prefix + "source_w"], z[prefix + "target"], z[prefix + "target_w"],
                           float(z[prefix + "sigma"]), int(z[prefix + "k_nn"]), float(z[prefix + "beta"]),
                           float(z[prefix + "omega"]), bool(z[prefix + "outlier"]))
        if kind == "vmf_mixture":
            return cls.vmf_mixture(z[prefix + "mu"], z[prefix + "weights"])
        if kind == "bingham":
            return cls.bingham(z[prefix + "A"], z[prefix + "b"] if prefix + "b" in z.files else None)
        if kind == "curve_vmf":
            return cls.curve_vmf(z[prefix + "knots"], float(z[prefix + "kappa"]))
        raise ValueError(kind)

    def log_prob(self, X):
        X = _f64(X)
        if X.ndim == 1:
            return float(lib().gor_logprob(C.byref(self.c), _p(X)))
        out = np.empty(len(X))
        lib().gor_logprob_batch(C.byref(self.c), _p(X), C.c_int64(len(X)), _p(out))
        return out


def pcg64_words(seed):
    """(state_hi, state_lo, inc_hi, inc_lo) of np.random.default_rng(seed)'s PCG64, one row per seed."""
    seeds = seed if isinstance(seed, (list, tuple)) else [seed]
    out = np.empty((len(seeds), 4), dtype=np.uint64)
    for i, sd in enumerate(seeds):
        st = np.random.default_rng(sd).bit_generator.state["state"]
        m = (1 << 64) - 1
        out[i] = [st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m]
    return out


def npy_fill(pcg_words, n_normal, n_uniform):
    """Draw n_normal standard normals then n_uniform uniforms from the restated numpy stream."""
    w = np.array(pcg_words, dtype=np.uint64).reshape(4).copy()
    z, u = np.empty(n_normal), np.empty(n_uniform)
    lib().gor_npy_fill(_p(w), C.c_int64(n_normal), _p(z), C.c_int64(n_uniform), _p(u))
    return z, u, w


def run(target, state, n_steps, seed=0, chain_offset=0, step_offset=0, sampler=SHRINK, thin=1, max_tries=100000,
        keep_samples=True, replay=None, trace_threshold=False, n_threads=1, numpy_seed=None):
    """Advance every row of `state` (n_chains, d) by n_steps transitions.

    Returns dict(state, samples (n_chains, n_keep, d) | None, n_reject, n_tries, err, threshold).
    """
    state = np.array(state, dtype=np.float64, order="C", copy=True)
    single = state.ndim == 1
    if single:
        state = state[None]
    n, d = state.shape
    assert d == target.d
    n_keep = n_steps // thin
    samples = np.empty((n, n_keep, d)) if keep_samples else None
    n_reject = np.zeros(n, dtype=np.int64)
    n_tries = np.zeros(n, dtype=np.int64)
    err = np.zeros(n, dtype=np.int32)
    thr = np.full((n, n_steps), np.nan) if trace_threshold else None
    stride = 0
    if replay is not None:
        replay = _f64(replay)
        if replay.ndim == 1:
            replay = replay[None]
        assert replay.shape[0] == n
        stride = replay.shape[1]
    pcg = None
    if numpy_seed is not None:  # numpy's own stream: one default_rng(seed) per chain
        pcg = pcg64_words(numpy_seed if isinstance(numpy_seed, (list, tuple)) else [numpy_seed])
        assert len(pcg) == n
    lib().gor_run(C.byref(target.c), _p(state), C.c_int64(n), C.c_int64(n_steps), C.c_int64(thin),
                  C.c_uint64(seed), C.c_uint64(chain_offset), C.c_uint64(step_offset), C.c_int(sampler),
                  C.c_int64(max_tries), _p(samples), _p(n_reject), _p(n_tries), _p(err), _p(replay),
                  C.c_int64(stride), _p(thr), C.c_int(n_threads), _p(pcg))
    return dict(state=state[0] if single else state, samples=samples, n_reject=n_reject, n_tries=n_tries, err=err,
                threshold=thr, pcg=pcg)


def gradient(target, x):
    x = _f64(x)
    g = np.empty_like(x)
    lib().gor_gradient(C.byref(target.c), _p(x), _p(g))
    return g


def npy_gamma(pcg_words, shape, n):
    """n variates of numpy's Generator.gamma(shape) from the restated stream."""
    w = np.array(pcg_words, dtype=np.uint64).reshape(4).copy()
    out = np.empty(n)
    lib().gor_npy_gamma_fill(_p(w), C.c_double(shape), C.c_int64(n), _p(out))
    return out, w


def mh_run(target, state, n_steps, sampler=RWMH, stepsize=0.1, adapt_steps=0, n_leapfrog=10, seed=0, chain_offset=0,
           step_offset=0, thin=1, keep_samples=True, replay=None, n_threads=1, numpy_seed=None, momenta=None,
           trace=False, mixing_probability=0.5, adapt_left=None):
    """MetropolisHastings / SphericalHMC (geosss/mcmc.py:118-332) for every row of `state`.
    Returns dict(state, momenta, samples, n_accept, stepsize, err, accept (trace), stepsize_trace, pcg)."""
    state = np.array(state, dtype=np.float64, order="C", copy=True)
    single = state.ndim == 1
    if single:
        state = state[None]
    n, d = state.shape
    assert d == target.d
    n_keep = n_steps // thin
    samples = np.empty((n, n_keep, d)) if keep_samples else None
    n_accept = np.zeros(n, dtype=np.int64)
    err = np.zeros(n, dtype=np.int32)
    eps = np.broadcast_to(np.asarray(stepsize, dtype=np.float64), (n,)).copy()
    mom = np.zeros((n, d)) if momenta is None else np.array(momenta, dtype=np.float64, order="C", copy=True).reshape(n, d)
    stride = 0
    if replay is not None:
        replay = _f64(replay)
        if replay.ndim == 1:
            replay = replay[None]
        assert replay.shape[0] == n
        stride = replay.shape[1]
    pcg = None
    if numpy_seed is not None:
        pcg = pcg64_words(numpy_seed if isinstance(numpy_seed, (list, tuple)) else [numpy_seed])
        assert len(pcg) == n
    acc = np.zeros((n, n_steps), dtype=np.uint8) if trace else None
    eps_tr = np.zeros((n, n_steps)) if trace else None
    prop = np.zeros((n, n_steps), dtype=np.uint8) if trace else None
    left = np.broadcast_to(np.asarray(adapt_steps if adapt_left is None else adapt_left, dtype=np.int64), (n,)).copy()
    n_rwmh = np.zeros(n, dtype=np.int64)
    lib().gor_mh_run(C.byref(target.c), _p(state), _p(mom), C.c_int64(n), C.c_int64(n_steps), C.c_int64(thin),
                     C.c_uint64(seed), C.c_uint64(chain_offset), C.c_uint64(step_offset), C.c_int(sampler), _p(eps),
                     C.c_int64(adapt_steps), C.c_int(n_leapfrog), _p(samples), _p(n_accept), _p(err), _p(replay),
                     C.c_int64(stride), C.c_int(n_threads), _p(pcg), _p(acc), _p(eps_tr), C.c_double(mixing_probability),
                     _p(left), _p(n_rwmh), _p(prop))
    return dict(state=state[0] if single else state, momenta=mom[0] if single else mom, samples=samples,
                n_accept=n_accept, stepsize=eps, err=err, accept=acc, stepsize_trace=eps_tr, pcg=pcg,
                adapt_left=left, n_rwmh=n_rwmh, use_rwmh=prop)


def sample_sphere(seed, n, d, chain_offset=0):
    out = np.empty((n, d))
    lib().gor_sample_sphere(C.c_uint64(seed), C.c_uint64(chain_offset), C.c_int64(n), C.c_int(d), _p(out))
    return out


def stream_block(seed, chain, step, blk):
    u = np.empty(2)
    lib().gor_stream_block(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(step), C.c_uint32(blk), _p(u))
    return u


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().gor_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def radial_projection(x):
    x = _f64(x)
    out = np.empty_like(x)
    lib().gor_radial_projection(_p(x), C.c_int(len(x)), _p(out))
    return out


def orthogonal_projection(x, y):
    x, y = _f64(x), _f64(y)
    out = np.empty_like(x)
    lib().gor_orthogonal_projection(_p(x), _p(y), C.c_int(len(x)), _p(out))
    return out


def spherical_projection(x, v):
    x, v = _f64(x), _f64(v)
    out = np.empty_like(x)
    lib().gor_spherical_projection(_p(x), _p(v), C.c_int(len(x)), _p(out))
    return out


def distance_slerp(x, a, b):
    x, a, b = _f64(x), _f64(a), _f64(b)
    y = np.empty_like(x)
    dist = lib().gor_distance_slerp(_p(x), _p(a), _p(b), C.c_int(len(x)), _p(y))
    return dist, y


def find_nearest(knots, x):
    knots, x = _f64(knots), _f64(x)
    out = np.empty_like(x)
    lib().gor_find_nearest(_p(knots), C.c_int(knots.shape[0]), C.c_int(knots.shape[1]), _p(x), _p(out))
    return out


def logsumexp(a):
    a = _f64(a)
    return float(lib().gor_logsumexp(_p(a), C.c_int(len(a))))

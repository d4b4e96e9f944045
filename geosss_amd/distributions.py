"""Target distributions of the geodesic slice sampler, API-compatible with
geosss/distributions.py for the families the hot path covers:

    VonMisesFisher(mu)                      distributions.py:117-160
    MixtureModel(components, weights=None)  distributions.py:209-227
    Bingham(A), random_bingham(...)         distributions.py:36-103, 230-258
    CurvedVonMisesFisher(curve, kappa)      distributions.py:261-278
    SlerpCurve(knots), brownian_curve(...)  spherical_curve.py:74-129

The objects are parameter carriers: they keep the same attributes as the reference
(`.mu`, `.pdfs`, `.weights`, `.A`, `.curve.knots`, `.kappa`, `.d`) and hand a packed
parameter block to the HIP library.  `log_prob` accepts a point (d,) or rows (n, d), as in
the reference, and is evaluated ON THE GPU through the C ABI (`gsss_logprob`); there is no
host implementation to fall back to.  `pdf.log_prob.num_calls` / `.reset_counters()` keep
the reference's call-counter protocol (utils.py:137-185); the samplers add the number of
log-density evaluations their kernels performed.
"""
import ctypes as C

import numpy as np
import torch
from scipy.special import i0, ive, iv, logsumexp

from . import _lib
from .sphere import _device_index, current_stream_ptr

__all__ = ["Distribution", "VonMisesFisher", "MixtureModel", "Bingham", "BinghamFisher", "CurvedVonMisesFisher", "SlerpCurve",
           "random_bingham", "brownian_curve", "counted"]


def counted(fn):
    """Call counter with the reference's protocol: `obj.method.num_calls`, `obj.method.reset_counters()`
    (the attributes live on the underlying function, so they are shared by all instances of the
    class, exactly like geosss.utils.count_calls)."""

    def method(self, *args, **kwargs):
        method.num_calls += 1
        return fn(self, *args, **kwargs)

    def reset_counters():
        method.num_calls = 0

    method.num_calls = 0
    method.reset_counters = reset_counters
    method.__name__ = fn.__name__
    method.__doc__ = fn.__doc__
    return method


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def log_bessel_i0(kappa):
    """log(i0(kappa)).  The reference evaluates np.log(i0(kappa)) (distributions.py:157), which
    overflows to +inf for kappa >~ 713.99 and then never terminates (mcmc.py:394); below that
    threshold we return the very same expression, above it the overflow-free equivalent
    log(ive(0, kappa)) + kappa."""
    kappa = np.asarray(kappa, dtype=np.float64)
    with np.errstate(over="ignore"):
        direct = np.log(i0(kappa))
    safe = np.log(ive(0, kappa)) + kappa
    return np.where(np.isfinite(direct), direct, safe)


class _DeviceTarget:
    """Owns one gsss_target handle (device parameter block)."""

    def __init__(self, desc_arrays, kind, d, k, kappa, device, extra=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = device
        self._keep = (desc_arrays, extra)  # keep the host arrays alive during create
        desc = _lib.TargetDesc(kind, d, k, 0,
                               *[a.ctypes.data_as(C.c_void_p) if a is not None else None for a in desc_arrays],
                               float(kappa))
        for name, value in (extra or {}).items():  # registration targets: the further fields of gsss_target_desc
            setattr(desc, name, value.ctypes.data_as(C.c_void_p) if isinstance(value, np.ndarray) else value)
        h = C.c_void_p()
        _lib.check(self.lib.gsss_target_create(C.byref(desc), device, C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.gsss_target_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Distribution:
    """Base class: packing + device evaluation shared by all targets."""

    def _pack(self):
        """-> (kind, d, k, kappa, (mu, logc, A, knots))"""
        raise TypeError(f"{type(self).__name__} has no device parameter block: the samplers run inside HIP kernels and cannot call a "
                        "Python log_prob; built-in targets are VonMisesFisher, MixtureModel (of vMF components), Bingham, "
                        "BinghamFisher, CurvedVonMisesFisher, Uniform, CoherentPointDrift and GaussianMixtureModel")

    def _device_target(self, device=None):
        dev = _device_index(device)
        packed = self._pack()
        kind, d, k, kappa, arrays = packed[:5]
        extra = packed[5] if len(packed) > 5 else None
        # the parameter attributes are public and mutable, as in the reference: key the device copy on
        # their current bytes so that an edited target is re-uploaded instead of silently reused
        key = (kind, d, k, kappa) + tuple(a.tobytes() if a is not None else None for a in arrays)
        if extra:
            key += tuple(v.tobytes() if isinstance(v, np.ndarray) else v for v in extra.values())
        cache = self.__dict__.setdefault("_targets", {})
        hit = cache.get(dev)
        if hit is None or hit[0] != key:
            cache[dev] = (key, _DeviceTarget(arrays, kind, d, k, kappa, dev, extra))
        return cache[dev][1]

    def _invalidate(self):
        self.__dict__.pop("_targets", None)

    def _log_prob_device(self, x):
        """x: numpy (d,), (n, d) or a CUDA float64 tensor (n, d) -> same kind of container."""
        if isinstance(x, torch.Tensor):
            if not x.is_cuda:
                raise ValueError("torch input to log_prob must live on the GPU")
            tgt = self._device_target(x.device)
            xt = x.to(torch.float64).contiguous()
            single = xt.ndim == 1
            xt2 = xt[None] if single else xt
            if xt2.ndim != 2 or xt2.shape[1] != self.d:
                raise ValueError(f"expected (..., {self.d}) input")
            out = torch.empty(xt2.shape[0], dtype=torch.float64, device=xt.device)
            dev = _device_index(xt.device)
            _lib.check(tgt.lib.gsss_logprob(tgt.handle, xt2.data_ptr(), xt2.shape[0], out.data_ptr(),
                                            current_stream_ptr(dev)))
            return out[0] if single else out
        x = np.asarray(x, dtype=np.float64)
        if x.ndim not in (1, 2) or x.shape[-1] != self.d:
            raise ValueError(f"expected (d,) or (n, d) input with d={self.d}")  # distributions.py:84
        _lib.require_device()
        dev = _device_index(None)
        xt = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(x))).to(f"cuda:{dev}")
        out = self._log_prob_device(xt).cpu().numpy()
        return float(out[0]) if x.ndim == 1 else out

    def _gradient_device(self, x):
        """Distribution.gradient through gsss_gradient (the functions the spherical HMC kernel evaluates): numpy (d,) or (n, d),
        or a CUDA float64 tensor (n, d) -> the same kind of container."""
        if isinstance(x, torch.Tensor):
            if not x.is_cuda:
                raise ValueError("torch input to gradient must live on the GPU")
            tgt = self._device_target(x.device)
            xt = x.to(torch.float64).contiguous()
            single = xt.ndim == 1
            xt2 = xt[None] if single else xt
            if xt2.ndim != 2 or xt2.shape[1] != self.d:
                raise ValueError(f"expected (..., {self.d}) input")
            out = torch.empty_like(xt2)
            dev = _device_index(xt.device)
            _lib.check(tgt.lib.gsss_gradient(tgt.handle, xt2.data_ptr(), xt2.shape[0], out.data_ptr(), current_stream_ptr(dev)))
            return out[0] if single else out
        x = np.asarray(x, dtype=np.float64)
        if x.ndim not in (1, 2) or x.shape[-1] != self.d:
            raise ValueError(f"expected (d,) or (n, d) input with d={self.d}")
        _lib.require_device()
        dev = _device_index(None)
        out = self._gradient_device(torch.from_numpy(np.ascontiguousarray(np.atleast_2d(x))).to(f"cuda:{dev}")).cpu().numpy()
        return out[0] if x.ndim == 1 else out

    def log_prob(self, x):
        raise NotImplementedError

    def gradient(self, x):
        """The gradient of log_prob in the ambient space, as the reference's classes define it; evaluated on the device."""
        return self._gradient_device(x)


class VonMisesFisher(Distribution):
    """vMF(x) ~ exp(mu.x);  log_prob = x.mu - log(2 pi) - log(i0(|mu|))  (distributions.py:156-157).
    `mu` is stored unnormalised, kappa = |mu| (distributions.py:126-136)."""

    def __init__(self, mu):
        self.mu = np.array(mu, dtype=np.float64)

    @property
    def d(self):
        return self.mu.size

    @property
    def kappa(self):
        return float(np.linalg.norm(self.mu))

    @property
    def mode(self):
        return self.mu / (np.linalg.norm(self.mu) + 1e-100)

    @property
    def max_log_prob(self):
        return self.kappa

    def _log_const(self):
        return -np.log(2 * np.pi) - float(log_bessel_i0(self.kappa))

    def _pack(self):
        mu = _as_f64(self.mu[None])
        logc = _as_f64([self._log_const()])
        return _lib.VMF_MIXTURE, self.d, 1, 0.0, (mu, logc, None, None)

    @counted
    def log_prob(self, x):
        return self._log_prob_device(x)

    @counted
    def gradient(self, x):
        return self.mu                                  # distributions.py:159-160 (whatever the shape of x)


class _HostDensity(Distribution):
    """Densities the reference defines beside its sampler targets (plot overlays, envelopes): evaluated on the host with
    numpy, as the reference does; they have no device parameter block and are refused as targets of the HIP samplers."""

    def _pack(self):
        raise TypeError(f"{type(self).__name__} is a host-side density (plotting / envelope); the HIP samplers run on "
                        "VonMisesFisher, MixtureModel, Bingham, BinghamFisher, CurvedVonMisesFisher, Uniform and the "
                        "registration targets")


class MarginalVonMisesFisher(_HostDensity, VonMisesFisher):
    """The density on [-1, 1] of coordinate `dim_idx` of a vMF(mu) variate (distributions.py:164-186): what the
    reference's diagnostics overlay on the histograms of the chains' coordinates."""

    def __init__(self, dim_idx, mu):
        VonMisesFisher.__init__(self, mu)
        self.dim_idx = dim_idx

    def prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        d, kappa = self.d, self.kappa
        m = self.mu[self.dim_idx] / kappa                        # the mean direction's own coordinate
        rest = (1.0 - m * m) * (1.0 - x * x)
        return (np.sqrt(kappa / (2.0 * np.pi)) / iv(0.5 * d - 1.0, kappa) * ((1.0 - x * x) / (1.0 - m * m)) ** (0.25 * (d - 3))
                * np.exp(kappa * m * x) * iv(0.5 * (d - 3), kappa * np.sqrt(rest)))

    @counted
    def log_prob(self, x):
        return np.log(np.clip(self.prob(x), 1e-308, None))


class MultivariateNormal(_HostDensity):
    """-(x - mu)^T C^-1 (x - mu) / 2 (distributions.py:189-197)."""

    def __init__(self, mu, C):
        self.mu = np.asarray(mu, dtype=np.float64)
        self.C = np.asarray(C, dtype=np.float64)
        self.invC = np.linalg.inv(self.C)

    @property
    def d(self):
        return len(self.C)

    def _quad(self, x):
        r = np.asarray(x, dtype=np.float64) - self.mu
        return np.sum(r * (r @ self.invC), axis=-1)

    @counted
    def log_prob(self, x):
        return -0.5 * self._quad(x)


class ACG(MultivariateNormal):
    """Angular central Gaussian: -(d / 2) log(x^T C^-1 x) (distributions.py:200-207); the envelope of rand.sample_bingham."""

    def __init__(self, C):
        super().__init__(np.zeros(len(C)), C)

    @counted
    def log_prob(self, x):
        return -0.5 * self.d * np.log(self._quad(x))


class MixtureModel(Distribution):
    """log_prob = logsumexp_k(log_prob_k(x) + log w_k), weights normalised to one
    (distributions.py:211-221).  Components must be VonMisesFisher of one dimension."""

    def __init__(self, components, weights=None):
        self.pdfs = list(components)
        if not self.pdfs or not all(isinstance(p, VonMisesFisher) for p in self.pdfs):
            raise TypeError("the HIP path covers mixtures of VonMisesFisher components")
        # a mixture of coordinate marginals (the reference's histogram overlay, scripts/vMF_diagnostics.py:106-108) is a
        # density on [-1, 1]: host arithmetic, never a sampler target
        self._marginal = all(isinstance(p, MarginalVonMisesFisher) for p in self.pdfs)
        if not self._marginal and any(isinstance(p, MarginalVonMisesFisher) for p in self.pdfs):
            raise TypeError("cannot mix coordinate marginals with densities on the sphere")
        if len({p.d for p in self.pdfs}) != 1:
            raise ValueError("all components must share the dimension")
        w = np.ones(len(self.pdfs)) if weights is None else np.array(weights, dtype=np.float64)
        if w.shape != (len(self.pdfs),):
            raise ValueError("one weight per component")
        self.weights = w / w.sum()

    @property
    def d(self):
        return self.pdfs[0].d

    def _pack(self):
        if self._marginal:
            return _HostDensity._pack(self)
        mu = _as_f64([p.mu for p in self.pdfs])
        with np.errstate(divide="ignore"):
            logw = np.log(self.weights)
        logc = _as_f64([p._log_const() for p in self.pdfs]) + logw
        return _lib.VMF_MIXTURE, self.d, len(self.pdfs), 0.0, (mu, _as_f64(logc), None, None)

    @counted
    def log_prob(self, x):
        if self._marginal:
            with np.errstate(divide="ignore"):
                return logsumexp(np.stack([p.log_prob(x) for p in self.pdfs], axis=-1) + np.log(self.weights), axis=-1)
        return self._log_prob_device(x)

    @counted
    def gradient(self, x):
        """The softmax-weighted mean of the components' gradients (distributions.py:223-227), on the device."""
        if self._marginal:
            return _HostDensity._pack(self)
        return self._gradient_device(x)


class Bingham(Distribution):
    """p(x) ~ exp(x^T A x) with symmetric A;  log_prob = sum((x @ A) * x)  (distributions.py:67-86)."""

    def __init__(self, A):
        A = np.array(A, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1] or not np.allclose(A, A.T):
            raise ValueError("A must be a symmetric square matrix")  # distributions.py:68
        self.A = A
        v, U = np.linalg.eigh(A)
        self.v, self.U = v[::-1], U[:, ::-1]  # descending, as the reference keeps them

    @property
    def d(self):
        return len(self.A)

    @property
    def mode(self):
        return self.U[:, 0]

    @property
    def max_log_prob(self):
        return self.v[0]

    def _pack(self):
        return _lib.BINGHAM, self.d, 0, 0.0, (None, None, _as_f64(self.A), None)

    @counted
    def log_prob(self, x):
        return self._log_prob_device(x)

    @counted
    def gradient(self, x):
        """2 A x (distributions.py:88-89); rows of points and device tensors go through the kernel's own evaluation."""
        if isinstance(x, torch.Tensor) or np.ndim(x) == 2:
            return self._gradient_device(x)
        return 2 * self.A @ np.asarray(x, dtype=np.float64)


class BinghamFisher(Bingham):
    """Fisher-Bingham: log_prob = x^T A x + x.b  (distributions.py:106-114)."""

    def __init__(self, A, b):
        super().__init__(A)
        b = np.array(b, dtype=np.float64)
        if b.shape != (len(self.A),):
            raise ValueError("b must have one entry per dimension")  # distributions.py:109
        self.b = b

    def _pack(self):
        return _lib.BINGHAM, self.d, 0, 0.0, (_as_f64(self.b), None, _as_f64(self.A), None)

    @counted
    def log_prob(self, x):
        return self._log_prob_device(x)

    # gradient: inherited from Bingham, 2 A x WITHOUT the linear term -- the reference's class does not override it
    # (distributions.py:106-114), and its spherical HMC therefore runs with that gradient; so does the kernel here.


class Uniform(Bingham):
    """The uniform distribution on the sphere (distributions.py:28-34): log_prob = 0.  The reference's class carries no
    dimension; to run a sampler on it give the ambient dimension, Uniform(d) -- on the device it is the Bingham target
    with A = 0."""

    def __init__(self, d=None):
        self._d = None if d is None else int(d)
        if d is not None:
            super().__init__(np.zeros((self._d, self._d)))

    @property
    def d(self):
        return self._d

    def _pack(self):
        if self._d is None:
            raise TypeError("Uniform() has no dimension: construct it as Uniform(d) to sample it on the device")
        return super()._pack()

    def log_prob(self, x):
        return 0.0

    def gradient(self, x):
        return np.zeros_like(x)


def random_bingham(d=2, vmax=None, vmin=None, eigensystem=False, seed=None):
    """Random Bingham target with the construction of distributions.py:230-258, so that the same
    seed gives the same precision matrix as the reference (e.g. scripts/bingham.py:131 uses
    d=10, vmax=30, vmin=0, eigensystem=True, seed=6982)."""
    g = np.random.default_rng(seed)
    M = g.standard_normal((d, d))
    spectrum, basis = np.linalg.eigh(M.T @ M)
    if vmin is not None:
        spectrum += vmin - spectrum.min()
    if vmax is not None:
        spectrum *= vmax / spectrum.max()
    if eigensystem:
        basis = np.eye(d)
    return Bingham((basis * spectrum) @ basis.T)


class SlerpCurve:
    """Piecewise-geodesic curve through `knots` (rows, unit vectors); spherical_curve.py:74-85."""

    def __init__(self, knots):
        self.knots = np.array(knots, dtype=np.float64)
        if self.knots.ndim != 2 or self.knots.shape[0] < 2:
            raise ValueError("need at least two knots")
        # arc lengths and the normalised arc-length parameter of the knots (spherical_curve.py:80-85)
        seg = np.arccos(np.clip(np.sum(self.knots[:-1] * self.knots[1:], axis=-1), -1, 1))
        self.theta = np.append(0.0, seg)
        self.bins = np.cumsum(self.theta) / np.sum(self.theta)
        self.widths = np.diff(self.bins)

    def __call__(self, t):
        """Points of the curve at arc-length parameters t in [0, 1] (spherical_curve.py:87-93) -- for plots; the sampler
        kernels never evaluate the curve, they project onto it."""
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        s = self.bins
        i = np.clip(np.digitize(t, s, right=True), 1, len(self.knots) - 1)
        omega, span = self.theta[i], s[i] - s[i - 1]
        wa = np.sin(omega * (s[i] - t) / span) / np.sin(omega)
        wb = np.sin(omega * (t - s[i - 1]) / span) / np.sin(omega)
        return wa[:, None] * self.knots[i - 1] + wb[:, None] * self.knots[i]

    def find_nearest(self, point):
        """The point of the curve closest to `point` (spherical_curve.py:95-102): per segment the clipped great-arc
        projection of distance_slerp, then the first segment of least distance.  Host helper for plots and checks; on
        the sampler's path the projection lives in the kernels (CurvedVonMisesFisher.log_prob evaluates there)."""
        best, out = np.inf, None
        for a, b in zip(self.knots[:-1], self.knots[1:]):
            dist, y = distance_slerp(point, a, b)
            if dist < best:
                best, out = dist, y
        return out


def distance_slerp(x, a, b):
    """(geodesic distance from `x` to the great arc a -> b, closest point of the arc) (spherical_curve.py:10-32)."""
    x, a, b = (np.asarray(v, dtype=np.float64) for v in (x, a, b))
    arc = np.arccos(np.clip(a @ b, -1, 1))
    ax, bx = a @ x, b @ x
    t = np.clip(np.arctan2(bx - ax * np.cos(arc), ax * np.sin(arc)), 0.0, arc)
    y = (np.sin(arc - t) * a + np.sin(t) * b) / (np.sin(arc) + 1e-10)
    return np.arccos(np.clip(x @ y, -1, 1)), y


def brownian_curve(n_points=100, dimension=6, step_size=0.05, seed=1234):
    """Knots of a random walk on the sphere, the construction of spherical_curve.py:105-129
    (same seed -> same knots as the reference: scripts/curve_vMF.py:577-582)."""
    g = np.random.default_rng(seed)
    pts = np.zeros((n_points, dimension))
    z = g.standard_normal(dimension)
    pts[0] = z / (np.linalg.norm(z) + 1e-100)
    for i in range(1, n_points):
        w = pts[i - 1] + g.normal(size=dimension) * step_size
        pts[i] = w / (np.linalg.norm(w) + 1e-100)
    return pts


def constrained_brownian_curve(n_points=100, dimension=6, step_size=0.05, seed=1234):
    """Knots of a smooth walk on the sphere that keeps its heading (spherical_curve.py:132-181; the curve of
    scripts/curve_3d.py): each step turns the unit heading by `step_size` towards a random direction orthogonal to both the
    position and the heading, re-tangentialises it, and moves the point by the angle `step_size` along it.  Same seed ->
    same knots as the reference."""
    g = np.random.default_rng(seed)
    pts = np.zeros((n_points, dimension))
    z = g.standard_normal(dimension)
    pts[0] = z / (np.linalg.norm(z) + 1e-100)

    def tangential_unit(w, p):
        w = w - (w @ p) * p
        return w / np.linalg.norm(w)

    heading = tangential_unit(g.standard_normal(dimension), pts[0])
    for i in range(1, n_points):
        p = pts[i - 1]
        kick = g.standard_normal(dimension)
        kick -= (kick @ p) * p
        kick -= (kick @ heading) * heading
        heading = tangential_unit(heading + step_size * kick / np.linalg.norm(kick), p)
        pts[i] = np.cos(step_size) * p + np.sin(step_size) * heading
    return pts


class CurvedVonMisesFisher(Distribution):
    """log_prob = kappa * x . nearest_point_on_curve(x)  (distributions.py:263-275)."""

    def __init__(self, curve, kappa=100.0):
        if not hasattr(curve, "knots"):
            raise TypeError("curve must expose .knots (SlerpCurve)")
        self.curve = curve
        self.kappa = float(kappa)

    @property
    def d(self):
        return self.curve.knots.shape[-1]

    def _pack(self):
        knots = _as_f64(self.curve.knots)
        return _lib.CURVE_VMF, self.d, knots.shape[0], self.kappa, (None, None, None, knots)

    @counted
    def log_prob(self, x):
        return self._log_prob_device(x)

    @counted
    def gradient(self, x):
        """kappa * the nearest point of the curve (distributions.py:277-278), on the device."""
        return self._gradient_device(x)

"""Sphere helpers around the sampler's path (geosss/sphere.py).

The projections of the sampler's inner loop run inside the HIP kernels; the functions here are what the callers on either
side of it use: the initial-state generator (`sample_sphere`, with its device twin), and the reference's small coordinate /
projection / great-circle helpers, written once for numpy arrays and torch tensors (a tensor stays on its device).
"""
import os

import numpy as np
import torch

from . import _lib


def _device_index(device):
    """None -> $GEOSSS_HIP_DEVICE if set, else torch's current device."""
    if device is None:
        env = os.environ.get("GEOSSS_HIP_DEVICE")
        if env is not None:
            return int(env)
        return int(torch.cuda.current_device()) if torch.cuda.is_available() else 0
    if isinstance(device, torch.device):
        return 0 if device.index is None else int(device.index)
    return int(device)


def current_stream_ptr(dev):
    return int(torch.cuda.current_stream(dev).cuda_stream)


def sample_sphere_device(d, size, seed=0, chain_offset=0, device=None):
    """`size` uniform points on S^d (d+1 ambient components, like geosss.sphere.sample_sphere,
    sphere.py:39-50), generated on the GPU from the counter-based stream.  Returns a
    COMPONENT-major torch tensor [d+1, size] (the layout the sampler keeps its states in)."""
    _lib.require_device()
    dev = _device_index(device)
    out = torch.empty((d + 1, int(size)), dtype=torch.float64, device=f"cuda:{dev}")
    _lib.check(_lib.load().gsss_sample_sphere(int(seed) & (2**64 - 1), int(chain_offset), int(size), d + 1,
                                              out.data_ptr(), dev, current_stream_ptr(dev)))
    return out


def sample_sphere(d=2, size=None, seed=None, device=None, rng="numpy"):
    """Same call signature as geosss.sphere.sample_sphere (sphere.py:39-50); rows are points.

    rng="numpy": the reference's own construction, `radial_projection(default_rng(seed).standard_normal(...))`
    on the host, so that target recipes such as `500 * sample_sphere(2, 10, seed=1234)`
    (scripts/mixture_vMF.py:406-411) or x0 = `sample_sphere(d - 1, seed=1345)` (scripts/curve_vMF.py:577-589)
    build the very arrays the reference builds -- the default, for every size, so that the points of a (d, seed) call never
    depend on how many are asked for beyond the prefix rule of numpy's stream.  rng="philox": the device twin
    (`sample_sphere_device`, the counter-based stream; what 10^6-chain ensembles are initialised with; it needs an explicit
    seed -- there is no fresh entropy on that path -- and honours `device`)."""
    if rng not in ("numpy", "philox"):
        raise ValueError("rng must be 'numpy' or 'philox'")
    n = 1 if size is None else int(size)
    if rng == "philox" and seed is None:
        raise ValueError("rng='philox' is a counter-based stream: pass a seed")
    if rng == "numpy":
        g = np.random.default_rng(seed)
        x = g.standard_normal(d + 1) if size is None else g.standard_normal((n, d + 1))
        norm = np.linalg.norm(x, axis=-1) + 1e-100  # sphere.py:14
        return x / norm if x.ndim == 1 else x / norm[:, None]
    x = sample_sphere_device(d, n, seed=seed, device=device).T.contiguous().cpu().numpy()
    return x[0] if size is None else x


def distance(x, y):
    """Great-circle distance, `geosss.sphere.distance` (sphere.py:64-68): see diagnostics.distance (host or device arrays)."""
    from .diagnostics import distance as _distance
    return _distance(x, y)


# ---- the reference's helper functions (geosss/sphere.py:10-130): numpy in -> numpy out, torch in -> torch out ----------

def _ns(*arrays):
    """torch if any argument is a tensor, numpy otherwise"""
    return torch if any(isinstance(a, torch.Tensor) for a in arrays) else np


def _norm_last(x):
    return torch.linalg.norm(x, dim=-1) if isinstance(x, torch.Tensor) else np.linalg.norm(x, axis=-1)


def radial_projection(x):
    """x / (|x| + 1e-100), row-wise for a batch (sphere.py:10-18; the kernels fuse the same expression)."""
    if not isinstance(x, torch.Tensor):
        x = np.asarray(x, dtype=np.float64)
    r = _norm_last(x) + 1e-100
    return x / r if x.ndim == 1 else x / r[..., None]


def orthogonal_projection(x, y):
    """The part of point(s) `x` orthogonal to the single direction `y` (sphere.py:21-26)."""
    n = radial_projection(y)
    if not isinstance(x, torch.Tensor) and isinstance(n, torch.Tensor):
        x = torch.as_tensor(x, dtype=n.dtype, device=n.device)
    along = x @ n
    return x - (along[..., None] if x.ndim > 1 else along) * n


def spherical_projection(x, v):
    """Point(s) `x` carried onto the great subsphere whose pole is `v` (sphere.py:29-33)."""
    return radial_projection(orthogonal_projection(x, v))


def sample_subsphere(v, seed=None):
    """One uniform point of the great subsphere with pole `v` (sphere.py:53-58): numpy's stream, as the reference."""
    g = np.random.default_rng(seed).standard_normal(len(v))
    if isinstance(v, torch.Tensor):
        g = torch.as_tensor(g, dtype=v.dtype, device=v.device)
    return spherical_projection(g, v)


def sample_marginal(d, size=None, seed=None):
    """One coordinate of a uniform point on S^{d-1}: +-sqrt(Beta(1/2, (d-1)/2)) (sphere.py:93-97)."""
    rng = np.random.default_rng(seed)
    s = rng.beta(0.5, 0.5 * (d - 1), size=size)
    return np.sqrt(s) * rng.choice([-1, 1], size=size)


def cartesian2polar(x):
    """Angle in [0, 2 pi) of the first two coordinates of each row (sphere.py:74-76)."""
    xp = _ns(x)
    return xp.remainder(xp.atan2(x[:, 1], x[:, 0]), 2 * np.pi) if xp is torch else np.mod(np.arctan2(x[:, 1], x[:, 0]), 2 * np.pi)


def polar2cartesian(theta):
    """Rows (cos theta, sin theta) (sphere.py:79-80)."""
    xp = _ns(theta)
    if xp is torch:
        return torch.stack([torch.cos(theta), torch.sin(theta)], dim=-1)
    theta = np.asarray(theta, dtype=np.float64)
    return np.stack([np.cos(theta), np.sin(theta)], axis=-1)


def spherical2cartesian(phi, theta):
    """Azimuth phi, polar angle theta -> rows on S^2 (sphere.py:83-86)."""
    xp = _ns(phi, theta)
    if xp is torch:
        phi, theta = torch.as_tensor(phi), torch.as_tensor(theta)
        st = torch.sin(theta)
        return torch.stack([torch.cos(phi) * st, torch.sin(phi) * st, torch.cos(theta) + 0 * phi], dim=-1)
    phi, theta = np.asarray(phi, dtype=np.float64), np.asarray(theta, dtype=np.float64)
    st = np.sin(theta)
    return np.stack([np.cos(phi) * st, np.sin(phi) * st, np.cos(theta) + 0 * phi], axis=-1)


def cartesian2spherical(x):
    """Rows on S^2 -> (azimuth in [0, 2 pi), polar angle in [0, pi)) (sphere.py:89-90)."""
    if isinstance(x, torch.Tensor):
        return cartesian2polar(x), torch.remainder(torch.acos(x[:, 2]), np.pi)
    return cartesian2polar(x), np.mod(np.arccos(x[:, 2]), np.pi)


def wrap(x, u, v):
    """sin(|x|) u + cos(|x|) v: the wrapping map of Mardia & Jupp (sphere.py:100-103)."""
    t = _norm_last(x)
    xp = _ns(t)
    return xp.sin(t) * u + xp.cos(t) * v


def slerp(u, v):
    """phi -> the point at angle phi from `u` on the great arc towards `v` (sphere.py:106-112)."""
    xp = _ns(u, v)
    omega = (xp.acos if xp is torch else np.arccos)(u @ v)

    def interpolation(phi):
        if xp is torch:
            phi = torch.as_tensor(phi, dtype=omega.dtype, device=omega.device)
        return (xp.sin(omega - phi) * u + xp.sin(phi) * v) / xp.sin(omega)

    return interpolation


def givens(u, v, x):
    """theta -> `x` rotated by theta in the plane of the orthonormal pair (u, v) (sphere.py:115-130)."""
    xp = _ns(u, v, x)
    a, b = u @ x, v @ x
    in_plane, turned = a * u + b * v, a * v - b * u

    def rotate(theta):
        if xp is torch:
            theta = torch.as_tensor(theta, dtype=in_plane.dtype, device=in_plane.device)
        c, s = xp.cos(theta), xp.sin(theta)
        return x + (c - 1.0) * in_plane + s * turned

    return rotate

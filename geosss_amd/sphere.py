"""Sphere helpers that sit on the sampler's path (geosss/sphere.py:10-50).

The projections themselves run inside the HIP kernels; what is exposed here is the
initial-state generator, the device twin of `sphere.sample_sphere`.
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


def sample_sphere(d=2, size=None, seed=None, device=None, rng="auto"):
    """Same call signature as geosss.sphere.sample_sphere (sphere.py:39-50); rows are points.

    rng="numpy": the reference's own construction, `radial_projection(default_rng(seed).standard_normal(...))`
    on the host, so that target recipes such as `500 * sample_sphere(2, 10, seed=1234)`
    (scripts/mixture_vMF.py:406-411) or x0 = `sample_sphere(d - 1, seed=1345)` (scripts/curve_vMF.py:577-589)
    build the very arrays the reference builds.  rng="philox": the device twin (`sample_sphere_device`,
    the counter-based stream; what 10^6-chain ensembles are initialised with).  rng="auto" (default) is
    "numpy" up to 10^5 points and "philox" beyond."""
    if rng not in ("auto", "numpy", "philox"):
        raise ValueError("rng must be 'auto', 'numpy' or 'philox'")
    n = 1 if size is None else int(size)
    if rng == "auto":
        rng = "numpy" if n <= 100_000 else "philox"
    if rng == "numpy":
        g = np.random.default_rng(seed)
        x = g.standard_normal(d + 1) if size is None else g.standard_normal((n, d + 1))
        norm = np.linalg.norm(x, axis=-1) + 1e-100  # sphere.py:14
        return x / norm if x.ndim == 1 else x / norm[:, None]
    x = sample_sphere_device(d, n, seed=0 if seed is None else seed, device=device).T.contiguous().cpu().numpy()
    return x[0] if size is None else x


def distance(x, y):
    """Great-circle distance, `geosss.sphere.distance` (sphere.py:64-68): see diagnostics.distance (host or device arrays)."""
    from .diagnostics import distance as _distance
    return _distance(x, y)

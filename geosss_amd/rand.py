"""Direct (non-MCMC) generators the reference's experiments draw their ground truth from (geosss/rand.py): von
Mises-Fisher by Wood's rejection scheme and Bingham by the angular-central-Gaussian envelope of Kent, Ganeiber and
Mardia.  Batch generators on the device (torch: they are one-shot set-up work of an experiment, not the sampler's hot
path); unlike the reference, which reads numpy's global stream, they take a `seed`.

    sample_vMF(pdf, size=1, seed=None)                 geosss/rand.py:31-66
    sample_bingham(A, n_samples, n_iter=1000, ...)     geosss/rand.py:145-193 (sample_bingham_2d / _3d: the same target
                                                       through the general scheme)
"""
import numpy as np
import torch

from .distributions import Bingham, VonMisesFisher
from .sphere import _device_index


def _generator(seed, dev):
    g = torch.Generator(device=dev)
    if seed is None:
        g.seed()
    else:
        g.manual_seed(int(seed) & (2**63 - 1))
    return g


def _dev(device):
    if not torch.cuda.is_available():
        raise RuntimeError("geosss_amd.rand generates on the GPU: no device is visible")
    return f"cuda:{_device_index(device)}"


def sample_vMF(pdf, size=1, seed=None, device=None, as_tensor=False):
    """`size` draws from VonMisesFisher(mu) (geosss/rand.py:31-66, Wood 1994): the component w along the mean direction
    by rejection from a transformed Beta(p/2, p/2) envelope (p = d - 1), a uniform direction on the orthogonal S^{d-2},
    and a Householder reflection that carries the north pole to mu / |mu| (the reference rotates with an SVD; any
    orthogonal map that sends the pole to the mode gives the same law).  Returns (size, d); size = 1 -> (d,)."""
    if not isinstance(pdf, VonMisesFisher):
        raise AssertionError("sample_vMF expects a VonMisesFisher")
    dev = _dev(device)
    g = _generator(seed, dev)
    mu = torch.as_tensor(np.asarray(pdf.mu, dtype=np.float64), device=dev)
    d = mu.numel()
    n = int(size)
    kappa = float(torch.linalg.norm(mu))
    z = torch.randn((n, d), dtype=torch.float64, device=dev, generator=g)
    if kappa < 1e-12:                                   # the uniform distribution (rand.py:44-45)
        x = z / torch.linalg.norm(z, dim=1, keepdim=True)
    else:
        p = d - 1
        b0 = (-2.0 * kappa + (4.0 * kappa * kappa + p * p) ** 0.5) / p
        x0 = (1.0 - b0) / (1.0 + b0)
        c = kappa * x0 + p * np.log(1.0 - x0 * x0)
        w = torch.empty(n, dtype=torch.float64, device=dev)
        todo = torch.arange(n, device=dev)
        while todo.numel() > 0:                         # whole batches of proposals until every draw is accepted
            m = todo.numel()
            # Beta(p/2, p/2) through two gammas of the seeded generator (torch's Beta sampler takes no generator)
            ga = torch._standard_gamma(torch.full((m,), 0.5 * p, dtype=torch.float64, device=dev), generator=g)
            gb = torch._standard_gamma(torch.full((m,), 0.5 * p, dtype=torch.float64, device=dev), generator=g)
            zb = ga / (ga + gb)
            u = torch.rand(m, dtype=torch.float64, device=dev, generator=g)
            cand = (1.0 - (1.0 + b0) * zb) / (1.0 - (1.0 - b0) * zb)
            ok = kappa * cand + p * torch.log(1.0 - x0 * cand) - c >= torch.log(u)
            w[todo[ok]] = cand[ok]
            todo = todo[~ok]
        v = z[:, :p] / torch.linalg.norm(z[:, :p], dim=1, keepdim=True)      # uniform on S^{d-2}
        y = torch.cat([torch.sqrt(torch.clamp(1.0 - w * w, min=0.0))[:, None] * v, w[:, None]], dim=1)
        # Householder reflection e_d -> mu / kappa
        e = torch.zeros(d, dtype=torch.float64, device=dev)
        e[-1] = 1.0
        h = e - mu / kappa
        hh = float(h @ h)
        x = y if hh < 1e-30 else y - (2.0 / hh) * (y @ h)[:, None] * h[None, :]
    if not as_tensor:
        x = x.cpu().numpy()
    return x[0] if n == 1 else x


def _bfind(v):
    """The b of the envelope: the root of 1 - sum_i 1 / (b + 2 v_i) in [1, d] (geosss/rand.py:132-142), by bisection."""
    v = np.asarray(v, dtype=np.float64)
    d = len(v)
    if np.allclose(v, 0.0):
        return float(d)
    f = lambda b: 1.0 - np.sum(1.0 / (b + 2.0 * v))
    lo, hi = 1.0, float(d)
    if f(lo) >= 0.0:
        return lo
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid) < 0.0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def sample_bingham(A, n_samples, n_iter=1000, return_efficiency=False, seed=None, device=None, as_tensor=False):
    """`n_samples` draws from the Bingham density exp(x^T A x) (the convention of distributions.Bingham.log_prob) by
    rejection from the angular central Gaussian envelope of Kent, Ganeiber and Mardia (geosss/rand.py:145-193).  `A`: the
    matrix, its eigenvalues as a vector, or a Bingham object."""
    if isinstance(A, Bingham):
        A = A.A
    A = np.asarray(A, dtype=np.float64)
    dev = _dev(device)
    g = _generator(seed, dev)
    if A.ndim == 1:
        v, U = -A.copy(), None
    else:
        v, U = np.linalg.eigh(-A)
    v = v - np.min(v)
    d = len(v)
    b = _bfind(v)
    log_m = -(d - b) / 2.0 + (d / 2.0) * np.log(d / b)
    vt = torch.as_tensor(v, device=dev)
    scale = torch.rsqrt(1.0 + 2.0 * vt / b)
    need = int(n_samples)
    kept, eff = [], []
    for _ in range(int(n_iter)):
        m = max(need, 1)
        x = torch.randn((m, d), dtype=torch.float64, device=dev, generator=g) * scale      # ACG(Omega = I + 2 diag(v) / b)
        x = x / torch.linalg.norm(x, dim=1, keepdim=True)
        u = (x * x) @ vt
        log_prob = -u + (d / 2.0) * torch.log1p(2.0 * u / b) - log_m                    # Bingham / (envelope M ACG)
        ok = torch.log(torch.rand(m, dtype=torch.float64, device=dev, generator=g)) < log_prob
        kept.append(x[ok])
        eff.append(float(ok.double().mean()))
        need -= int(ok.sum())
        if need <= 0:
            break
    out = torch.cat(kept, 0)[: int(n_samples)]
    if U is not None:
        out = out @ torch.as_tensor(U, device=dev).T
    if not as_tensor:
        out = out.cpu().numpy()
    return (out, float(np.mean(eff))) if return_efficiency else out


def sample_bingham_2d(pdf, n_samples=1, **kw):
    """geosss/rand.py:69-90 (a von Mises draw there); here the general envelope scheme on the same target."""
    if not (isinstance(pdf, Bingham) and pdf.d == 2):
        raise ValueError("expected 2D Bingham distribution")
    return sample_bingham(pdf, n_samples, **kw)


def sample_bingham_3d(pdf, n_samples=1, **kw):
    """geosss/rand.py:93-129 (a Gibbs sampler there: correlated draws); here independent draws by the general scheme."""
    if not (isinstance(pdf, Bingham) and pdf.d == 3):
        raise ValueError("3D Bingham expected")
    return sample_bingham(pdf, n_samples, **kw)

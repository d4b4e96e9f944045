"""Chain diagnostics with the reference's definitions, batched over chains and device-resident.

    acf, acf_fft, IAT, n_eff      geosss/utils.py:96-134
    ess_bulk                      the arviz estimator the paper's relative ESS comes from (scripts/bingham.py:43-57)
    distance                      geosss/sphere.py:64-68
    hopping_frequency             scripts/bingham.py:23-25
    mode_occupancy, mode_kl       scripts/vMF_diagnostics.py:335-342

Inputs may be numpy arrays or torch tensors (CPU or GPU); the time axis is the last-but-one for
sample arrays (..., n_draws, d) and the last one for scalar series (..., n_draws), so a whole
(chains, draws, dims) ensemble from `sampler.sample(..., as_tensor=True)` is analysed without leaving
the GPU.  One scalar series gives the reference's scalar.
"""
import numpy as np
import torch

__all__ = ["acf", "acf_fft", "IAT", "n_eff", "distance", "hopping_frequency", "mode_occupancy", "mode_kl", "from_running",
           "iat_from_acf", "ess_bulk", "ess_between_chains"]


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float64))


def _back(y, like):
    if isinstance(like, torch.Tensor):
        return y
    y = y.cpu().numpy()
    return y.item() if y.ndim == 0 else y


def acf_fft(x):
    """Autocorrelation by the convolution theorem, lags 0 .. n//2-1 (utils.py:113-116), along the last axis."""
    t = _t(x).to(torch.float64)
    n = t.shape[-1]
    z = (t - t.mean(-1, keepdim=True)) / t.std(-1, unbiased=False, keepdim=True)
    f = torch.fft.rfft(z, dim=-1)
    # like np.fft.irfft without `n`: the inverse has 2*(len(f)-1) points (n-1 for odd n), as in the reference
    ac = torch.fft.irfft(f.conj() * f, n=2 * (f.shape[-1] - 1), dim=-1)[..., : n // 2] / n
    return _back(ac, x)


def acf(x, n_max=None):
    """Direct autocorrelation estimate for lags < n_max (utils.py:96-110), along the last axis."""
    t = _t(x).to(torch.float64)
    n = t.shape[-1]
    n_max = n_max or n // 2
    z = t - t.mean(-1, keepdim=True)
    ac = torch.stack([(z[..., i:] * z[..., : n - i]).mean(-1) for i in range(n_max)], dim=-1)
    return _back(ac / ac[..., :1], x)


def IAT(x, n=None):
    """Integrated autocorrelation time by the heuristic of utils.py:119-131: adjacent-pair sums of the
    autocorrelation from lag 2, truncated at the first negative pair."""
    ac = _t(acf_fft(_t(x)))
    if n:
        ac = ac[..., :n]
    m = ac.shape[-1]
    tail = ac[..., 2:-1] if m % 2 != 0 else ac[..., 2:]
    sums = tail.reshape(*tail.shape[:-1], -1, 2).sum(-1)
    neg = sums < 0
    has = neg.any(-1)
    first = torch.argmax(neg.to(torch.int64), dim=-1)                 # index of the first negative pair
    L = torch.where(has, 1 + 2 * first, torch.full_like(first, m - 1))
    lag = torch.arange(m, device=ac.device)
    keep = (lag >= 1) & (lag <= L.unsqueeze(-1))
    s = (ac * keep).sum(-1)
    return _back(1.0 + torch.clamp(2.0 * s, min=0.0), x)


def n_eff(x, n=None):
    """Effective sample size len(x) / IAT(x) (utils.py:132-134)."""
    t = _t(x)
    return _back(t.shape[-1] / _t(IAT(t, n)), x)


def distance(x, y):
    """Great-circle distance arccos(clip(x.y, -1, 1)) (sphere.py:64-68)."""
    a, b = _t(x).to(torch.float64), _t(y).to(torch.float64)
    return _back(torch.arccos(torch.clamp((a * b).sum(-1), -1.0, 1.0)), x)


def hopping_frequency(samples, mode):
    """Fraction of consecutive draws on opposite sides of the mode's equator (scripts/bingham.py:23-25);
    samples (..., n_draws, d)."""
    s = _t(samples).to(torch.float64)
    side = torch.sign(s @ _t(mode).to(s))
    return _back((side[..., 1:] != side[..., :-1]).to(torch.float64).mean(-1), samples)


def mode_occupancy(samples, modes):
    """Share of draws whose nearest mode (largest x.mu_k) is k; samples (..., d) flattened."""
    s = _t(samples).to(torch.float64).reshape(-1, _t(samples).shape[-1])
    k = torch.argmax(s @ _t(modes).to(s).T, dim=1)
    occ = torch.bincount(k, minlength=len(modes)).to(torch.float64) / len(k)
    return _back(occ, samples)


def mode_kl(occupancy, weights):
    """KL(occupancy || weights) with the reference's 1e-100 floor for empty modes."""
    p = _t(occupancy).to(torch.float64)
    p = torch.where(p > 0, p, torch.full_like(p, 1e-100))
    w = _t(weights).to(p)
    return _back((p * torch.log(p / w)).sum(), occupancy)


def saff_sphere(n=1000):
    """n points spread over S^2 along the Saff-Kuijlaars spiral, as the reference's evaluation notebooks lay out their
    histogram cells (scripts/visualize_curve_vMF.ipynb `saff_sphere`): heights h_k equally spaced in [-1, 1], azimuth
    advancing by 3.6 / sqrt(n (1 - h_k^2)); the two poles at azimuth 0."""
    h = np.linspace(-1.0, 1.0, n)
    phi = np.zeros(n)
    phi[1:-1] = np.cumsum(3.6 / np.sqrt(n * (1.0 - h[1:-1] ** 2)))   # azimuth of point k: the increments of points 1 .. k
    theta = np.arccos(h)
    return np.stack([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)], axis=1)


def grid_kl(pdf, samples, n_saff=1500, eps=1e-12):
    """KL divergence between a target on S^2 and the draws, the estimator of the reference's curve evaluation
    (scripts/visualize_curve_vMF.ipynb `calc_kld`): p = the target's probabilities on the `n_saff` spiral points
    (normalised over them), q = the share of draws whose nearest spiral point is each cell (+ eps), summed over the
    cells with p > eps.  samples: (draws, 3) or (chains, draws, 3) -> one value per chain.  Runs where the draws live."""
    s = _t(samples).to(torch.float64)
    single = s.dim() == 2
    if single:
        s = s[None]
    grid_np = saff_sphere(n_saff)
    logp = torch.as_tensor(np.asarray(pdf.log_prob(grid_np), dtype=np.float64), device=s.device)
    p = torch.exp(logp - torch.logsumexp(logp, 0))
    grid = torch.as_tensor(grid_np, device=s.device)
    out = []
    for chain in s:                                    # nearest spiral point = largest dot product on the unit sphere
        cell = torch.cat([torch.argmax(c @ grid.T, dim=1) for c in chain.split(1 << 18)])
        q = torch.bincount(cell, minlength=n_saff).to(torch.float64) / len(cell) + eps
        m = p > eps
        out.append((p[m] * (torch.log(p[m]) - torch.log(q[m]))).sum())
    kl = torch.stack(out)
    return _back(kl[0] if single else kl, samples)


def pair_sum_went_negative(ac):
    """Whether the pair-sum rule of utils.py:119-131 found its stopping point inside the given lags (per series)."""
    ac = _t(ac)
    m = ac.shape[-1]
    tail = ac[..., 2:-1] if m % 2 != 0 else ac[..., 2:]
    return (tail.reshape(*tail.shape[:-1], -1, 2).sum(-1) < 0).any(-1)


def iat_from_acf(ac):
    """The IAT heuristic of utils.py:119-131 applied to a given autocorrelation (lags 0 .. m-1 along the last
    axis): adjacent-pair sums from lag 2, truncated at the first negative pair."""
    ac = _t(ac)
    m = ac.shape[-1]
    tail = ac[..., 2:-1] if m % 2 != 0 else ac[..., 2:]
    sums = tail.reshape(*tail.shape[:-1], -1, 2).sum(-1)
    neg = sums < 0
    has = neg.any(-1)
    first = torch.argmax(neg.to(torch.int64), dim=-1)
    L = torch.where(has, 1 + 2 * first, torch.full_like(first, m - 1))
    lag = torch.arange(m, device=ac.device)
    keep = (lag >= 1) & (lag <= L.unsqueeze(-1))
    return 1.0 + torch.clamp(2.0 * (ac * keep).sum(-1), min=0.0)


def from_running(acc, d, n_modes, n_lags, second_moment=True):
    """Diagnostics from the running statistics the sampler kernels accumulate (include/gsss.h: gsss_run_args.stats_dev;
    rows x chains).  Per chain, with the reference's definitions on the retained series:
        n, mean (d), second_moment (d, d; absent when the rows were left out: GSSS_STATS_NO_SECOND_MOMENT),
        geodesic_step (sphere.distance of consecutive draws, mean),
        hopping_frequency (scripts/bingham.py:23-25), mode_occupancy (K; scripts/vMF_diagnostics.py:335-342),
        acf (lags 0 .. L of the projection, the direct estimator utils.acf, utils.py:96-110 -- identical to
        diagnostics.acf(series, n_max=L+1)), iat / n_eff (pair-sum heuristic of utils.py:119-134 on that acf),
        iat_truncated (True where no adjacent pair of the L lags went negative: the sum stopped at the window's end and
        the IAT is a LOWER bound -- raise `lags` or thin more)."""
    acc = _t(acc).to(torch.float64)
    T = d * (d + 1) // 2 if second_moment else 0
    r_sum, r_xx = 1 + d, 1 + 2 * d
    r_dist = r_xx + T
    r_hop, r_mode = r_dist + 1, r_dist + 2
    r_p = r_mode + n_modes
    r_lag, r_ring, r_head = r_p + 2, r_p + 2 + n_lags, r_p + 2 + 2 * n_lags
    n = acc[0]
    out = {"n": n, "mean": (acc[r_sum:r_sum + d] / n).T}
    if second_moment:
        sm = torch.zeros((acc.shape[1], d, d), dtype=torch.float64, device=acc.device)
        iu = torch.triu_indices(d, d)
        sm[:, iu[0], iu[1]] = (acc[r_xx:r_xx + T] / n).T
        sm[:, iu[1], iu[0]] = (acc[r_xx:r_xx + T] / n).T
        out["second_moment"] = sm
    out["geodesic_step"] = acc[r_dist] / (n - 1)
    out["hopping_frequency"] = acc[r_hop] / (n - 1)
    if n_modes:
        out["mode_occupancy"] = (acc[r_mode:r_mode + n_modes] / n).T
    out["proj_mean"] = acc[r_p] / n                                            # mean and (biased) variance of p = x . w per chain
    out["proj_var"] = acc[r_p + 1] / n - out["proj_mean"] ** 2
    if n_lags:
        L = n_lags
        if bool((n <= L).any()):
            raise ValueError("the autocorrelation needs more than `lags` retained draws per chain")
        sp, spp = acc[r_p], acc[r_p + 1]
        mu = sp / n
        lag = torch.arange(1, L + 1, device=acc.device, dtype=torch.float64)[:, None]
        head = torch.cumsum(acc[r_head:r_head + L], dim=0)                    # sum of the first l values
        # the last l values: ring slot (n - j) mod L holds p_{n-j}, j = 1 .. L
        j = torch.arange(1, L + 1, device=acc.device)[:, None]
        slot = torch.remainder(n.to(torch.int64)[None, :] - j, L)
        tail = torch.cumsum(torch.gather(acc[r_ring:r_ring + L], 0, slot), dim=0)
        c = acc[r_lag:r_lag + L]
        cov = (c - mu * ((sp - head) + (sp - tail)) + (n - lag) * mu * mu) / (n - lag)
        var = (spp - n * mu * mu) / n
        ac = torch.cat([torch.ones_like(var)[None], cov / var], dim=0).T     # (chains, L + 1)
        out["acf"] = ac
        out["iat"] = iat_from_acf(ac)
        out["n_eff"] = n / out["iat"]
        out["iat_truncated"] = ~pair_sum_went_negative(ac)
    return out


def ess_between_chains(chain_means, n, chain_vars):
    """Integrated autocorrelation time and effective sample size of a scalar quantity from MANY independent stationary chains,
    with no lag window at all: for a stationary chain Var(mean of n draws) = Var(x) tau_n / n with
    tau_n = 1 + 2 sum_{k<n} (1 - k/n) rho_k -> tau, so over C chains of n draws each

        tau = n Var_c(mean_c) / Var(x),      ESS per chain = n / tau = Var(x) / Var_c(mean_c),

    Var(x) = the pooled variance of all draws = mean_c(within-chain variance) + Var_c(mean_c) (law of total variance).
    Inputs per chain: the mean, the number of draws and the (biased, 1/n) variance of the series -- `proj_mean`, `n`,
    `proj_var` of `from_running`, i.e. the sums the sampler kernels already keep.  What the windowed estimators
    (geosss/utils.py:109-134 IAT / n_eff on a truncated autocorrelation) can only bound from below when the chain mixes
    slower than the window, this measures -- to a relative standard error sqrt(2 / (C - 1)) -- provided the chains ARE
    independent and stationary (start them from draws of the target, or burn in several tau) and n >> tau (the finite-n
    factor (1 - k/n) biases tau low by ~ tau / n otherwise).  Returns a dict of floats:
    tau (in retained draws), ess_per_chain, ess_total, rel_se, n, chains."""
    m, v = _t(chain_means).to(torch.float64).reshape(-1), _t(chain_vars).to(torch.float64).reshape(-1)
    nn = _t(n).to(torch.float64).reshape(-1)
    if m.numel() < 2:
        raise ValueError("the between-chain estimator needs at least two chains")
    if float(nn.max() - nn.min()) != 0.0:
        raise ValueError("the between-chain estimator takes chains of equal length")
    n_draws = float(nn[0])
    between = float(m.var(unbiased=True))
    total = float(v.mean()) + between
    if not between > 0.0 or not total > 0.0:
        return {"tau": float("nan"), "ess_per_chain": float("nan"), "ess_total": float("nan"), "rel_se": float("nan"),
                "n": n_draws, "chains": int(m.numel())}
    tau = n_draws * between / total
    return {"tau": tau, "ess_per_chain": n_draws / tau, "ess_total": m.numel() * n_draws / tau,
            "rel_se": float(np.sqrt(2.0 / (m.numel() - 1))), "n": n_draws, "chains": int(m.numel())}


def _average_ranks(flat):
    """scipy.stats.rankdata(method='average') of a 1-D tensor: 1-based ranks, ties share the mean of their ranks."""
    order = torch.argsort(flat, stable=True)
    sorted_vals = flat[order]
    n = flat.numel()
    pos = torch.arange(1, n + 1, dtype=torch.float64, device=flat.device)
    # runs of equal values: every member gets the mean position of the run
    _, inverse, counts = torch.unique_consecutive(sorted_vals, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).to(torch.float64)
    mean_pos = ends - (counts.to(torch.float64) - 1.0) / 2.0
    ranks = torch.empty(n, dtype=torch.float64, device=flat.device)
    ranks[order] = mean_pos[inverse]
    del pos
    return ranks


def _autocov_fft(z):
    """Biased autocovariance of every row (divided by n), lags 0 .. n-1, by the convolution theorem on a zero-padded transform."""
    n = z.shape[-1]
    m = 1 << int(2 * n - 1).bit_length()
    c = z - z.mean(-1, keepdim=True)
    f = torch.fft.rfft(c, n=m, dim=-1)
    return torch.fft.irfft(f * f.conj(), n=m, dim=-1)[..., :n] / n


def ess_bulk(x, relative=False):
    """Rank-normalised split-chain bulk effective sample size of a scalar quantity over several chains, x (chains, draws) --
    what `arviz.ess(..., method="bulk")` computes and the reference's experiments report as "relative ESS"
    (scripts/bingham.py:43-57: the draws projected on the target's mode, 10 chains, az.ess(relative=True);
    scripts/vMF_diagnostics.py:466-478).  Restated from the published algorithm (Vehtari, Gelman, Simpson, Carpenter, Buerkner
    2021, "Rank-normalization, folding, and localization: an improved R-hat", section 3, as implemented in Stan and ArviZ 0.x;
    arviz itself is not installed in this image):
      1. split every chain into its two halves (2 C chains of N = draws // 2);
      2. replace the pooled values by their average ranks r, z = Phi^-1((r - 3/8) / (S + 1/4)), S = 2 C N;
      3. per-chain biased autocovariances (FFT), W = mean_c acov_c[0] N / (N - 1), var+ = W (N - 1) / N + var_c(chain means);
      4. rho_t = 1 - (W - mean_c acov_c[t]) / var+, Geyer's initial positive sequence over pairs (rho_2k + rho_2k+1 > 0), then the
         initial monotone sequence;
      5. tau = -1 + 2 sum_{t <= max_t} rho_t + rho_{max_t + 1}, floored at 1 / log10(S); ESS = S / tau (relative: 1 / tau).
    The ranking, the transforms and the correlations run where `x` lives (a CUDA tensor stays on the device; 10 chains x 10^5
    draws take milliseconds); the sequential truncation runs on the host over the S / (2 C) lags."""
    t = _t(x).to(torch.float64)
    if t.dim() != 2 or t.shape[1] < 4:
        raise ValueError("ess_bulk takes (chains, draws) with at least 4 draws")
    half = t.shape[1] // 2
    split = torch.cat([t[:, :half], t[:, t.shape[1] - half:]], dim=0)            # (2 C, N)
    n_chain, n_draw = split.shape
    size = split.numel()
    if float(split.max() - split.min()) < 1e-15:
        return float(1.0 if relative else size)
    ranks = _average_ranks(split.reshape(-1))
    z = torch.special.ndtri((ranks - 0.375) / (size + 0.25)).reshape(n_chain, n_draw)
    acov = _autocov_fft(z)                                                      # (2 C, N)
    mean_acov = acov.mean(0)
    mean_var = float(mean_acov[0]) * n_draw / (n_draw - 1.0)
    var_plus = mean_var * (n_draw - 1.0) / n_draw
    if n_chain > 1:
        var_plus += float(z.mean(1).var(unbiased=True))
    rho = (1.0 - (mean_var - mean_acov) / var_plus).cpu().numpy()               # rho_hat for every lag
    rho_t = np.zeros(n_draw)
    rho_even, rho_odd = 1.0, float(rho[1])
    rho_t[0], rho_t[1] = rho_even, rho_odd
    k = 1
    while k < n_draw - 3 and rho_even + rho_odd > 0.0:                          # Geyer's initial positive sequence
        rho_even, rho_odd = float(rho[k + 1]), float(rho[k + 2])
        if rho_even + rho_odd >= 0.0:
            rho_t[k + 1], rho_t[k + 2] = rho_even, rho_odd
        k += 2
    max_t = k - 2
    if rho_even > 0.0:
        rho_t[max_t + 1] = rho_even
    k = 1
    while k <= max_t - 2:                                                       # Geyer's initial monotone sequence
        if rho_t[k + 1] + rho_t[k + 2] > rho_t[k - 1] + rho_t[k]:
            rho_t[k + 1] = (rho_t[k - 1] + rho_t[k]) / 2.0
            rho_t[k + 2] = rho_t[k + 1]
        k += 2
    tau = -1.0 + 2.0 * float(np.sum(rho_t[: max_t + 1])) + float(np.sum(rho_t[max_t + 1: max_t + 2]))
    tau = max(tau, 1.0 / np.log10(size))
    return float((1.0 if relative else size) / tau)

"""Geodesic slice samplers on the sphere for many chains at once, API-compatible with
geosss/mcmc.py for the two slice samplers:

    RejectionSphericalSliceSampler(distribution, initial_state, seed=None)   mcmc.py:335-374
    ShrinkageSphericalSliceSampler(distribution, initial_state, seed=None)   mcmc.py:377-401
        next(sampler)                                  one transition (of every chain)
        sampler.sample(n_samples, burnin=0, return_all_samples=False)        mcmc.py:55-77
        sampler.target / .state / .rng / .n_reject                           mcmc.py:43-45, 355

`initial_state` may be one point (d,) -- then everything has the reference's shapes -- or
(n_chains, d): then `sample` returns (n_chains, n_samples, d), the (chains, draws, dims)
convention the reference's harness feeds to its ESS estimator.

All transitions run in the HIP kernels behind `gsss_run` (include/gsss.h); chain states stay
resident in HBM between calls.  Two random streams:

  rng="philox" (default)  the library's counter-based stream (DESIGN.md §3 "Random streams"), keyed by
      `seed`, chain id and step id: results do not depend on how chains are split over devices or
      steps over calls; this is the throughput path.
  rng="numpy"             numpy's own PCG64 + ziggurat stream restated on the device: a chain seeded
      like the reference (`np.random.default_rng(seed)`, mcmc.py:45) consumes exactly the numbers
      the reference consumes, so `Sampler(pdf, x0, seed).sample(n, burnin)` reproduces geosss's
      output from the seed alone (to rounding, ~1e-14).  For one chain `sampler.rng` is kept in
      step with the device stream, as the reference's attribute would be.  Many chains: one generator per
      chain (a list of seeds, or SeedSequence(seed).spawn(n_chains)); fast mode serves the stream for the
      lane-per-chain shapes (one lane per chain in large ensembles), the exact kernels for every shape.
"""
import copy
import ctypes as C
import warnings

import numpy as np
import torch

import contextlib
import os

from . import _lib, _pinned
from .sphere import _device_index, current_stream_ptr

__all__ = ["determine_burnin", "RejectionSphericalSliceSampler", "ShrinkageSphericalSliceSampler", "MetropolisHastings",
           "SphericalHMC"]

_MODES = {"exact": _lib.MODE_EXACT, "fast": _lib.MODE_FAST, "auto": None}
_MAX_STEPS_PER_LAUNCH = 4096
# sample() -> ndarray: below this many bytes the array is copied in one piece; above, blocks of chains are sampled and copied in a
# pipeline (the copy of block k under the kernel of block k + 1)
_PIPELINE_MIN_BYTES = 64 << 20
_PINNED_MIN_BYTES = 1 << 20   # below: an ordinary (pageable) array
_COPY_GBS = 45.0          # what a device-to-host copy into page-locked memory sustains (planning figure for the number of blocks)
_ROUND_STEP_S = 16e-6     # one transition of one resident round of workgroups (README mixture: 15 us; planning figure)


def determine_burnin(n_samples, burnin):
    """float = fraction of n_samples, int = count (mcmc.py:13-19)."""
    if isinstance(burnin, float):
        if not 0 <= burnin <= 1.0:
            raise AssertionError("fractional burnin must be in [0, 1]")
        return int(burnin * n_samples)
    if burnin < 0:
        raise AssertionError("burnin must be >= 0")
    return int(burnin)


_children_built = {}  # id(bit generator) -> [the bit generator, samplers keyed from it so far]


def _forget_dead_generators():
    """Drop the bit generators nobody but this table holds any more (numpy's bit generators take no weak references, so the
    table holds strong ones and looks at the reference count: the entry's list, the loop variable and getrefcount's argument).
    A generator that is still alive keeps its entry -- evicting it would restart its count and hand the next sampler built from
    it the key of the first."""
    import sys
    for key in list(_children_built):
        bg = _children_built[key][0]
        if sys.getrefcount(bg) <= 3:
            del _children_built[key]
        del bg


def seed_to_key(seed):
    """64-bit Philox key from what np.random.default_rng accepts as `seed` (mcmc.py:45).  A Generator / BitGenerator is
    neither advanced nor otherwise changed: successive samplers built from one generator get successive children of its
    seed sequence (the reference's samplers would share the generator and so draw different numbers too)."""
    if seed is None:
        w = np.random.SeedSequence().generate_state(2, np.uint32)
    elif isinstance(seed, (int, np.integer)):
        if seed < 0:
            raise ValueError("seed must be non-negative")
        if int(seed) < 2**64:
            return int(seed)
        w = np.random.SeedSequence(int(seed)).generate_state(2, np.uint32)
    elif isinstance(seed, np.random.SeedSequence):
        w = seed.generate_state(2, np.uint32)
    elif isinstance(seed, (np.random.Generator, np.random.BitGenerator)):
        # The reference shares the caller's generator (default_rng(gen) returns it, mcmc.py:45), so two samplers built from one
        # generator draw different numbers.  Here every construction keys the Philox stream from a further child of the
        # generator's seed sequence: successive samplers get different keys, the caller's stream position and seed sequence
        # are not touched.  A generator without a seed sequence (restored from a state) has nothing to derive from: the key
        # is drawn from the generator itself, which advances it -- as round 1 did.
        bg = seed.bit_generator if isinstance(seed, np.random.Generator) else seed
        ss = getattr(bg, "seed_seq", None)
        if isinstance(ss, np.random.SeedSequence):
            # a child of the generator's seed sequence WITHOUT touching the caller's object (ss.spawn would advance its
            # n_children_spawned and so change what the caller's own later spawn() calls return): the library counts the
            # samplers built from this bit generator itself and derives child number 2^31 + count (a fresh generator of
            # the same seed starts at 0 again: same program, same keys)
            entry = _children_built.get(id(bg))
            if entry is None or entry[0] is not bg:
                if len(_children_built) >= 1024:
                    _forget_dead_generators()                  # (bounded by the LIVE generators: a live one's count never restarts)
                entry = _children_built[id(bg)] = [bg, 0]      # (the reference keeps the id from being reused)
            count = entry[1]
            entry[1] = count + 1
            child = np.random.SeedSequence(entropy=ss.entropy, spawn_key=tuple(ss.spawn_key) + (2**31 + count,),
                                           pool_size=ss.pool_size)
            w = child.generate_state(2, np.uint32)
        else:
            g = seed if isinstance(seed, np.random.Generator) else np.random.Generator(seed)
            w = g.integers(0, 2**32, size=2, dtype=np.uint64)
    else:
        w = np.random.SeedSequence(seed).generate_state(2, np.uint32)
    return int(w[0]) | (int(w[1]) << 32)


_warned_shapes = set()


def _warn_exact_fallback(distribution):
    """mode='auto' landed on the generic (exact) kernels: say so once per target shape."""
    key = (type(distribution).__name__, distribution.d)
    if key in _warned_shapes or hasattr(distribution, "source"):  # (registration targets have no restricted form)
        return
    _warned_shapes.add(key)
    warnings.warn(f"geosss_amd: no fast-mode kernel covers this {key[0]} target (d={key[1]}); mode='auto' uses the "
                  "exact kernels (2.5-10x lower throughput; see gsss_mode_supported)", RuntimeWarning, stacklevel=3)


class RejectionSphericalSliceSampler:
    """Geodesic rejection slice sampler (mcmc.py:335-374), many chains."""

    _sampler = _lib.REJECT

    def __init__(self, distribution, initial_state, seed=None, *, device=None, mode="auto", max_tries=None,
                 chain_offset=0, step_offset=0, variant=0, rng="philox", placement="auto", screen=True):
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' or 'numpy'")
        if placement not in ("auto", "packed", "spread"):
            raise ValueError("placement must be 'auto', 'packed' or 'spread'")
        many_seeds = isinstance(seed, (list, tuple))
        if many_seeds and rng != "numpy":
            raise ValueError("a list of seeds (one numpy generator per chain) needs rng='numpy'; the Philox stream "
                             "takes one seed and keys every chain by its global chain id")
        _lib.require_device()
        self._lib = _lib.load()
        self.target = distribution
        self.rng_kind = rng
        self._placement = {"auto": 0, "packed": 1, "spread": 2}[placement]
        self._seed_arg = seed
        self.rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(
            seed[0] if many_seeds else seed)
        self.seed = 0 if rng == "numpy" else seed_to_key(seed[0] if many_seeds else seed)
        self.device = _device_index(device)
        self._tdev = f"cuda:{self.device}"
        self.mode = mode
        if mode not in _MODES:
            raise ValueError(f"mode must be one of {sorted(_MODES)}")
        self.max_tries = int(max_tries) if max_tries is not None else (1 << 20)
        self.chain_offset = int(chain_offset)
        self.variant = int(variant)
        # fast mode: single-precision screening of the tries (same chains); False = the all-double kernels; "verify" = the
        # default kernel with its screen's verdicts ignored where it can (GSSS_VARIANT_FAST_VERIFY: must give the same bits)
        self.screen = screen if screen == "verify" else bool(screen)
        self._step = int(step_offset)
        self._target_dev = distribution._device_target(self.device)
        self._set_state(initial_state)
        if rng == "numpy":
            # a generator per chain, sequential: fast mode serves it where a chain is one lane's (or, small ensembles, one
            # wavefront's) -- the lane-per-chain shapes; the cooperative shapes run the exact kernels
            lane_ok = not variant and self._lib.gsss_variant_name(self._target_dev.handle, _lib.MODE_FAST, 0) == b"fast-lane"
            if mode == "fast" and not lane_ok:
                raise ValueError("rng='numpy' in fast mode needs a shape the lane-per-chain fast kernels are built for "
                                 "(vMF mixtures and Bingham targets up to d = 10, the listed curve dimensions); "
                                 "use mode='exact' or 'auto'")
            if mode == "auto":
                mode = self.mode = "fast" if lane_ok else "exact"
        if mode == "auto":  # the throughput kernels where they are built for this shape, else the generic ones
            fast_ok = self._lib.gsss_mode_supported(self._target_dev.handle, _lib.MODE_FAST) and not variant
            self.mode = "fast" if fast_ok else "exact"
            if not fast_ok and not variant:
                _warn_exact_fallback(distribution)
        n = self.n_chains
        self._n_reject = torch.zeros(n, dtype=torch.int64, device=self._tdev)
        self._n_tries = torch.zeros(n, dtype=torch.int64, device=self._tdev)
        self._err = torch.zeros(n, dtype=torch.int32, device=self._tdev)
        self._tries_reported = 0
        self._rng_state = self._numpy_states(seed) if rng == "numpy" else None
        self._stats = None

    # ------------------------------------------------------------------ running statistics
    def enable_stats(self, lags=32, projection=None, hop=None, modes=None, second_moment=None):
        """Accumulate running statistics of the retained series inside the sampler kernels (gsss_run_args.stats_dev):
        moments, geodesic step between consecutive draws, hopping frequency across the equator of `hop`, occupancy
        of `modes`, and the lag sums of the projection x . `projection` from which the reference's autocorrelation
        (utils.py:96-110) and its IAT / n_eff heuristic (:119-134) follow -- without storing a single draw.
        Statistics are taken at the cadence of `advance(..., thin=t)` (every t-th state), like diagnostics computed
        from a thinned stored chain.  Defaults: projection = first coordinate; hop = the target's `.mode` if it has
        one; modes = the component means of a MixtureModel.  second_moment: keep the d (d + 1) / 2 sums of x_i x_j too
        (default: for d <= 16; they grow as d^2 -- 20 100 rows per chain at d = 200).  Every slice-sampler kernel family
        accumulates them (lane, lane-group and cooperative layouts)."""
        d = self.d
        w = np.zeros(d) if projection is None else np.asarray(projection, dtype=np.float64)
        if projection is None:
            w[0] = 1.0
        if hop is None:
            hop = getattr(self.target, "mode", None)
            hop = np.zeros(d) if hop is None or np.ndim(hop) != 1 else hop
        if modes is None:
            pdfs = getattr(self.target, "pdfs", None)
            modes = np.array([p.mu for p in pdfs]) if pdfs else np.zeros((0, d))
        modes = np.asarray(modes, dtype=np.float64).reshape(-1, d)
        dirs = np.concatenate([w[None], np.asarray(hop, dtype=np.float64)[None], modes], axis=0)
        if dirs.shape[1] != d:
            raise ValueError("projection / hop / modes must have d components")
        second_moment = d <= 16 if second_moment is None else bool(second_moment)
        flags = 0 if second_moment else _lib.STATS_NO_SECOND_MOMENT
        rows = int(self._lib.gsss_stats_rows(d, len(modes), int(lags), flags))
        if rows < 0:
            raise ValueError("bad statistics shape")
        self._stats = {"lags": int(lags), "modes": len(modes), "flags": flags,
                       "dirs": torch.from_numpy(np.ascontiguousarray(dirs)).to(self._tdev),
                       "acc": torch.zeros((rows, self.n_chains), dtype=torch.float64, device=self._tdev)}
        return self

    def stats(self):
        """The running statistics as a dict of CUDA tensors, one entry per chain (see diagnostics.from_running)."""
        if self._stats is None:
            raise ValueError("call enable_stats() first")
        from . import diagnostics
        return diagnostics.from_running(self._stats["acc"], self.d, self._stats["modes"], self._stats["lags"],
                                        second_moment=not (self._stats["flags"] & _lib.STATS_NO_SECOND_MOMENT))

    def _numpy_states(self, seed):
        """[n_chains, 4] PCG64 words (state_hi, state_lo, inc_hi, inc_lo), one default_rng per chain:
        the sampler's own generator for a single chain; for many chains the given list of seeds, or
        SeedSequence(seed).spawn(n_chains) (the pattern of scripts/bingham.py:87-88)."""
        n = self.n_chains
        if n == 1:
            gens = [self.rng]
        elif isinstance(seed, (list, tuple)):
            if len(seed) != n:
                raise ValueError("one seed per chain")
            gens = [s if isinstance(s, np.random.Generator) else np.random.default_rng(s) for s in seed]
        else:
            root = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(seed)
            gens = [np.random.default_rng(s) for s in root.spawn(n)]
        words = np.empty((n, 4), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i, g in enumerate(gens):
            bg = g.bit_generator
            if not isinstance(bg, np.random.PCG64):
                raise TypeError("rng='numpy' restates numpy's default PCG64 bit generator")
            st = bg.state["state"]
            words[i] = [st["state"] >> 64, st["state"] & mask, st["inc"] >> 64, st["inc"] & mask]
        return torch.from_numpy(words.view(np.int64)).to(self._tdev)

    def _sync_rng(self):
        """One chain: put the device stream's position back into `self.rng` (what the reference's
        sampler.rng would hold after the same calls)."""
        if self._rng_state is None or self.n_chains != 1:
            return
        w = self._rng_state.cpu().numpy().view(np.uint64)[0]
        st = self.rng.bit_generator.state
        st["state"] = {"state": (int(w[0]) << 64) | int(w[1]), "inc": (int(w[2]) << 64) | int(w[3])}
        st["has_uint32"], st["uinteger"] = 0, 0
        self.rng.bit_generator.state = st

    # ------------------------------------------------------------------ state handling
    def _set_state(self, x):
        if isinstance(x, torch.Tensor):
            xt = x.detach().to(self._tdev, torch.float64)
        else:
            xt = torch.from_numpy(np.array(x, dtype=np.float64)).to(self._tdev)
        self._single = xt.ndim == 1
        if self._single:
            xt = xt[None]
        if xt.ndim != 2 or xt.shape[1] != self.target.d:
            raise ValueError(f"initial_state must be (d,) or (n_chains, d) with d={self.target.d}")
        xt = xt.contiguous()
        n, d = xt.shape
        self.n_chains, self.d = int(n), int(d)
        self._state = torch.empty((d, n), dtype=torch.float64, device=self._tdev)  # component-major
        _lib.check(self._lib.gsss_rows_to_components(xt.data_ptr(), self._state.data_ptr(), n, d, self.device,
                                                     self._stream()))

    def _stream(self):
        return current_stream_ptr(self.device)

    @property
    def state_device(self):
        """Component-major [d, n_chains] CUDA tensor holding the current states (no copy)."""
        return self._state

    def state_rows(self):
        """Current states as an (n_chains, d) CUDA tensor."""
        out = torch.empty((self.n_chains, self.d), dtype=torch.float64, device=self._tdev)
        _lib.check(self._lib.gsss_components_to_rows(self._state.data_ptr(), out.data_ptr(), self.n_chains, self.d,
                                                     self.device, self._stream()))
        return out

    @property
    def state(self):
        x = self.state_rows().cpu().numpy()
        return x[0] if self._single else x

    @state.setter
    def state(self, value):
        n_old = self.n_chains
        self._set_state(value)
        if self.n_chains != n_old:
            raise ValueError("the number of chains is fixed at construction")

    # ------------------------------------------------------------------ counters
    @property
    def n_reject_per_chain(self):
        return self._n_reject.cpu().numpy()

    @property
    def n_tries_per_chain(self):
        return self._n_tries.cpu().numpy()

    @property
    def n_reject(self):
        """Total number of rejected proposals over all chains (mcmc.py:355, :374, :401)."""
        return int(self._n_reject.sum().item())

    @property
    def errors(self):
        """Per-chain GSSS_CHAIN_* bits (0 everywhere in a healthy run)."""
        return self._err.cpu().numpy()

    def _account_calls(self, n_steps):
        """pdf.log_prob.num_calls protocol: one threshold evaluation per step plus one per try
        (calls = steps + tries)."""
        fn = getattr(type(self.target), "log_prob", None)
        if fn is None or not hasattr(fn, "num_calls"):
            return
        tries = int(self._n_tries.sum().item())
        fn.num_calls += n_steps * self.n_chains + (tries - self._tries_reported)
        self._tries_reported = tries

    def _check_errors(self):
        bad = int((self._err != 0).sum().item())
        if bad:
            bits = int(self._err.max().item())
            raise _lib.GsssError(f"{bad} chain(s) stopped with error bits (max {bits}): "
                                 "1=max_tries, 2=non-finite log_prob, 4=replay exhausted, 8=try counter saturated")

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything needed to continue this run elsewhere (plain numpy): states, stream position,
        counters.  The counter-based stream makes a resumed run bit-identical to an uninterrupted one."""
        d = {"state": self.state_rows().cpu().numpy(), "step": self._step, "seed": self.seed,
             "chain_offset": self.chain_offset, "n_reject": self.n_reject_per_chain,
             "n_tries": self.n_tries_per_chain, "err": self.errors, "rng": self.rng_kind,
             "sampler": self._sampler, "mode": self.mode}
        if self._rng_state is not None:
            d["rng_state"] = self._rng_state.cpu().numpy().view(np.uint64)
        return d

    def load_state_dict(self, d):
        """Inverse of state_dict() on a sampler built for the same target and number of chains."""
        if d["state"].shape != (self.n_chains, self.d) or d["rng"] != self.rng_kind:
            raise ValueError("checkpoint does not match this sampler")
        single = self._single
        self._set_state(d["state"])
        self._single = single
        self._step, self.seed, self.chain_offset = int(d["step"]), int(d["seed"]), int(d["chain_offset"])
        self._n_reject.copy_(torch.from_numpy(np.asarray(d["n_reject"], dtype=np.int64)))
        self._n_tries.copy_(torch.from_numpy(np.asarray(d["n_tries"], dtype=np.int64)))
        self._err.copy_(torch.from_numpy(np.asarray(d["err"], dtype=np.int32)))
        self._tries_reported = int(self._n_tries.sum().item())
        if self._rng_state is not None:
            self._rng_state.copy_(torch.from_numpy(np.asarray(d["rng_state"], dtype=np.uint64).view(np.int64)))
            self._sync_rng()

    # ------------------------------------------------------------------ running
    def _launch(self, n_steps, samples=None, thin=1, replay=None, chain_rows=0, samples_ptr=None, stats=False):
        a = _lib.RunArgs()
        a.state_dev = self._state.data_ptr()
        a.samples_dev = samples_ptr if samples_ptr is not None else (samples.data_ptr() if samples is not None else None)
        a.samples_chain_rows = int(chain_rows)
        a.placement = self._placement
        a.n_reject_dev = self._n_reject.data_ptr()
        a.n_tries_dev = self._n_tries.data_ptr()
        a.err_dev = self._err.data_ptr()
        if replay is not None:
            a.replay_dev = replay.data_ptr()
            a.replay_stride = replay.shape[1]
        a.n_chains = self.n_chains
        a.n_steps = int(n_steps)
        a.thin = int(thin)
        a.seed = self.seed
        a.chain_offset = self.chain_offset
        a.step_offset = self._step
        a.sampler = self._sampler
        a.mode = _MODES[self.mode]
        a.max_tries = min(self.max_tries, 2**31 - 1)
        a.variant = self.variant if self.mode != "fast" else (
            _lib.VARIANT_FAST_VERIFY if self.screen == "verify" else (0 if self.screen else _lib.VARIANT_FAST_DOUBLE))
        if self._rng_state is not None:
            if replay is not None:
                raise ValueError("replay and rng='numpy' are mutually exclusive")
            a.rng_state_dev = self._rng_state.data_ptr()
        self._launch_extra(a, int(n_steps))
        if self._stats is not None and stats:
            a.stats_dev = self._stats["acc"].data_ptr()
            a.stats_dirs_dev = self._stats["dirs"].data_ptr()
            a.stats_lags, a.stats_modes, a.stats_flags = self._stats["lags"], self._stats["modes"], self._stats["flags"]
        _lib.check(self._lib.gsss_run(self._target_dev.handle, C.byref(a), self._stream()))
        self._step += int(n_steps)

    def _launch_extra(self, a, n_steps):
        """Hook for samplers with further launch arguments."""

    _sample_buffer = None

    def _begin_sample_buffer(self, out):
        """sample() has allocated its (chains, draws, dims) buffer and filled row 0; the retained rows follow."""
        self._sample_buffer = out

    def advance(self, n_steps, *, thin=None, out=None, replay=None, chain_major=False, row0=0, keep=True):
        """Advance every chain by n_steps transitions on the GPU (asynchronously).

        thin=None keeps nothing; thin=t >= 1 keeps the state after every t-th step and returns a
        CUDA tensor [n_steps // t, d, n_chains] (component-major, the kernels' native layout), or --
        with chain_major=True -- writes rows row0 .. row0 + n_steps//t - 1 of `out`, a contiguous
        (n_chains, R, d) tensor in the reference's (chains, draws, dims) order (every chain appends to
        its own contiguous run; no layout pass afterwards).
        `replay` (n_chains, stride) replays recorded draws instead of the Philox stream.
        keep=False with thin=t stores nothing and only feeds the running statistics (enable_stats) with every
        t-th state; with keep=True they are fed with the very states that are stored.
        """
        n_steps = int(n_steps)
        if n_steps < 0:
            raise ValueError("n_steps must be >= 0")
        if replay is not None:
            if not isinstance(replay, torch.Tensor):
                replay = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(replay), dtype=np.float64))
            replay = replay.to(self._tdev, torch.float64).contiguous()
            if replay.shape[0] != self.n_chains:
                raise ValueError("replay needs one row per chain")
        if thin is None:
            done = 0
            while done < n_steps:
                m = n_steps - done if replay is not None else min(_MAX_STEPS_PER_LAUNCH, n_steps - done)
                self._launch(m, replay=replay)
                done += m
            return None
        thin = int(thin)
        if thin < 1:
            raise ValueError("thin must be >= 1")
        n_keep = n_steps // thin
        if not keep:
            if self._stats is None:
                raise ValueError("keep=False only makes sense with enable_stats()")
            per = max(thin, (_MAX_STEPS_PER_LAUNCH // thin) * thin)
            done = 0
            while done < n_steps:
                m = min(per, n_steps - done)
                self._launch(m, thin=thin, stats=True)
                done += m
            return None
        if chain_major:
            if (out is None or out.ndim != 3 or out.shape[0] != self.n_chains or out.shape[2] != self.d
                    or out.dtype != torch.float64 or not out.is_contiguous() or row0 + n_keep > out.shape[1]):
                raise ValueError("chain_major needs a contiguous float64 out of shape (n_chains, R, d) with room "
                                 "for the rows written")
            total = int(out.shape[1])

            def ptr(r):
                return out.data_ptr() + 8 * self.d * (row0 + r)
        else:
            if out is None:
                out = torch.empty((n_keep, self.d, self.n_chains), dtype=torch.float64, device=self._tdev)
            elif (tuple(out.shape) != (n_keep, self.d, self.n_chains) or out.dtype != torch.float64
                  or not out.is_contiguous()):
                raise ValueError("out must be a contiguous float64 tensor [n_steps//thin, d, n_chains]")
            total = 0

            def ptr(r):
                return out.data_ptr() + 8 * self.d * self.n_chains * r
        if replay is not None:
            self._launch(n_steps, thin=thin, replay=replay, chain_rows=total, samples_ptr=ptr(0), stats=True)
            return out
        per = max(thin, (_MAX_STEPS_PER_LAUNCH // thin) * thin)
        done = 0
        while done < n_steps:
            m = min(per, n_steps - done)
            if m // thin:
                self._launch(m, thin=thin, chain_rows=total, samples_ptr=ptr(done // thin), stats=True)
            else:
                self._launch(m)
            done += m
        return out

    def __iter__(self):
        return self

    def __next__(self):
        """One transition of every chain; returns the new state(s) like the reference (mcmc.py:398-399)."""
        self.advance(1)
        self._account_calls(1)
        self._check_errors()
        self._sync_rng()
        return self.state

    def sample(self, n_samples, burnin=0, return_all_samples=False, *, thin=1, as_tensor=False, blocks=None):
        """Markov chain(s) of the desired size (mcmc.py:55-77): the initial state is row 0 of the
        chain, `n_samples + burnin - 1` transitions are simulated, the first `burnin` rows are
        dropped unless return_all_samples.  One chain -> (n_samples, d); many -> (n_chains, n_samples, d).

        Extensions: thin=t keeps every t-th of the transitions after burn-in (row 0 stays the
        post-burn-in state); as_tensor=True returns a CUDA tensor instead of a numpy array.

        The ndarray is returned at the speed of the PCIe link: it lives in page-locked host memory (geosss_amd/_pinned.py) and,
        when it is large, the ensemble is sampled in `blocks` contiguous blocks of chains -- a block's rows are one contiguous
        run of the (chains, draws, dims) array -- so that the copy of block k runs under the kernel of block k + 1.  The
        streams are keyed by global chain id (or are one generator per chain): the result does not depend on `blocks`
        (None: chosen from the sizes; 1: one launch sequence for all chains, as `as_tensor=True` always does).
        """
        if not n_samples > 0:
            raise AssertionError("n_samples must be positive")  # mcmc.py:62
        burnin = determine_burnin(n_samples, burnin)
        n_rows = n_samples + burnin if return_all_samples else n_samples
        skip = 0 if return_all_samples else burnin
        steps0 = self._step
        if as_tensor:
            out = torch.empty((self.n_chains, n_rows, self.d), dtype=torch.float64, device=self._tdev)
            self._sample_rows(out, skip, n_rows, thin)
        else:
            out = self._sample_to_host(skip, n_rows, thin, blocks)
        self._account_calls(self._step - steps0)
        self._check_errors()
        self._sync_rng()
        return out[0] if self._single else out

    def _sample_rows(self, out, skip, n_rows, thin):
        """burn-in, row 0 = the state after it, then the kept rows, written straight into `out` (chains, draws, dims)."""
        if skip:
            self.advance(skip)
        out[:, 0, :] = self.state_rows()
        self._begin_sample_buffer(out)     # (samplers that keep further per-draw rows lay them out like `out`: _launch_extra)
        try:
            if n_rows > 1:
                self.advance((n_rows - 1) * thin, thin=thin, out=out, chain_major=True, row0=1)
        finally:
            self._sample_buffer = None

    _blockwise = True   # (samplers with further per-chain launch arguments sample all chains in one launch sequence)

    def _plan_blocks(self, skip, n_rows, thin, blocks):
        """How many blocks of chains sample() -> ndarray runs.  The copy is the long pole (10^6 chains x 100 rows on S^2: 2.4 GB,
        ~50 ms; the kernel: 2.5 ms), so as many blocks as keep the sum of the blocks' kernel times -- a block that does not
        fill the chip still takes a resident round's time per transition -- inside the copy time."""
        n = self.n_chains
        nbytes = 8 * n * n_rows * self.d
        if blocks is None:
            env = os.environ.get("GSSS_SAMPLE_BLOCKS")
            blocks = int(env) if env else None
        if blocks is None:
            if nbytes < _PIPELINE_MIN_BYTES or not self._blockwise or self._stats is not None:
                return 1
            copy_s = nbytes / (_COPY_GBS * 1e9)
            block_s = (skip + (n_rows - 1) * thin) * _ROUND_STEP_S
            blocks = int(min(16, nbytes // (32 << 20), copy_s / max(block_s, 1e-6)))
        blocks = max(1, min(int(blocks), n))
        if blocks > 1 and (not self._blockwise or self._stats is not None):
            raise ValueError("this sampler carries per-chain launch state beyond the slice samplers': blocks must be 1")
        return blocks

    @contextlib.contextmanager
    def _chain_block(self, c0, c1, step):
        """The sampler restricted to chains c0 .. c1 - 1 at stream position `step`: counters and generator states are views, the
        states a compact [d, m] copy written back on exit (the kernels index state[j * n_chains + c])."""
        keep = (self._state, self.n_chains, self.chain_offset, self._n_reject, self._n_tries, self._err, self._rng_state, self._step)
        sub = self._state[:, c0:c1].contiguous()
        self._state, self.n_chains, self.chain_offset = sub, c1 - c0, keep[2] + c0
        self._n_reject, self._n_tries, self._err = keep[3][c0:c1], keep[4][c0:c1], keep[5][c0:c1]
        self._rng_state = None if keep[6] is None else keep[6][c0:c1]
        self._step = step
        try:
            yield
            step_end = self._step
        finally:
            (self._state, self.n_chains, self.chain_offset, self._n_reject, self._n_tries, self._err, self._rng_state, self._step) = keep
        self._state[:, c0:c1] = sub
        self._step = step_end

    def _sample_to_host(self, skip, n_rows, thin, blocks):
        n, d = self.n_chains, self.d
        blocks = self._plan_blocks(skip, n_rows, thin, blocks)
        if 8 * n * n_rows * d < _PINNED_MIN_BYTES and blocks == 1:
            # a small array (one chain's README call: 24 KB): locking pages would cost more than the staged copy of a pageable one
            out = torch.empty((n, n_rows, d), dtype=torch.float64, device=self._tdev)
            self._sample_rows(out, skip, n_rows, thin)
            return out.cpu().numpy()
        host = _pinned.empty((n, n_rows, d), self.device)
        lib, dev = self._lib, self.device
        main = torch.cuda.current_stream(dev)
        if blocks == 1:
            out = torch.empty((n, n_rows, d), dtype=torch.float64, device=self._tdev)
            self._sample_rows(out, skip, n_rows, thin)
            _lib.check(lib.gsss_memcpy_d2h_async(host.ctypes.data, out.data_ptr(), host.nbytes, dev, main.cuda_stream))
            main.synchronize()
            return host
        per = -(-n // blocks)
        bufs = [torch.empty((per, n_rows, d), dtype=torch.float64, device=self._tdev) for _ in range(2)]
        copy = torch.cuda.Stream(dev)
        copied = [None, None]                       # the event after which a buffer may be written again
        step0, row_bytes = self._step, 8 * n_rows * d
        for k, c0 in enumerate(range(0, n, per)):
            c1 = min(n, c0 + per)
            buf = bufs[k % 2][:c1 - c0]
            if copied[k % 2] is not None:
                main.wait_event(copied[k % 2])
            with self._chain_block(c0, c1, step0):
                self._sample_rows(buf, skip, n_rows, thin)
            done = torch.cuda.Event()
            done.record(main)
            copy.wait_event(done)
            _lib.check(lib.gsss_memcpy_d2h_async(host.ctypes.data + c0 * row_bytes, buf.data_ptr(), (c1 - c0) * row_bytes, dev, copy.cuda_stream))
            copied[k % 2] = torch.cuda.Event()
            copied[k % 2].record(copy)
        copy.synchronize()
        main.synchronize()
        return host


class ShrinkageSphericalSliceSampler(RejectionSphericalSliceSampler):
    """Geodesic shrinkage slice sampler (mcmc.py:377-401), many chains."""

    _sampler = _lib.SHRINK


class MetropolisHastings(RejectionSphericalSliceSampler):
    """Random-walk Metropolis-Hastings on the sphere for many chains (geosss/mcmc.py:118-176): project into ambient
    space, perturb, project back; `stepsize` adapts during burn-in (AdaptiveStepsize, mcmc.py:80-115), each chain its
    own.  Same constructor and attributes as the reference: `.stepsize`, `.n_accept`, `.reset(burnin)`."""

    _sampler = _lib.RWMH
    _calls_per_step = 2  # log_prob(proposal) and log_prob(state), mcmc.py:152

    _blockwise = False  # (per-chain stepsizes, acceptance counters, momenta: sample() runs all chains in one launch sequence)

    def __init__(self, distribution, initial_state, seed=None, stepsize=1e-1, *, record_stepsize=False, **kwargs):
        if kwargs.get("mode", "exact") not in ("exact", "auto"):
            raise ValueError("RWMH / HMC evaluate log_prob from the point itself: mode='exact'")
        kwargs["mode"] = "exact"
        super().__init__(distribution, initial_state, seed, **kwargs)
        # stepsize after every RWMH proposal, kept on the device per launch ([steps, chains], NaN where the step proposed
        # otherwise): 8 bytes per chain-step, so only on request -- or by default for ONE chain of the mixture sampler below
        self._record_stepsize = bool(record_stepsize)
        self._trace_buf, self._trace_len = None, 0   # one [capacity, n_chains] buffer that doubles: no allocation per launch
        if not float(stepsize) > 0.0:
            raise AssertionError("stepsize must be positive")  # mcmc.py:98
        self._stepsize = torch.full((self.n_chains,), float(stepsize), dtype=torch.float64, device=self._tdev)
        self._n_accept = torch.zeros(self.n_chains, dtype=torch.int64, device=self._tdev)
        self.reset(0)

    def reset(self, burnin):
        """AdaptiveStepsize.reset (mcmc.py:100-103): the next `burnin` transitions adapt the stepsize."""
        self._counter = 0
        self._burnin = int(burnin)

    def _launch_extra(self, a, n_steps):
        a.stepsize_dev = self._stepsize.data_ptr()
        a.n_accept_dev = self._n_accept.data_ptr()
        a.adapt_steps = max(0, min(n_steps, self._burnin - self._counter))
        a.n_reject_dev = None
        a.n_tries_dev = None
        self._counter += n_steps
        if self._record_stepsize and self._sampler != _lib.HMC and n_steps > 0:
            need = self._trace_len + n_steps
            if self._trace_buf is None or need > self._trace_buf.shape[0]:
                grown = torch.empty((max(need, 2 * self._trace_len, 64), self.n_chains), dtype=torch.float64, device=self._tdev)
                if self._trace_len:
                    grown[: self._trace_len] = self._trace_buf[: self._trace_len]
                self._trace_buf = grown
            rows = self._trace_buf[self._trace_len:need]
            rows.fill_(float("nan"))
            a.stepsize_trace_dev = rows.data_ptr()
            self._trace_len = need

    def stepsize_trace(self):
        """[steps, chains] CUDA tensor: the stepsize after every recorded step that made a RWMH proposal, NaN for the others
        (needs record_stepsize=True).  Accumulates over the sampler's lifetime, as the reference's `rwmh_stepsize_vals` list
        does (mcmc.py:201, 228: reset() does not clear it); `clear_stepsize_trace()` drops it."""
        if self._trace_buf is None:
            return torch.empty((0, self.n_chains), dtype=torch.float64, device=self._tdev)
        return self._trace_buf[: self._trace_len]

    def clear_stepsize_trace(self):
        self._trace_buf, self._trace_len = None, 0

    @property
    def stepsize(self):
        e = self._stepsize.cpu().numpy()
        return float(e[0]) if self._single else e

    @stepsize.setter
    def stepsize(self, value):
        self._stepsize.copy_(torch.as_tensor(np.broadcast_to(np.asarray(value, dtype=np.float64), (self.n_chains,)).copy()))

    @property
    def n_accept(self):
        """Accepted proposals, total over chains (mcmc.py:136); per chain: n_accept_per_chain."""
        return int(self._n_accept.sum().item())

    @property
    def n_accept_per_chain(self):
        return self._n_accept.cpu().numpy()

    def _account_calls(self, n_steps):
        fn = getattr(type(self.target), "log_prob", None)
        if fn is not None and hasattr(fn, "num_calls"):
            fn.num_calls += self._calls_per_step * n_steps * self.n_chains

    def sample(self, n_samples, burnin=0, return_all_samples=False, *, thin=1, as_tensor=False):
        """mcmc.py:169-176: the stepsize adapts during the first `burnin` transitions, then Sampler.sample."""
        self.reset(determine_burnin(n_samples, burnin))
        return super().sample(n_samples, burnin, return_all_samples, thin=thin, as_tensor=as_tensor)

    def enable_stats(self, *a, **k):
        raise ValueError("running statistics are accumulated by the slice-sampler kernels")

    def state_dict(self):
        d = super().state_dict()
        d.update(stepsize=self._stepsize.cpu().numpy(), n_accept=self.n_accept_per_chain, counter=self._counter,
                 burnin=self._burnin)
        return d

    def load_state_dict(self, d):
        super().load_state_dict(d)
        self._stepsize.copy_(torch.from_numpy(np.asarray(d["stepsize"], dtype=np.float64)))
        self._n_accept.copy_(torch.from_numpy(np.asarray(d["n_accept"], dtype=np.int64)))
        self._counter, self._burnin = int(d["counter"]), int(d["burnin"])


class IndependenceSampler(MetropolisHastings):
    """Metropolis-Hastings with the uniform distribution on the sphere as proposal (geosss/mcmc.py:179-182), many chains.
    The class inherits RWMH's stepsize adaptation, which runs and changes nothing, exactly as in the reference."""

    _sampler = _lib.INDEP


class MixtureRWMHIndependenceSampler(MetropolisHastings):
    """Mixture of the RWMH kernel (local moves, probability `mixing_probability`) and the independence kernel with uniform
    proposal (global jumps), geosss/mcmc.py:185-234, many chains.  As in the reference only RWMH proposals adapt the
    stepsize, and the burn-in counter only advances on them: a chain adapts during its first `burnin` RWMH proposals.
    Attributes of the reference: `.alpha`, `.n_accept`, `.rwmh_counter`, `.indep_counter` (totals over chains; per chain:
    `rwmh_counter_per_chain`); `.rwmh_stepsize_vals`: the stepsize after every RWMH proposal (mcmc.py:201, 228) -- a list of
    floats for one chain as in the reference, a list of per-chain arrays for many; recorded by default for one chain,
    `record_stepsize=True` for an ensemble (8 bytes per chain-step on the device)."""

    _sampler = _lib.MIX

    def __init__(self, distribution, initial_state, seed=None, stepsize=1e-1, mixing_probability=0.5, *, record_stepsize=None,
                 **kwargs):
        one = np.ndim(initial_state) == 1 or len(initial_state) == 1
        super().__init__(distribution, initial_state, seed, stepsize=stepsize,
                         record_stepsize=one if record_stepsize is None else record_stepsize, **kwargs)
        self.alpha = float(mixing_probability)
        if not 0.0 <= self.alpha <= 1.0:
            raise ValueError("mixing_probability must lie in [0, 1]")
        self._adapt_left = torch.zeros(self.n_chains, dtype=torch.int64, device=self._tdev)
        self._n_rwmh = torch.zeros(self.n_chains, dtype=torch.int64, device=self._tdev)
        self._steps_run = 0

    def reset(self, burnin):
        """AdaptiveStepsize.reset (mcmc.py:100-103); the counter it resets advances on RWMH proposals only (:226-228)."""
        super().reset(burnin)
        if hasattr(self, "_adapt_left"):
            self._adapt_left.fill_(int(burnin))

    def _launch_extra(self, a, n_steps):
        super()._launch_extra(a, n_steps)
        a.adapt_steps = 0
        a.mixing_probability = self.alpha
        a.adapt_left_dev = self._adapt_left.data_ptr()
        a.n_rwmh_dev = self._n_rwmh.data_ptr()
        self._steps_run += n_steps

    @property
    def rwmh_counter_per_chain(self):
        return self._n_rwmh.cpu().numpy()

    @property
    def rwmh_stepsize_vals(self):
        if not self._record_stepsize:
            raise ValueError("the stepsizes of an ensemble are recorded on request: record_stepsize=True")
        t = self.stepsize_trace().cpu().numpy()
        per_chain = [t[~np.isnan(t[:, c]), c] for c in range(self.n_chains)]
        return [float(v) for v in per_chain[0]] if self._single else per_chain

    @property
    def rwmh_counter(self):
        return int(self._n_rwmh.sum().item())

    @property
    def indep_counter(self):
        return self._steps_run * self.n_chains - self.rwmh_counter

    def state_dict(self):
        d = super().state_dict()
        d.update(adapt_left=self._adapt_left.cpu().numpy(), n_rwmh=self.rwmh_counter_per_chain, steps_run=self._steps_run)
        return d

    def load_state_dict(self, d):
        super().load_state_dict(d)
        self._adapt_left.copy_(torch.from_numpy(np.asarray(d["adapt_left"], dtype=np.int64)))
        self._n_rwmh.copy_(torch.from_numpy(np.asarray(d["n_rwmh"], dtype=np.int64)))
        self._steps_run = int(d["steps_run"])


class SphericalHMC(MetropolisHastings):
    """Spherical Hamiltonian Monte Carlo for many chains (geosss/mcmc.py:236-332): `n_steps` leapfrog steps of size
    `stepsize` along great circles, Metropolis correction with the Hamiltonian; needs the target's gradient (device
    functors for the three target families; BinghamFisher's is 2 A x as in the reference, distributions.py:88-114)."""

    _sampler = _lib.HMC

    def __init__(self, distribution, initial_state, seed=None, stepsize=1e-3, n_steps=10, **kwargs):
        super().__init__(distribution, initial_state, seed, stepsize=stepsize, **kwargs)
        self.n_steps = int(n_steps)
        if self.n_steps < 1:
            raise ValueError("n_steps must be >= 1")
        self._momenta = torch.zeros((self.d, self.n_chains), dtype=torch.float64, device=self._tdev)  # mcmc.py:262
        self._momenta_rows = None

    def _launch_extra(self, a, n_steps):
        super()._launch_extra(a, n_steps)
        a.n_leapfrog = self.n_steps
        a.momenta_dev = self._momenta.data_ptr()
        if self._momenta_rows is not None and self._sample_buffer is not None and a.samples_dev:
            # the momenta of the retained rows, laid out like the positions' buffer (same shape, same row offsets)
            a.momenta_samples_dev = self._momenta_rows.data_ptr() + (a.samples_dev - self._sample_buffer.data_ptr())

    @property
    def momenta(self):
        v = self._momenta.T.contiguous().cpu().numpy()
        return v[0] if self._single else v

    @property
    def state(self):
        """[x, v] stacked, as the reference keeps it (mcmc.py:262)."""
        x = self.state_rows().cpu().numpy()
        xv = np.hstack([x, self._momenta.T.cpu().numpy()])
        return xv[0] if self._single else xv

    @state.setter
    def state(self, value):
        value = np.asarray(value, dtype=np.float64)
        if value.shape[-1] == 2 * self.d:
            x, v = value[..., : self.d], value[..., self.d:]
            self._momenta.copy_(torch.from_numpy(np.ascontiguousarray(np.atleast_2d(v).T)))
            value = x
        n_old = self.n_chains
        self._set_state(value)
        if self.n_chains != n_old:
            raise ValueError("the number of chains is fixed at construction")

    def sample(self, n_samples, burnin=0, return_momenta=False, return_all_samples=False, *, thin=1, as_tensor=False):
        """mcmc.py:321-332: the positions, or (positions, momenta) -- the two halves of the reference's [x, v] rows."""
        if not return_momenta:
            return super().sample(n_samples, burnin, return_all_samples, thin=thin, as_tensor=as_tensor)
        if not n_samples > 0:
            raise AssertionError("n_samples must be positive")
        b = determine_burnin(n_samples, burnin)
        n_rows = n_samples + b if return_all_samples else n_samples
        self._momenta_rows = torch.zeros((self.n_chains, n_rows, self.d), dtype=torch.float64, device=self._tdev)
        try:
            x = super().sample(n_samples, burnin, return_all_samples, thin=thin, as_tensor=True)
            v = self._momenta_rows[0] if self._single else self._momenta_rows
        finally:
            self._momenta_rows = None
        return (x, v) if as_tensor else (x.cpu().numpy(), v.cpu().numpy())

    def _begin_sample_buffer(self, out):
        super()._begin_sample_buffer(out)
        if self._momenta_rows is not None:  # row 0: the momenta of the state the retained rows start from (mcmc.py:67, 262)
            self._momenta_rows[:, 0, :] = self._momenta.T

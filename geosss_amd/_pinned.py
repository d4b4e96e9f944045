"""Page-locked host memory for the arrays `sampler.sample()` returns.

The reference's `Sampler.sample` returns an ndarray (geosss/mcmc.py:55-77) and every script reads it on the host
(scripts/curve_vMF.py:119-120).  At 10^6 chains that array is gigabytes; a device-to-host copy into ordinary (pageable) memory
is staged by the driver at ~9 GB/s, a sixth of the PCIe link.  So the returned array LIVES in page-locked memory
(`gsss_malloc_host`): the device writes into it directly, block of chains by block of chains, while the next block is being
sampled.  Locking pages costs about as much as touching them for the first time, so blocks go back to a small pool when the array
(and every view of it) is dropped and the next `sample()` of the same size reuses them.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib

# cached (free) bytes the pool may hold before it gives blocks back to the system
POOL_BYTES = int(os.environ.get("GSSS_PINNED_POOL_BYTES", str(8 << 30)))
_lock = threading.Lock()
_free = []  # [(bytes, address)]


def _take(nbytes):
    with _lock:
        best = None
        for i, (size, _) in enumerate(_free):
            if nbytes <= size <= nbytes + (nbytes >> 2) + (1 << 20) and (best is None or size < _free[best][0]):
                best = i
        if best is not None:
            return _free.pop(best)
    return None


def _give_back(size, addr):
    lib = _lib.load()
    with _lock:
        _free.append((size, addr))
        total = sum(s for s, _ in _free)
        drop = []
        while total > POOL_BYTES and _free:      # oldest first
            s, a = _free.pop(0)
            total -= s
            drop.append(a)
    for a in drop:
        lib.gsss_free_host(C.c_void_p(a))


def trim():
    """Give every cached block back to the system."""
    lib = _lib.load()
    with _lock:
        drop = [a for _, a in _free]
        _free.clear()
    for a in drop:
        lib.gsss_free_host(C.c_void_p(a))


class _Block:
    """Owner of one page-locked allocation; numpy arrays made from it keep it alive through `.base`."""

    def __init__(self, nbytes, device):
        got = _take(nbytes)
        if got is None:
            p = C.c_void_p()
            _lib.check(_lib.load().gsss_malloc_host(C.byref(p), max(1, nbytes), device))
            got = (max(1, nbytes), p.value)
        self.size, self.addr = got
        self.nbytes = nbytes

    def array(self, shape):
        self.__array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (self.addr, False), "version": 3}
        return np.asarray(self)

    def __del__(self):
        addr, self.addr = getattr(self, "addr", None), None
        if addr:
            try:
                _give_back(self.size, addr)
            except Exception:  # interpreter shutdown
                pass


_warned = False


def empty(shape, device=0):
    """A float64 ndarray of `shape` in page-locked host memory (C-contiguous).  Dropping the array (and its views) returns the
    memory to the pool.  Where the system refuses to lock that many pages (a memlock limit) the array is an ordinary one: the
    copies into it are then staged by the driver -- slower, the same bytes."""
    global _warned
    shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(v) for v in shape)
    nbytes = 8 * int(np.prod(shape, dtype=np.int64)) if shape else 8
    try:
        return _Block(nbytes, device).array(shape)
    except _lib.GsssError as e:
        trim()                                   # cached blocks count against the same limit: give them back and try once more
        try:
            return _Block(nbytes, device).array(shape)
        except _lib.GsssError:
            if not _warned:
                import warnings
                warnings.warn(f"geosss_amd: {nbytes} bytes of page-locked host memory were refused ({e}); sample() returns a "
                              "pageable array (the device-to-host copy runs at a fraction of the link's rate)", RuntimeWarning)
                _warned = True
            return np.empty(shape, dtype=np.float64)

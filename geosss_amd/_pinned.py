"""Page-locked host memory for the arrays `sampler.sample()` returns.

The reference's `Sampler.sample` returns an ndarray (geosss/mcmc.py:55-77) and every script reads it on the host
(scripts/curve_vMF.py:119-120).  At 10^6 chains that array is gigabytes; a device-to-host copy into ordinary (pageable) memory
is staged by the driver at ~11 GB/s, a fifth of the PCIe link.  So the returned array LIVES in page-locked memory: the device
writes into it directly, block of chains by block of chains, while the next block is being sampled.

Locking pages is cheap (4 ms for 2.4 GB) -- what costs is FAULTING them in, which `hipHostMalloc` / `hipHostRegister` do one page
after the other inside the call (105 ms).  So the memory is an ordinary numpy allocation whose pages are touched from several
threads first (11 ms on 16 threads) and locked where they lie (`gsss_host_register`; tools/microbench_pinning.py has the
measurements).  Blocks go back to a small pool when the array (and every view of it) is dropped, and the next `sample()` of the
same size reuses them with nothing to fault or lock.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib

# cached (free) bytes the pool may hold before it gives blocks back to the system
POOL_BYTES = int(os.environ.get("GSSS_PINNED_POOL_BYTES", str(8 << 30)))
_PAGE = 4096
_lock = threading.Lock()
_free = []  # [_Memory]


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


class _Memory:
    """One page-locked region: a numpy byte buffer (which owns the pages) registered with the device from its first page boundary."""

    def __init__(self, nbytes, device):
        size = (max(1, nbytes) + _PAGE - 1) // _PAGE * _PAGE
        self.buf = np.empty(size + _PAGE, dtype=np.uint8)
        self.addr = (self.buf.ctypes.data + _PAGE - 1) // _PAGE * _PAGE
        self.size = size
        off = self.addr - self.buf.ctypes.data
        view = self.buf[off: off + size]
        k = _threads() if size >= (64 << 20) else 1

        def touch(i):                       # one write per page: the fault is the cost (numpy releases the GIL for the strided store)
            view[i * size // k: (i + 1) * size // k: _PAGE] = 0

        if k == 1:
            touch(0)
        else:
            workers = [threading.Thread(target=touch, args=(i,)) for i in range(k)]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
        _lib.check(_lib.load().gsss_host_register(C.c_void_p(self.addr), size, device))
        self.registered = True

    def release(self):
        if getattr(self, "registered", False):
            self.registered = False
            try:
                _lib.load().gsss_host_unregister(C.c_void_p(self.addr))
            except Exception:  # interpreter shutdown
                pass
        self.buf = None

    def __del__(self):
        self.release()


def _take(nbytes):
    with _lock:
        best = None
        for i, m in enumerate(_free):
            if nbytes <= m.size <= nbytes + (nbytes >> 2) + (1 << 20) and (best is None or m.size < _free[best].size):
                best = i
        if best is not None:
            return _free.pop(best)
    return None


def _give_back(mem):
    with _lock:
        _free.append(mem)
        total = sum(m.size for m in _free)
        drop = []
        while total > POOL_BYTES and _free:      # oldest first
            m = _free.pop(0)
            total -= m.size
            drop.append(m)
    for m in drop:
        m.release()


def trim():
    """Give every cached block back to the system."""
    with _lock:
        drop = list(_free)
        _free.clear()
    for m in drop:
        m.release()


class _Block:
    """Owner of one page-locked region while an array made from it is alive; numpy arrays keep it alive through `.base`."""

    def __init__(self, nbytes, device):
        self.mem = _take(nbytes) or _Memory(nbytes, device)
        self.nbytes = nbytes

    def array(self, shape):
        self.__array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (self.mem.addr, False), "version": 3}
        return np.asarray(self)

    def __del__(self):
        mem, self.mem = getattr(self, "mem", None), None
        if mem is not None:
            try:
                _give_back(mem)
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

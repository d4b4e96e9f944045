#!/usr/bin/env python3
"""How fast can 2.4 GB of host memory become page-locked?  hipHostMalloc against np.empty + hipHostRegister, with and without
touching the pages first from several threads.  GPU box: python tools/microbench_pinning.py"""
import ctypes as C, time, threading, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
N = 2_400_000_000

def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r

def malloc():
    p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), N, 0) == 0; return p
for _ in range(2):
    dt, p = t(malloc); print(f"hipHostMalloc 2.4 GB: {dt*1e3:.1f} ms"); hip.hipHostFree(p)

def reg_fresh():
    a = np.empty(N, dtype=np.uint8); assert hip.hipHostRegister(a.ctypes.data, N, 0) == 0; return a
for _ in range(2):
    dt, a = t(reg_fresh); print(f"np.empty + hipHostRegister (untouched pages): {dt*1e3:.1f} ms"); hip.hipHostUnregister(a.ctypes.data); del a

def touch(a, k):
    def w(i):
        lo, hi = i * N // k, (i + 1) * N // k
        a[lo:hi:4096] = 0
    th = [threading.Thread(target=w, args=(i,)) for i in range(k)]
    [x.start() for x in th]; [x.join() for x in th]
for k in (1, 4, 16):
    a = np.empty(N, dtype=np.uint8)
    dt1, _ = t(lambda: touch(a, k))
    dt2, _ = t(lambda: hip.hipHostRegister(a.ctypes.data, N, 0))
    print(f"touch with {k} threads: {dt1*1e3:.1f} ms, then hipHostRegister: {dt2*1e3:.1f} ms")
    hip.hipHostUnregister(a.ctypes.data); del a
# per-block registration (8 blocks)
a = np.empty(N, dtype=np.uint8)
blk = (N // 8) // 4096 * 4096
base = (a.ctypes.data + 4095) // 4096 * 4096
t0 = time.perf_counter()
for i in range(7):
    assert hip.hipHostRegister(base + i * blk, blk, 0) == 0
print(f"7 blocks of {blk/1e6:.0f} MB registered one by one: {(time.perf_counter()-t0)*1e3:.1f} ms")

"""Accuracy of the sampler's own elementary functions (geosss_amd/csrc/gsss_math.h), measured
on a host build of the same header against numpy/libm.  The functions are pure FMA arithmetic, so
the device evaluates the same operations (the GPU parity tests cover the device side end to end)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "math_host.cpp")
HDR = os.path.join(ROOT, "geosss_amd", "csrc", "gsss_math.h")
OUT = os.path.join(ROOT, "tests", "_build", "libmath_host.so")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                               "-o", OUT, SRC])
    return C.CDLL(OUT)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_sincos_small(lib):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, 400000), np.linspace(-8, 8, 100001),
                        np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, -2 * np.pi, 1e-300, 1e-9])])
    s, c = np.empty_like(x), np.empty_like(x)
    lib.t_sincos_small(_p(x), C.c_long(len(x)), _p(s), _p(c))
    xl = x.astype(np.longdouble)
    ws, wc = np.sin(xl).astype(np.float64), np.cos(xl).astype(np.float64)
    # absolute error near zeros of sin/cos is what matters for y = cos*x + sin*u
    assert np.max(np.abs(s - ws)) < 2.5e-16 and np.max(np.abs(c - wc)) < 2.5e-16
    big = np.abs(ws) > 0.1
    assert np.max(ulps(s[big], ws[big])) <= 2.0
    big = np.abs(wc) > 0.1
    assert np.max(ulps(c[big], wc[big])) <= 2.0


def test_sincos_2pi(lib):
    rng = np.random.default_rng(2)
    u = np.concatenate([rng.random(400000), np.array([0.0, 0.125, 0.25, 0.5, 0.75, 1 - 2.0**-53])])
    s, c = np.empty_like(u), np.empty_like(u)
    lib.t_sincos_2pi(_p(u), C.c_long(len(u)), _p(s), _p(c))
    a = 2 * np.longdouble("3.14159265358979323846264338327950288") * u.astype(np.longdouble)
    assert np.max(np.abs(s - np.sin(a).astype(np.float64))) < 2.5e-16
    assert np.max(np.abs(c - np.cos(a).astype(np.float64))) < 2.5e-16
    assert s[-6] == 0.0 and c[-6] == 1.0  # u = 0


def test_exp(lib):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-745, 709, 400000), rng.uniform(-5, 5, 200000), np.linspace(-1, 1, 20001),
                        np.array([0.0, -0.0, 1.0, -1.0, 709.7, -744.0])])
    y = np.empty_like(x)
    lib.t_exp(_p(x), C.c_long(len(x)), _p(y))
    want = np.exp(x.astype(np.longdouble)).astype(np.float64)
    normal = want > 1e-300
    assert np.max(ulps(y[normal], want[normal])) <= 1.5
    tiny = ~normal  # near / below the normal range: 1 ulp of a subnormal is coarse, allow it
    assert np.all(np.abs(y[tiny] - want[tiny]) <= np.maximum(4e-16 * want[tiny], 1e-323))
    yb = np.empty_like(x)
    lib.t_exp_bounded(_p(x), C.c_long(len(x)), _p(yb))
    assert np.array_equal(yb, y)  # same polynomial, same rounding on the common domain
    big = np.array([-5000.0, 5000.0, -1e5, 1e5])
    yb = np.empty_like(big)
    lib.t_exp_bounded(_p(big), C.c_long(len(big)), _p(yb))
    assert yb[0] == 0.0 and yb[1] == np.inf and yb[2] == 0.0 and yb[3] == np.inf
    e = np.array([-np.inf, np.inf, -1e4, 1e4, np.nan])
    y = np.empty_like(e)
    lib.t_exp(_p(e), C.c_long(len(e)), _p(y))
    assert y[0] == 0.0 and y[1] == np.inf and y[2] == 0.0 and y[3] == np.inf and np.isnan(y[4])


def test_log(lib):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.random(400000), (rng.integers(0, 2**32, 200000) + 1.0) / 2.0**32,
                        np.exp(rng.uniform(-700, 700, 100000)), np.array([1.0, 0.5, 2.0, 2.0**-32, 1 - 2.0**-53])])
    y = np.empty_like(x)
    lib.t_log(_p(x), C.c_long(len(x)), _p(y))
    want = np.log(x.astype(np.longdouble)).astype(np.float64)
    nz = want != 0
    assert np.max(ulps(y[nz], want[nz])) <= 1.5
    assert np.all(y[~nz] == 0.0)
    e = np.array([0.0, -1.0, np.nan])
    y = np.empty_like(e)
    lib.t_log(_p(e), C.c_long(len(e)), _p(y))
    assert y[0] == -np.inf and np.isnan(y[1]) and np.isnan(y[2])


def test_table_driven_functions(lib):
    """The table-driven sincos / log of the throughput kernels (64 circle points, 129 log points, short polynomials):
    as accurate as the polynomial versions."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, 400000), np.linspace(-8, 8, 100001),
                        np.array([0.0, np.pi / 64, -np.pi / 64, np.pi / 2, np.pi, 2 * np.pi, -2 * np.pi, 1e-300, 1e-9])])
    s, c = np.empty_like(x), np.empty_like(x)
    lib.t_sincos_tab(_p(x), C.c_long(len(x)), _p(s), _p(c))
    xl = x.astype(np.longdouble)
    assert np.max(np.abs(s - np.sin(xl).astype(np.float64))) < 3.4e-16
    assert np.max(np.abs(c - np.cos(xl).astype(np.float64))) < 3.4e-16
    w = np.concatenate([rng.integers(0, 2**32, 400000, dtype=np.uint64), np.array([0, 1, 2**25 - 1, 2**25, 2**26, 2**31, 2**32 - 1])]).astype(np.uint32)
    s, c = np.empty(len(w)), np.empty(len(w))
    lib.t_sincos_word_tab(_p(w), C.c_long(len(w)), _p(s), _p(c))
    a = 2 * np.longdouble("3.14159265358979323846264338327950288") * (w.astype(np.longdouble) / np.longdouble(2.0**32))
    assert np.max(np.abs(s - np.sin(a).astype(np.float64))) < 3.4e-16
    assert np.max(np.abs(c - np.cos(a).astype(np.float64))) < 3.4e-16
    assert s[-7] == 0.0 and c[-7] == 1.0                                                    # w = 0
    y = np.empty(len(w))
    lib.t_log_word_tab(_p(w), C.c_long(len(w)), _p(y))
    want = np.log((w.astype(np.longdouble) + 1) / np.longdouble(2.0**32)).astype(np.float64)
    nz = want != 0
    assert np.max(np.abs(y[nz] - want[nz]) / np.abs(want[nz])) < 1e-15
    assert y[-1] == 0.0                                                                     # w = 2^32 - 1 -> log 1


def test_single_precision_box_muller_host_oracle_bit_identical(lib):
    """fm::box_muller_f32 (gsss_math.h, what the kernels run) and the oracle's restatement gor_box_muller32 are the same
    IEEE single-precision operations: identical bits on 2 million word pairs incl. the corners; and the pair is the
    Box-Muller transform to single precision (radius and angle against libm in double)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(9)
    n = 2_000_000
    wr = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    wa = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    corners = np.array([0, 1, 2, 2**23, 2**24 - 1, 2**24, 2**29 - 1, 2**29, 2**31, 2**32 - 2, 2**32 - 1], dtype=np.uint32)
    wr[:len(corners)] = corners
    wa[len(corners):2 * len(corners)] = corners
    a0, a1 = np.empty(n), np.empty(n)
    lib.t_box_muller_f32(_p(wr), _p(wa), C.c_long(n), _p(a0), _p(a1))
    b0, b1 = np.empty(n), np.empty(n)
    orc.lib().gor_box_muller32_fill(_p(wr), _p(wa), C.c_int64(n), _p(b0), _p(b1))
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
    r = np.sqrt(-2.0 * np.log(((wr >> 8).astype(np.float64) + 1.0) / 2.0**24))
    ang = 2.0 * np.pi * wa.astype(np.float64) / 2.0**32
    assert np.max(np.abs(a0 - r * np.cos(ang))) < 4e-6 and np.max(np.abs(a1 - r * np.sin(ang))) < 4e-6
    assert np.max(np.abs(np.hypot(a0, a1) - r) / np.maximum(r, 1e-3)) < 1e-6
    assert abs(a0.mean()) < 3e-3 and abs(a0.var() - 1) < 5e-3 and abs(np.mean(a0 * a1)) < 3e-3

"""The constants the single-precision screen's rigour rests on, re-derived on the box that runs the tests.

Fast mode decides four tries out of five on v_sin_f32 / v_cos_f32 / v_exp_f32 / v_log_f32 / v_sqrt_f32 with an error margin formed
from four constants (geosss_amd/csrc/gsss_screen_consts.h).  A try the screen calls certain is decided as the double-precision
test of geosss/mcmc.py:397 (`if p(y) > threshold`) would ONLY if each instruction's worst-case error is inside its constant.
Rounds 2-4 took those errors from one hand-run microbenchmark; here `gsss_f32_error_sweep` evaluates EVERY float of each
instruction's argument range (2^25 + 2^32 + 2^24 + 2^31 floats) on the device against double precision, the rounding of the
double argument to single included where the kernels round one, and the measured maxima must lie inside the constants
compiled into the loaded library (`gsss_screen_constants`)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _constants():
    from geosss_amd import _lib
    out = (C.c_double * 5)()
    _lib.check(_lib.load().gsss_screen_constants(out, 5))
    return dict(zip(("sincos", "exp2", "log2", "sqrt", "unit"), out))


def test_constants_are_the_headers():
    """(CPU) the exported constants are the literals of gsss_screen_consts.h -- the one place the kernels read them from."""
    text = open(os.path.join(ROOT, "geosss_amd", "csrc", "gsss_screen_consts.h")).read()
    lit = {m.group(1): float(m.group(2)) for m in re.finditer(r"constexpr float (k\w+) = ([0-9.e+-]+)f;", text)}
    k = _constants()
    import numpy as np
    for name, key in (("kSinCosErr32", "sincos"), ("kExp2Err32", "exp2"), ("kLog2Err32", "log2"), ("kSqrtRelErr32", "sqrt"), ("kUnit32", "unit")):
        assert float(np.float32(lit[name])) == k[key], name
    assert k["unit"] == 2.0 ** -24
    # no kernel source carries its own copy of an error constant
    for f in os.listdir(os.path.join(ROOT, "geosss_amd", "csrc")):
        if f.endswith((".h", ".hip")) and f != "gsss_screen_consts.h":
            assert not re.search(r"constexpr float k(SinCos|Exp2|Log2|SqrtRel)Err32", open(os.path.join(ROOT, "geosss_amd", "csrc", f)).read()), f


def _sweep(which):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from geosss_amd import _lib
    _lib.require_device()
    out, n = (C.c_double * 4)(), C.c_uint64(0)
    _lib.check(_lib.load().gsss_f32_error_sweep(which, out, C.byref(n), 0, None))
    return list(out), int(n.value)


@pytest.mark.gpu
def test_sin_cos_of_every_float_in_one_revolution():
    k = _constants()
    (e_sin, e_cos, full_sin, full_cos), n = _sweep(0)
    assert n == 2 * (0x3F800000 + 1)                       # every float of [0, 1] and of [-1, -0]
    print(f"v_sin_f32 {e_sin:.4e}, v_cos_f32 {e_cos:.4e} at the float; with the argument's rounding {full_sin:.4e}, {full_cos:.4e}; "
          f"kSinCosErr32 = {k['sincos']:.4e}")
    assert 0 < e_sin < 2e-7 and 0 < e_cos < 2e-7
    # theta / 2 pi is formed in double precision before it is rounded: 1e-15 covers that product
    assert max(full_sin, full_cos) + 1e-15 <= k["sincos"]


@pytest.mark.gpu
def test_exp2_of_every_finite_float():
    k = _constants()
    (rel, sub, not_inf, _), n = _sweep(1)
    assert n == 2 * (0x7F7FFFFF + 1)
    print(f"v_exp_f32: relative {rel:.4e} on normal results (kExp2Err32 = {k['exp2']:.4e}), absolute {sub:.4e} below the normals")
    assert 0 < rel <= k["exp2"]
    assert sub <= 1.1754943508222875e-38                   # a flushed result is off by less than the smallest normal
    assert not_inf == 0


@pytest.mark.gpu
def test_log2_of_every_mantissa():
    k = _constants()
    (at_float, full, _, _), n = _sweep(2)
    assert n == 0x40000000 - 0x3F000000 + 1                # every float of [0.5, 2]
    print(f"v_log_f32 {at_float:.4e} at the float on [0.5, 2]; with the mantissa's rounding on [0.5, 1] {full:.4e}; kLog2Err32 = {k['log2']:.4e}")
    assert 0 < at_float and full <= k["log2"]


@pytest.mark.gpu
def test_sqrt_of_every_positive_normal_float():
    k = _constants()
    (rel, _, _, _), n = _sweep(3)
    assert n == 0x7F7FFFFF - 0x00800000 + 1
    print(f"v_sqrt_f32: relative {rel:.4e} (kSqrtRelErr32 = {k['sqrt']:.4e})")
    assert 0 < rel <= k["sqrt"]

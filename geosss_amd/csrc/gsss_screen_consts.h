// gsss_screen_consts.h -- worst-case errors of the hardware single-precision instructions the screened kernels bound a try's
// level with (gsss_screen.h, gsss_curvespec.h; the decision they protect: geosss/mcmc.py:397 `if p(y) > threshold`).
//
// These four numbers are what makes the screen RIGOROUS: a try the single-precision evaluation calls "certainly rejected" or
// "certainly accepted" is decided as the double-precision test would only if every instruction's error is inside its constant.
// They are not taken on trust: gsss_f32_error_sweep (gsss_verify.hip) evaluates EVERY float of each instruction's argument
// range on the device against double precision, with the rounding of the double argument to single included where the
// kernels round one, and tests/test_hip_screen_bounds.py holds the measured maxima to the constants compiled in here
// (gsss_screen_constants) on every GPU test run.
#pragma once

namespace gsss {

// |v_cos_f32(fl32(theta / 2 pi)) - cos(theta)| for |theta| <= 2 pi: the hardware's error at the rounded argument (1.254e-7,
// exhaustive sweep) + 2 pi x half an ulp of the argument in revolutions (|t| <= 1: 2 pi 2^-25 = 1.873e-7); same for sin
constexpr float kSinCosErr32 = 3.5e-7f;
constexpr float kUnit32 = 5.9604645e-8f;  // 2^-24
constexpr float kExp2Err32 = 8.5e-8f;     // relative error of v_exp_f32 on normal results (exhaustive sweep)
// log2 of a double via frexp + v_log_f32 of the mantissa m in [0.5, 1) rounded to single: the hardware's error at the rounded
// mantissa (5.95e-8) + half an ulp of it over m ln 2 (8.6e-8 at m = 0.5); the exhaustive sweep of the SUM gives 1.4267e-7.
// (Rounds 2-4 had 1.3e-7 here -- the two maxima taken apart and the second misjudged; the 25 % the margins carry on top covered
// it, the first run of the sweep as a test found it.)
constexpr float kLog2Err32 = 1.45e-7f;
// relative error of v_sqrt_f32 on normal arguments (the curve screen's |P y|; inside the "40 x 2^-24 of arithmetic" of
// Curve32::eval_error, which allows 2^-22 for it)
constexpr float kSqrtRelErr32 = 2.3841858e-7f;  // 2^-22

}  // namespace gsss

// ll_fast.h -- log(1 + exp(-a)) for the elliptical-slice kernels (rng_ess.hip), host- and device-compilable.
//
// src/log-likelihood.cpp:20,34 writes the term as log(1 + exp(-a)); through the device library that is 141 vector
// instructions per evaluation (exp 42, log 98: double-double inside), and draw_f evaluates it 8192 x 1024 x (2 + k) ~ 39 M
// times per iteration -- the slice kernel was within 2x of that instruction count's throughput bound.  This form keeps
// the library's exp and replaces the logarithm:
//     log(1 + e^-a) = max(-a, 0) + log1p(t),   t = exp(-|a|) in [0, 1]
//     1 + t = 2^c (1 + u)  with  c = 1, u = (t - 1) / 2  for t > sqrt(2) - 1,  c = 0, u = t  otherwise:  |u| <= 0.4143
//     log1p(u) = 2 atanh(s),  s = u / (2 + u),  |s| <= 0.1716:  2 s (1 + z/3 + z^2/5 + ... + z^10/21),  z = s^2
// (truncation 0.0295^11 / 23 = 6e-19 relative).  ~70 vector instructions.  Against the exact value it is within 2 ulp
// (tests/test_ll_fast.py, against long double); against the formula AS WRITTEN it differs by the rounding of 1 + exp(-a)
// that the written form commits and this one does not: at most 1.2e-16 ABSOLUTE per term (for a > 37 the written form
// returns 0, this one e^-a).  Overflow: the written form gives +inf for a < -709.78 (exp overflows), this one -a; the
// slice then compares a huge finite number instead of -inf with log_y and rejects all the same.  NaN stays NaN.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define GP_LL_HD __host__ __device__
#else
#define GP_LL_HD
#endif

GP_LL_HD inline double gp_ll_rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);
#else
    return (double)(1.0f / (float)d);      // a single-precision seed, like the hardware estimate
#endif
}

GP_LL_HD inline double ll_term_fast(double a)
{
    const double x = fabs(a);
    const double t = exp(-x);
    const bool big = t > 0.41421356237309503;
    const double u = big ? fma(0.5, t, -0.5) : t;
    const double d = 2.0 + u;
    double r = gp_ll_rcp(d);                                   // 1 / d: estimate + two Newton steps
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = u * r;                                          // u / d, one correction
    s = fma(fma(-d, s, u), r, s);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double s2 = s + s;
    double l = fma(s2 * z, p, s2);                             // log1p(u)
    if (big) l = 0.693147180369123816490 + (l + 1.90821492927058770002e-10);      // + ln 2 (hi + lo)
    return (a < 0.0 ? x : 0.0) + l;
}

// ---- a SCREEN for the slice loop's accept test (rng_ess.hip, ess_kernel_reg) ----------------------------------------------
// The loop only needs the SIGN of  ll_bar(f') - log_y  (src/draw-f.cpp:45): the same term in single precision through the
// hardware's exp2 / log2 (v_exp_f32, v_log_f32: ~12 instructions instead of 91) decides it whenever the sum is further from
// log_y than the screen's error bound; only inside that band the pass is repeated with the full-precision term, so every
// accept / reject decision -- and with it every draw -- is the full-precision one.  Error of one screened term against
// ll_term_fast: the part max(-a, 0) is exact (fp64); t = -|a| -> float (2^-24 relative), exp(t) <= 1 with relative error
// <= 1.2e-7 + 1.8e-7 |t| (argument product + 1 ulp), log(1 + e) with absolute error <= 3e-7: below 5e-7 absolute in all.
// (ess_kernel_reg screens BOTH sums of the test -- the trial point's and the current state's -- and decides by their
// difference against log(u) with twice the band; the full-precision pair runs only inside it.)
// LL_SCREEN_ERR is eight times that; tests/test_ll_fast.py measures the actual maximum on the device (gpirt_debug_ll_term,
// form 2) over four million points (1.2e-7) and holds it under LL_SCREEN_ERR / 4.
#define LL_SCREEN_ERR 4.0e-6
#if defined(__HIPCC__)
__device__ __forceinline__ double ll_term_screen(double a)
{
    const double x = fabs(a);
    const float e = __expf(-(float)x);
    return (a < 0.0 ? x : 0.0) + (double)__logf(1.0f + e);
}
#endif

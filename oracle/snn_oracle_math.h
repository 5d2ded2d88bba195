/*
 * oracle/snn_oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Portable transcendental helpers of the CPU oracle.
 *
 * The reference computes `f32::exp` and `f32::powf` through Rust `std`, i.e.
 * the platform libm (glibc `expf` / `powf`; call sites e.g.
 * backend/src/neuron/ion_channels/mod.rs:224-228,234,270-271,280,
 * iterate_and_spike/mod.rs:149,1133, plasticity/mod.rs:52-54,
 * spike_train/mod.rs:85).  glibc's implementations are not under
 * /root/reference (third-party, un-vendored) and are not bit-reproducible on a
 * GPU, so the oracle and the HIP product each carry an own, independently
 * written implementation of ONE published algorithm:
 *
 *   exp(x) = 2^k * P(r),  k = rint(x / ln2),  r = x - k*ln2  (|r| <= ln2/2),
 *   P = degree-13 Taylor polynomial evaluated by Horner's rule in IEEE
 *   binary64 with plain mul/add (no FMA), then ONE rounding to binary32.
 *
 * Truncation error < 1e-17 relative, so the binary32 result is the correctly
 * rounded one except when exp(x) lies within ~1e-16 relative of a rounding
 * boundary.  glibc expf documents <= 0.502 ULP, so the two agree bit-for-bit
 * on all but a small fraction of inputs and never differ by more than 1 ULP
 * (tests/test_oracle_math.py measures both against the container's libm).
 *
 * pow(x, 3) and pow(x, 4) are evaluated in binary64 (x*x is exact there) and
 * rounded once to binary32 -- again the correctly rounded value up to double
 * rounding, which is what glibc powf returns (<= 0.52 ULP documented).
 *
 * Build flags that matter: -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#ifndef SNN_ORACLE_MATH_H
#define SNN_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>

/* exp of a binary64 argument, |x| < 700: the polynomial core of snn_o_expf and of the hyperbolic functions below */
static inline double snn_o_exp_core(double xd)
{
    const double INV_LN2 = 1.4426950408889634;       /* 0x3FF71547652B82FE */
    const double LN2_HI  = 6.93147180369123816490e-01; /* 0x3FE62E42FEE00000 */
    const double LN2_LO  = 1.90821492927058770002e-10; /* 0x3DEA39EF35793C76 */
    const double SHIFT   = 6755399441055744.0;        /* 1.5 * 2^52 */

    double kd = (xd * INV_LN2 + SHIFT) - SHIFT;   /* rint under round-to-nearest */
    double r  = (xd - kd * LN2_HI) - kd * LN2_LO;

    /* Horner, coefficients 1/n! */
    double p = 1.6059043836821613e-10;            /* 1/13! */
    p = p * r + 2.08767569878681e-09;             /* 1/12! */
    p = p * r + 2.505210838544172e-08;            /* 1/11! */
    p = p * r + 2.755731922398589e-07;            /* 1/10! */
    p = p * r + 2.7557319223985893e-06;           /* 1/9!  */
    p = p * r + 2.48015873015873e-05;             /* 1/8!  */
    p = p * r + 1.984126984126984e-04;            /* 1/7!  */
    p = p * r + 1.388888888888889e-03;            /* 1/6!  */
    p = p * r + 8.333333333333333e-03;            /* 1/5!  */
    p = p * r + 4.1666666666666664e-02;           /* 1/4!  */
    p = p * r + 1.6666666666666666e-01;           /* 1/3!  */
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;

    int64_t k = (int64_t)kd;                      /* |k| <= 1010 */
    uint64_t bits = (uint64_t)(k + 1023) << 52;   /* 2^k as a normal double */
    double scale;
    memcpy(&scale, &bits, sizeof scale);
    return p * scale;
}

static inline float snn_o_expf(float x)
{
    if (!(x == x)) return x;                 /* NaN in, NaN out */
    if (x > 89.0f) return __builtin_inff();  /* expf overflows above 88.72 */
    if (x < -104.0f) return 0.0f;            /* below half the least subnormal */
    return (float)snn_o_exp_core((double)x);
}

/* f32::tanh / sinh / cosh for generated models (build_test/nb_macro/src/lib.rs:9152-9163 forward to libm):
 * binary64 through the exp core, Taylor series where e^x - e^-x cancels, one rounding to binary32. */
static inline float snn_o_tanhf(float x)
{
    if (!(x == x)) return x;
    double d = (double)x, a = (d < 0.0) ? -d : d, t;
    if (a < 0.05) {
        double z = a * a;
        double p = 62.0 / 2835.0;
        p = p * z - 17.0 / 315.0;
        p = p * z + 2.0 / 15.0;
        p = p * z - 1.0 / 3.0;
        p = p * z + 1.0;
        t = a * p;
    } else if (a > 20.0) {
        t = 1.0;
    } else {
        t = 1.0 - 2.0 / (snn_o_exp_core(2.0 * a) + 1.0);
    }
    return (float)((d < 0.0) ? -t : t);
}

static inline float snn_o_sinhf(float x)
{
    if (!(x == x)) return x;
    double d = (double)x, a = (d < 0.0) ? -d : d, t;
    if (a < 0.05) {
        double z = a * a;
        double p = 1.0 / 5040.0;
        p = p * z + 1.0 / 120.0;
        p = p * z + 1.0 / 6.0;
        p = p * z + 1.0;
        t = a * p;
    } else if (a > 90.0) {
        t = (double)__builtin_inff();
    } else {
        double e = snn_o_exp_core(a);
        t = (e - 1.0 / e) * 0.5;
    }
    return (float)((d < 0.0) ? -t : t);
}

static inline float snn_o_coshf(float x)
{
    if (!(x == x)) return x;
    double d = (double)x, a = (d < 0.0) ? -d : d;
    if (a > 90.0) return __builtin_inff();
    double e = snn_o_exp_core(a);
    return (float)((e + 1.0 / e) * 0.5);
}

/* sin / cos / tan of generated models (nb_macro lib.rs:9164-9175 forward to the platform libm): Cody-Waite reduction */
/* by pi/2 in binary64 (k * PIO2_HI is exact for |k| < 2^20, i.e. |x| < 1.6e6; beyond that the result stays */
/* deterministic but loses accuracy), Taylor polynomials on [-pi/4, pi/4], one rounding to binary32. */
static inline void snn_o_sincos_core(double x, double *sp, double *cp)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;     /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;     /* pi/2 - pio2_hi */
    const double shift = 6755399441055744.0;               /* 1.5 * 2^52 */
    const double kd = (x * two_over_pi + shift) - shift;
    const double r = (x - kd * pio2_hi) - kd * pio2_lo;
    const double z = r * r;
    double ps = -1.0 / 355687428096000.0;                  /* -1/17! */
    ps = ps * z + 1.0 / 1307674368000.0;                   /* 1/15! */
    ps = ps * z - 1.0 / 6227020800.0;                      /* -1/13! */
    ps = ps * z + 1.0 / 39916800.0;                        /* 1/11! */
    ps = ps * z - 1.0 / 362880.0;                          /* -1/9! */
    ps = ps * z + 1.0 / 5040.0;                            /* 1/7! */
    ps = ps * z - 1.0 / 120.0;                             /* -1/5! */
    ps = ps * z + 1.0 / 6.0;                               /* 1/3!  (sign folded below) */
    const double sr = r - (r * z) * ps;
    double pc = 1.0 / 20922789888000.0;                    /* 1/16! */
    pc = pc * z - 1.0 / 87178291200.0;                     /* -1/14! */
    pc = pc * z + 1.0 / 479001600.0;                       /* 1/12! */
    pc = pc * z - 1.0 / 3628800.0;                         /* -1/10! */
    pc = pc * z + 1.0 / 40320.0;                           /* 1/8! */
    pc = pc * z - 1.0 / 720.0;                             /* -1/6! */
    pc = pc * z + 1.0 / 24.0;                              /* 1/4! */
    pc = pc * z - 0.5;                                     /* -1/2! */
    const double cr = pc * z + 1.0;
    const long long q = (long long)kd & 3ll;
    *sp = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    *cp = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
}
static inline float snn_o_sinf(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;          /* NaN, +-inf -> NaN */
    if (x == 0.0f) return x;                               /* keeps the sign of zero */
    double s, c;
    snn_o_sincos_core((double)x, &s, &c);
    return (float)s;
}
static inline float snn_o_cosf(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;
    double s, c;
    snn_o_sincos_core((double)x, &s, &c);
    return (float)c;
}
static inline float snn_o_tanf(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;
    if (x == 0.0f) return x;                               /* keeps the sign of zero */
    double s, c;
    snn_o_sincos_core((double)x, &s, &c);
    return (float)(s / c);
}

/* x.powf(n), n an integer literal: square-and-multiply in binary64, one rounding */
static inline float snn_o_powif(float x, int n)
{
    double b = (double)x, r = 1.0;
    int m = (n < 0) ? -n : n;
    while (m) {                      /* square and multiply: x^3 = x * x^2, x^4 = (x^2)^2 as snn_o_pow3f / snn_o_pow4f form them */
        if (m & 1) r = r * b;
        m >>= 1;
        if (m) b = b * b;
    }
    return (float)((n < 0) ? 1.0 / r : r);
}

/* x^3 as libm powf(x, 3.) returns it (ion_channels/mod.rs:234) */
static inline float snn_o_pow3f(float x)
{
    double d = (double)x;
    return (float)((d * d) * d);
}

/* x^4 as libm powf(x, 4.) returns it (ion_channels/mod.rs:280) */
static inline float snn_o_pow4f(float x)
{
    double d = (double)x;
    double d2 = d * d;
    return (float)(d2 * d2);
}

#endif /* SNN_ORACLE_MATH_H */

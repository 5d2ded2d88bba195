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

static inline float snn_o_expf(float x)
{
    if (!(x == x)) return x;                 /* NaN in, NaN out */
    if (x > 89.0f) return __builtin_inff();  /* expf overflows above 88.72 */
    if (x < -104.0f) return 0.0f;            /* below half the least subnormal */

    const double INV_LN2 = 1.4426950408889634;       /* 0x3FF71547652B82FE */
    const double LN2_HI  = 6.93147180369123816490e-01; /* 0x3FE62E42FEE00000 */
    const double LN2_LO  = 1.90821492927058770002e-10; /* 0x3DEA39EF35793C76 */
    const double SHIFT   = 6755399441055744.0;        /* 1.5 * 2^52 */

    double xd = (double)x;
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

    int64_t k = (int64_t)kd;                      /* |k| <= 151 */
    uint64_t bits = (uint64_t)(k + 1023) << 52;   /* 2^k as a normal double */
    double scale;
    memcpy(&scale, &bits, sizeof scale);
    return (float)(p * scale);
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

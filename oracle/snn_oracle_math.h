/*
 * oracle/snn_oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Transcendental helpers of the CPU oracle.
 *
 * The reference computes `f32::exp` and `f32::powf` through Rust `std`, i.e.
 * the platform libm: glibc `expf` / `powf` (call sites e.g.
 * backend/src/neuron/ion_channels/mod.rs:224-228,234,270-271,280,
 * iterate_and_spike/mod.rs:149,1133, plasticity/mod.rs:52-54,
 * spike_train/mod.rs:85).  glibc is a third-party dependency that is not under
 * /root/reference (un-vendored; backend/Cargo.lock pins no libm -- the crate
 * takes whatever libm.so.6 the host has).  This header restates glibc's
 * published algorithm (sysdeps/ieee754/flt-32/e_expf.c, e_powf.c,
 * e_exp2f_data.c, e_powf_log2_data.c -- the ARM optimized-routines code glibc
 * has shipped since 2.27) in the form glibc 2.35 selects on every x86-64 CPU
 * with FMA (the `__expf_fma` / `__powf_fma` ifunc variants: the same source
 * built with -mfma -mavx2, so `a * b + c` is ONE rounding wherever the
 * source has that shape).  Which products are fused was read off the
 * disassembly of the container's libm.so.6 (Ubuntu GLIBC 2.35-0ubuntu3.11);
 * every fused operation is an explicit fma() here and the file is built with
 * -ffp-contract=off, so nothing else fuses.
 *
 * Pinned EXHAUSTIVELY: oracle/check_libm.c compares all 2^32 inputs of expf
 * and all 2^32 values of x for powf(x, 3.) and powf(x, 4.) (plus a sampled
 * (x, y) plane) with the container's libm.so.6 -- zero mismatches outside NaN
 * payloads (tests/test_oracle_math.py runs it).
 *
 * The tanh / sinh / cosh / sin / cos / tan helpers further down serve generated
 * models only (out of the hot path) and keep their own binary64 evaluation.
 *
 * Build flags that matter: -ffp-contract=off -fno-fast-math -mfma (oracle/Makefile).
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

/* ---- bit casts ------------------------------------------------------------------------------ */
static inline uint32_t snn_o_asuint(float f) { uint32_t u; memcpy(&u, &f, sizeof u); return u; }
static inline float snn_o_asfloat(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }
static inline uint64_t snn_o_asuint64(double f) { uint64_t u; memcpy(&u, &f, sizeof u); return u; }
static inline double snn_o_asdouble(uint64_t u) { double f; memcpy(&f, &u, sizeof f); return f; }

/* glibc __exp2f_data (e_exp2f_data.c), EXP2F_TABLE_BITS = 5: tab[i] = bits(2^(i/32)) - (i << 47) */
static const uint64_t SNN_O_EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
#define SNN_O_EXP2F_C0 0x1.c6af84b912394p-5   /* __exp2f_data.poly[] */
#define SNN_O_EXP2F_C1 0x1.ebfce50fac4f3p-3
#define SNN_O_EXP2F_C2 0x1.62e42ff0c52d6p-1

/* glibc expf (e_expf.c, non-TOINT_INTRINSICS branch), FMA build:
 *   z = InvLn2N * x is never rounded on its own: kd = fma(InvLn2N, x, Shift), r = fma(InvLn2N, x, -(kd - Shift));
 *   the polynomial's three `a * b + c` are fused, `r * r` and `y * s` are plain products. */
static inline float snn_o_expf(float x)
{
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;     /* invln2_scaled */
    const double Shift = 0x1.8p+52;
    const uint32_t ux = snn_o_asuint(x);
    const uint32_t abstop = (ux >> 20) & 0x7ff;
    const double xd = (double)x;
    if (abstop >= 0x42b) {                                   /* |x| >= 88 or NaN */
        if (ux == 0xff800000u) return 0.0f;                  /* -inf */
        if (abstop >= 0x7f8) return x + x;                   /* NaN, +inf */
        if (x > 0x1.62e42ep6f) return __builtin_inff();      /* x > log(2^128): overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;                 /* x < log(2^-150): underflow */
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;            /* x < log(2^-149): __math_may_uflowf = 0x1.4p-75f squared */
    }
    double kd = __builtin_fma(InvLn2N, xd, Shift);
    const uint64_t ki = snn_o_asuint64(kd);
    kd -= Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    const double s = snn_o_asdouble(SNN_O_EXP2F_TAB[ki & 31] + (ki << 47));
    const double z = __builtin_fma(r, SNN_O_EXP2F_C0 / 32.0 / 32.0 / 32.0, SNN_O_EXP2F_C1 / 32.0 / 32.0);   /* poly_scaled[] */
    const double r2 = r * r;
    double y = __builtin_fma(r, SNN_O_EXP2F_C2 / 32.0, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

/* glibc __powf_log2_data (e_powf_log2_data.c), POWF_LOG2_TABLE_BITS = 4, POWF_SCALE = 1: {1/c, log2(c)} */
static const double SNN_O_POWF_LOG2_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};

/* checkint of e_powf.c: 0 = not an integer, 1 = odd, 2 = even */
static inline int snn_o_powf_checkint(uint32_t iy)
{
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
static inline int snn_o_powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

/* glibc powf (e_powf.c: log2_inline, exp2_inline, __powf), FMA build.  Fused: r = fma(z, invc, -1), the five
 * polynomial steps of log2_inline and the three of exp2_inline; plain: y0 = logc + k, r * r, r2 * r2, y * logx,
 * xd + Shift, xd - kd, y * s. */
static inline float snn_o_powf(float x, float y)
{
    uint64_t sign_bias = 0;
    uint32_t ix = snn_o_asuint(x);
    const uint32_t iy = snn_o_asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || snn_o_powf_zeroinfnan(iy)) {
        /* either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan) */
        if (snn_o_powf_zeroinfnan(iy)) {
            if (2 * iy == 0) return ((ix & 0x7fc00000u) == 0x7f800000u && (ix & 0x003fffffu)) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return ((iy & 0x7fc00000u) == 0x7f800000u && (iy & 0x003fffffu)) ? x + y : 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   /* |x|<1 && y==inf or |x|>1 && y==-inf */
            return y * y;
        }
        if (snn_o_powf_zeroinfnan(ix)) {
            float x2 = x * x;
            int neg = 0;
            if ((ix & 0x80000000u) && snn_o_powf_checkint(iy) == 1) { x2 = -x2; neg = 1; }
            if (2 * ix == 0 && (iy & 0x80000000u)) return neg ? -__builtin_inff() : __builtin_inff();
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        /* x and y are non-zero finite */
        if (ix & 0x80000000u) {
            const int yint = snn_o_powf_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);      /* finite x < 0, non-integer y: NaN */
            if (yint == 1) sign_bias = 1ull << 16;        /* SIGN_BIAS */
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                           /* subnormal x: normalise */
            ix = snn_o_asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    /* log2_inline: x = 2^k z, z in [OFF, 2 OFF); log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k */
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = SNN_O_POWF_LOG2_TAB[i][0], logc = SNN_O_POWF_LOG2_TAB[i][1];
    const double z = (double)snn_o_asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double r2 = r * r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * yy;                  /* cannot overflow, y is single precision */
    if ((snn_o_asuint64(ylogx) >> 47 & 0xffff) >= (snn_o_asuint64(126.0) >> 47)) {
        /* |y * log2(x)| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
        if (ylogx < -149.0) return sign_bias ? -0x1p-149f : 0x1p-149f;   /* __math_may_uflowf */
    }
    /* exp2_inline: 2^ylogx = 2^(k/32) * 2^r */
    const double ShiftScaled = 0x1.8p+52 / 32.0;
    double kd = ylogx + ShiftScaled;
    const uint64_t ki = snn_o_asuint64(kd);
    kd -= ShiftScaled;
    const double rr = ylogx - kd;
    const double s = snn_o_asdouble(SNN_O_EXP2F_TAB[ki & 31] + ((ki + sign_bias) << 47));
    const double zz = __builtin_fma(SNN_O_EXP2F_C0, rr, SNN_O_EXP2F_C1);
    const double rr2 = rr * rr;
    double out = __builtin_fma(SNN_O_EXP2F_C2, rr, 1.0);
    out = __builtin_fma(zz, rr2, out);
    out = out * s;
    return (float)out;
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

/* x.powf(n) for an integer literal n, as rustc -O compiles it: LLVM's libcall simplifier folds powf(x, 2.) to x * x,
 * powf(x, 1.) to x, powf(x, 0.) to 1 and powf(x, -1.) to 1 / x without any fast-math flag; every other exponent stays a
 * call of libm powf. */
static inline float snn_o_powif(float x, int n)
{
    if (n == 2) return x * x;
    if (n == 1) return x;
    if (n == 0) return 1.0f;
    if (n == -1) return 1.0f / x;
    return snn_o_powf(x, (float)n);
}

/* powf(x, 3.) (ion_channels/mod.rs:234) and powf(x, 4.) (ion_channels/mod.rs:280): genuine libm calls */
static inline float snn_o_pow3f(float x) { return snn_o_powf(x, 3.0f); }
static inline float snn_o_pow4f(float x) { return snn_o_powf(x, 4.0f); }

#endif /* SNN_ORACLE_MATH_H */

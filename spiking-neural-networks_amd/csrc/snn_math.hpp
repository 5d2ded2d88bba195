// Device-side scalar math of the lattice stepper (gfx950).
//
// The reference evaluates f32::exp / f32::powf through the platform libm, i.e. glibc expf / powf
// (backend/src/neuron/ion_channels/mod.rs:224-228,234,270-271,280;
//  iterate_and_spike/mod.rs:149,1133; plasticity/mod.rs:52-54;
//  spike_train/mod.rs:85).  ocml's expf/powf are not bit-compatible with it, so the stepper carries glibc's
// published algorithm (sysdeps/ieee754/flt-32/e_expf.c, e_powf.c and their two tables) in the form glibc >= 2.27
// runs on every x86-64 CPU with FMA: binary64 throughout, a 32-entry 2^(i/32) table, a 16-entry log2 table, and
// v_fma_f64 exactly where the FMA build of glibc fuses (written as explicit fma(); the translation unit is compiled
// with -ffp-contract=off, so nothing else fuses), one rounding to binary32 at the end.  The test suite's CPU twin of
// these functions is pinned against libm.so.6 on all 2^32 inputs and tests/test_gpu_math.py holds the device functions
// to that twin on 2^28+ inputs.  MI355X issues FP64 vector ops at half the FP32 rate and the stepper is
// HBM-bound, so the f64 evaluation is free in practice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snn {

// exp of a binary64 argument, |x| < 700: the polynomial core of the hyperbolic functions of generated models
__device__ __forceinline__ double exp_core(double xd)
{
    const double inv_ln2 = 1.4426950408889634;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double shift = 6755399441055744.0;   // 1.5 * 2^52

    const double kd = (xd * inv_ln2 + shift) - shift;
    const double r = (xd - kd * ln2_hi) - kd * ln2_lo;

    double p = 1.6059043836821613e-10;
    p = p * r + 2.08767569878681e-09;
    p = p * r + 2.505210838544172e-08;
    p = p * r + 2.755731922398589e-07;
    p = p * r + 2.7557319223985893e-06;
    p = p * r + 2.48015873015873e-05;
    p = p * r + 1.984126984126984e-04;
    p = p * r + 1.388888888888889e-03;
    p = p * r + 8.333333333333333e-03;
    p = p * r + 4.1666666666666664e-02;
    p = p * r + 1.6666666666666666e-01;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;

    const int k = (int)kd;   // |k| <= 1010
    const double scale = __longlong_as_double((long long)(k + 1023) << 52);
    return p * scale;
}

// glibc __exp2f_data.tab (EXP2F_TABLE_BITS = 5): bits(2^(i/32)) - (i << 47)
static __device__ const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
// glibc __powf_log2_data.tab (POWF_LOG2_TABLE_BITS = 4): {1/c, log2(c)}
static __device__ const double POWF_LOG2_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
constexpr double EXP2F_C0 = 0x1.c6af84b912394p-5, EXP2F_C1 = 0x1.ebfce50fac4f3p-3, EXP2F_C2 = 0x1.62e42ff0c52d6p-1;

// The main path of expf_glibc (|x| < 88, not NaN) on its own, branch-free: several independent calls in one basic block have their
// table loads in flight together, where the full function's early exits put each call -- and its dependent table load -- into a
// block of its own (round 6: the seven exponentials of a Hodgkin-Huxley step were seven round trips to the table).  Outside the
// main path the value returned is meaningless and `special` is set: the caller then evaluates expf_glibc.
template <bool LOCAL_CONSTANTS = false>
__device__ __forceinline__ float expf_glibc_main(float x, bool &special)
{
    constexpr double inv_ln2_n = 0x1.71547652b82fep+0 * 32.0;
    constexpr double shift = 0x1.8p+52;
    special = special || ((__float_as_uint(x) >> 20) & 0x7ff) >= 0x42b;
    const double xd = (double)x;
    double kd = __builtin_fma(inv_ln2_n, xd, shift);
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= shift;
    const double r = __builtin_fma(inv_ln2_n, xd, -kd);
    const double s = __longlong_as_double((long long)(EXP2F_TAB[ki & 31] + (ki << 47)));
    double c1 = EXP2F_C1 / 32.0 / 32.0;
    if (LOCAL_CONSTANTS) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(c1);
        uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));       // two scalar literals here, moved into the register pair at the use
        c1 = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    const double z = __builtin_fma(r, EXP2F_C0 / 32.0 / 32.0 / 32.0, c1);
    const double r2 = r * r;
    double y = __builtin_fma(r, EXP2F_C2 / 32.0, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

// expf as glibc computes it (e_expf.c, FMA build): k + r = x * 32/ln2 without rounding the product on its own,
// 2^(k/32) from the table, a cubic in r.
// LOCAL_CONSTANTS: the cubic's constant term is materialised where it is used (it is the accumulator of an FMA and has to sit
// in a register pair; inside a long-lived loop the compiler otherwise keeps that pair alive across the loop -- and spilled it
// in k_run_resident, whose registers are the weights').  Same arithmetic.
template <bool LOCAL_CONSTANTS = false>
__device__ __forceinline__ float expf_glibc(float x)
{
    const uint32_t ux = __float_as_uint(x);
    const uint32_t abstop = (ux >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                                   // |x| >= 88 or NaN
        if (ux == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();      // x > log(2^128)
        if (x < -0x1.9fe368p6f) return 0.0f;                 // x < log(2^-150)
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;            // x < log(2^-149)
    }
    bool unused = false;
    return expf_glibc_main<LOCAL_CONSTANTS>(x, unused);
}

// checkint of e_powf.c: 0 = not an integer, 1 = odd, 2 = even
__device__ __forceinline__ int powf_checkint(uint32_t iy)
{
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
__device__ __forceinline__ bool powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

// log2(x) * y and 2^that of powf_glibc's main path (x a positive normal number after the caller's adjustments): the caller has
// dealt with the special operands; `big` = |y log2 x| >= 126 (overflow / underflow handling of the full function)
__device__ __forceinline__ float powf_glibc_core(uint32_t ix, float y, uint64_t sign_bias, bool &big, double &ylogx_out)
{
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = POWF_LOG2_TAB[i][0], logc = POWF_LOG2_TAB[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    constexpr double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                     A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double r2 = r * r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    ylogx_out = ylogx;
    big = ((uint64_t)__double_as_longlong(ylogx) >> 47 & 0xffff) >= (0x405f800000000000ull >> 47);   // |y log2 x| >= 126
    constexpr double shift_scaled = 0x1.8p+52 / 32.0;
    double kd = ylogx + shift_scaled;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= shift_scaled;
    const double rr = ylogx - kd;
    const double s = __longlong_as_double((long long)(EXP2F_TAB[ki & 31] + ((ki + sign_bias) << 47)));
    const double zz = __builtin_fma(EXP2F_C0, rr, EXP2F_C1);
    const double rr2 = rr * rr;
    double out = __builtin_fma(EXP2F_C2, rr, 1.0);
    out = __builtin_fma(zz, rr2, out);
    out = out * s;
    return (float)out;
}

// the main path of powf_glibc on its own, branch-free (see expf_glibc_main): x a positive normal number, y finite and non-zero,
// |y log2 x| < 126; otherwise `special` is set and the value meaningless
__device__ __forceinline__ float powf_glibc_main(float x, float y, bool &special)
{
    const uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    bool big;
    double ylogx;
    const float out = powf_glibc_core(ix, y, 0, big, ylogx);
    special = special || big || ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy);
    return out;
}

// powf as glibc computes it (e_powf.c, FMA build): log2(x) from a 16-entry table + quartic, times y, then exp2.
// With a literal y (3.f, 4.f) the compiler folds every test on y.
__device__ __forceinline__ float powf_glibc(float x, float y)
{
    uint64_t sign_bias = 0;
    uint32_t ix = __float_as_uint(x);
    const uint32_t iy = __float_as_uint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
        if (powf_zeroinfnan(iy)) {
            if (2 * iy == 0) return ((ix & 0x7fc00000u) == 0x7f800000u && (ix & 0x003fffffu)) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return ((iy & 0x7fc00000u) == 0x7f800000u && (iy & 0x003fffffu)) ? x + y : 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (powf_zeroinfnan(ix)) {
            float x2 = x * x;
            bool neg = false;
            if ((ix & 0x80000000u) && powf_checkint(iy) == 1) { x2 = -x2; neg = true; }
            if (2 * ix == 0 && (iy & 0x80000000u)) return neg ? -__builtin_inff() : __builtin_inff();
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000u) {                              // finite x < 0
            const int yint = powf_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 1ull << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                              // subnormal x: normalise
            ix = __float_as_uint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    bool big;
    double ylogx;
    const float out = powf_glibc_core(ix, y, sign_bias, big, ylogx);
    if (big) {
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
        if (ylogx < -149.0) return sign_bias ? -0x1p-149f : 0x1p-149f;
    }
    return out;
}

// f32::tanh / sinh / cosh of generated models (build_test/nb_macro/src/lib.rs:9152-9163 forward to the platform
// libm): binary64 through exp_core, a Taylor polynomial where e^x - e^-x would cancel, one rounding to binary32.
__device__ __forceinline__ float tanhf_portable(float x)
{
    if (!(x == x)) return x;
    const double d = (double)x, a = (d < 0.0) ? -d : d;
    double t;
    if (a < 0.05) {
        const double z = a * a;
        double p = 62.0 / 2835.0;
        p = p * z - 17.0 / 315.0;
        p = p * z + 2.0 / 15.0;
        p = p * z - 1.0 / 3.0;
        p = p * z + 1.0;
        t = a * p;
    } else if (a > 20.0) {
        t = 1.0;
    } else {
        t = 1.0 - 2.0 / (exp_core(2.0 * a) + 1.0);
    }
    return (float)((d < 0.0) ? -t : t);
}
__device__ __forceinline__ float sinhf_portable(float x)
{
    if (!(x == x)) return x;
    const double d = (double)x, a = (d < 0.0) ? -d : d;
    double t;
    if (a < 0.05) {
        const double z = a * a;
        double p = 1.0 / 5040.0;
        p = p * z + 1.0 / 120.0;
        p = p * z + 1.0 / 6.0;
        p = p * z + 1.0;
        t = a * p;
    } else if (a > 90.0) {
        t = (double)__builtin_inff();
    } else {
        const double e = exp_core(a);
        t = (e - 1.0 / e) * 0.5;
    }
    return (float)((d < 0.0) ? -t : t);
}
__device__ __forceinline__ float coshf_portable(float x)
{
    if (!(x == x)) return x;
    const double d = (double)x, a = (d < 0.0) ? -d : d;
    if (a > 90.0) return __builtin_inff();
    const double e = exp_core(a);
    return (float)((e + 1.0 / e) * 0.5);
}

// sin / cos / tan of generated models (nb_macro lib.rs:9164-9175 forward to the platform libm): Cody-Waite reduction
// by pi/2 in binary64 (k * PIO2_HI is exact for |k| < 2^20, i.e. |x| < 1.6e6; beyond that the result stays
// deterministic but loses accuracy), Taylor polynomials on [-pi/4, pi/4], one rounding to binary32.
__device__ __forceinline__ void sincos_core(double x, double &s, double &c)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;     // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;     // pi/2 - pio2_hi
    const double shift = 6755399441055744.0;               // 1.5 * 2^52
    const double kd = (x * two_over_pi + shift) - shift;
    const double r = (x - kd * pio2_hi) - kd * pio2_lo;
    const double z = r * r;
    double ps = -1.0 / 355687428096000.0;                  // -1/17!
    ps = ps * z + 1.0 / 1307674368000.0;                   // 1/15!
    ps = ps * z - 1.0 / 6227020800.0;                      // -1/13!
    ps = ps * z + 1.0 / 39916800.0;                        // 1/11!
    ps = ps * z - 1.0 / 362880.0;                          // -1/9!
    ps = ps * z + 1.0 / 5040.0;                            // 1/7!
    ps = ps * z - 1.0 / 120.0;                             // -1/5!
    ps = ps * z + 1.0 / 6.0;                               // 1/3!  (sign folded below)
    const double sr = r - (r * z) * ps;
    double pc = 1.0 / 20922789888000.0;                    // 1/16!
    pc = pc * z - 1.0 / 87178291200.0;                     // -1/14!
    pc = pc * z + 1.0 / 479001600.0;                       // 1/12!
    pc = pc * z - 1.0 / 3628800.0;                         // -1/10!
    pc = pc * z + 1.0 / 40320.0;                           // 1/8!
    pc = pc * z - 1.0 / 720.0;                             // -1/6!
    pc = pc * z + 1.0 / 24.0;                              // 1/4!
    pc = pc * z - 0.5;                                     // -1/2!
    const double cr = pc * z + 1.0;
    const long long q = (long long)kd & 3ll;
    s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
}
__device__ __forceinline__ float sinf_portable(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;          // NaN, +-inf -> NaN
    if (x == 0.0f) return x;                               // keeps the sign of zero
    double s, c;
    sincos_core((double)x, s, c);
    return (float)s;
}
__device__ __forceinline__ float cosf_portable(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;
    double s, c;
    sincos_core((double)x, s, c);
    return (float)c;
}
__device__ __forceinline__ float tanf_portable(float x)
{
    if (!(x == x) || x - x != 0.0f) return x - x;
    if (x == 0.0f) return x;                               // keeps the sign of zero
    double s, c;
    sincos_core((double)x, s, c);
    return (float)(s / c);
}

// x.powf(n) for an integer literal n (the only exponents the generator accepts) as rustc -O compiles it: LLVM folds
// powf(x, 2.) to x * x, powf(x, 1.) to x, powf(x, 0.) to 1 and powf(x, -1.) to 1 / x without fast-math flags; every other
// exponent stays a libm call.
__device__ __forceinline__ float powif_glibc(float x, int n)
{
    if (n == 2) return x * x;
    if (n == 1) return x;
    if (n == 0) return 1.0f;
    if (n == -1) return 1.0f / x;
    return powf_glibc(x, (float)n);
}

// nb_macro's heaviside (lib.rs:9176-9178): `if x < 0 { 0 } else { x }`
__device__ __forceinline__ float heaviside_rs(float x) { return (x < 0.0f) ? 0.0f : x; }

// powf(x, 3.) / powf(x, 4.) of the Na / K channel currents (ion_channels/mod.rs:234, 280): genuine libm calls
__device__ __forceinline__ float pow3f_glibc(float x) { return powf_glibc(x, 3.0f); }
__device__ __forceinline__ float pow4f_glibc(float x) { return powf_glibc(x, 4.0f); }

// f32::max / f32::min as Rust defines them (a NaN operand yields the other one)
__device__ __forceinline__ float max_rs(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return (a > b) ? a : b;
}
__device__ __forceinline__ float min_rs(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return (a < b) ? a : b;
}

// nb_macro's `a r^ b` (lib.rs:136): (a.max(0.0f32).powf(b))
__device__ __forceinline__ float rpowf_glibc(float a, float b) { return powf_glibc(max_rs(a, 0.0f), b); }

// xorshift32 of the reference's Poisson kernel (spike_train/mod.rs:380-388)
__device__ __forceinline__ uint32_t xorshift32(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

// DeltaDiracRefractoriness::get_effect (spike_train/mod.rs:67-88)
template <bool LOCAL_CONSTANTS = false>
__device__ __forceinline__ float delta_dirac_effect(long long timestep, int last_firing_time,
                                                    float v_th, float v_resting, float k, float dt)
{
    const float a = v_th - v_resting;
    const float td = (float)(timestep - (long long)last_firing_time);
    return a * expf_glibc<LOCAL_CONSTANTS>((-1.0f / (k / dt)) * (td * td)) + v_resting;
}

// ExponentialDecayRefractoriness::get_effect (spike_train/mod.rs:164-178)
template <bool LOCAL_CONSTANTS = false>
__device__ __forceinline__ float exponential_decay_effect(long long timestep, int last_firing_time,
                                                          float v_th, float v_resting, float k, float dt)
{
    const float a = v_th - v_resting;
    const float td = (float)(timestep - (long long)last_firing_time);
    return a * expf_glibc<LOCAL_CONSTANTS>((-1.0f / (k / dt)) * td) + v_resting;
}

// STDP::update_weight (plasticity/mod.rs:45-66): the delta added to the weight
__device__ __forceinline__ float stdp_delta(int t_pre, int t_post, float a_plus, float a_minus,
                                            float tau_plus, float tau_minus, float dt)
{
    if (t_pre < 0 || t_post < 0) return 0.0f;
    const float tp = (float)t_pre, tq = (float)t_post;
    if (tp < tq) return a_plus * expf_glibc(-1.0f * __builtin_fabsf((tp - tq) * dt) / tau_plus);
    if (tp > tq) return -1.0f * a_minus * expf_glibc(-1.0f * __builtin_fabsf((tq - tp) * dt) / tau_minus);
    return 0.0f;
}

// Firing-rate bookkeeping shared by BCMIzhikevichNeuron (integrate_and_fire/mod.rs:1458-1469, 1484-1495) and
// BCMPoissonNeuron (spike_train/mod.rs:943-954); num_spikes is never reset by the reference.  per_dt: the electrical-only
// neuron path and the spike train divide by (window * dt), the neuron's neurotransmission path by the window alone.
__device__ __forceinline__ void bcm_window_update(float &clock, float window, float dt, uint32_t num_spikes,
                                                  uint32_t period, float &current_activity, float &average_activity,
                                                  bool per_dt)
{
    clock += dt;
    if (clock >= window) {
        clock = 0.0f;
        current_activity = per_dt ? (float)num_spikes / (window * dt) : (float)num_spikes / window;
        average_activity -= average_activity / (float)period;
        average_activity += current_activity / (float)period;
    }
}

// One weight update of a lattice's plasticity rule.  Table row of PL_STRIDE floats per lattice:
// {a_plus, a_minus, tau_plus, tau_minus, dt, kind (0 STDP, 1 BCM), bcm decay, bcm average_scalar}.
// STDP::update_weight plasticity/mod.rs:45-66; BCM::update_weight :102-107 (dt shared with slot 4).
constexpr int PL_STRIDE = 8;
__device__ __forceinline__ float plasticity_weight(const float *prm, float w, int t_pre, int t_post,
                                                   float pre_activity, float post_activity, float post_average)
{
    if (prm[5] != 0.0f) {
        const float sliding_threshold = post_average / prm[7];
        const float activity_term = post_activity * (post_activity - sliding_threshold);
        const float weight_decay = prm[6] * w;
        return w + (activity_term * pre_activity - weight_decay) * prm[4];
    }
    return w + stdp_delta(t_pre, t_post, prm[0], prm[1], prm[2], prm[3], prm[4]);
}

// ---- neurotransmitter / receptor kinetics ------------------------------------------------------
// exp_decay, iterate_and_spike/mod.rs:345-347
__device__ __forceinline__ float exp_decay(float x, float l, float dt)
{
    return -x * expf_glibc(dt / -l);
}

// NeurotransmitterKinetics::apply_t_change.  kind 0 Approximate (iterate_and_spike/mod.rs:193-196), 1 Destexhe
// (:148-150), 2 DiscreteSpike (:300-302), 3 ExponentialDecay (:350-354; `c` = decay_constant instead of the
// clearance constant).
__device__ __forceinline__ float nt_apply(int kind, float t, float t_max, float c, float v_p, float k_p,
                                          float voltage, uint32_t spiking, float dt)
{
    const float s = spiking ? 1.0f : 0.0f;
    if (kind == 1) return t_max / (1.0f + expf_glibc(-(voltage - v_p) / k_p));
    if (kind == 2) return t_max * s;
    if (kind == 3) t += exp_decay(t, c, dt) + (s * t_max);
    else t += dt * -c * t + (s * t_max);
    return min_rs(t_max, max_rs(t, 0.0f));
}

// ReceptorKinetics::apply_r_change.  kind 0 Approximate (iterate_and_spike/mod.rs:435-437), 1 Destexhe (:404-406),
// 2 ExponentialDecay (:510-513; alpha = r_max, beta = decay_constant).
__device__ __forceinline__ float rc_apply(int kind, float r, float t, float alpha, float beta, float dt)
{
    if (kind == 1) return r + (alpha * t * (1.0f - r) - beta * r) * dt;
    if (kind == 2) {
        r += exp_decay(r, beta, dt) + t;
        return min_rs(alpha, max_rs(r, 0.0f));
    }
    return t;
}

// Counter-based synthetic data (splitmix64 finaliser), used by the device-side
// graph / state generators so that benchmark-size inputs never cross PCIe.
__host__ __device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t index)
{
    uint64_t x = index + seed * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
__host__ __device__ __forceinline__ float uniform_from_hash(uint64_t seed, uint64_t index, float lo, float hi)
{
    const float u = (float)(hash32(seed, index) >> 8) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * u;
}

} // namespace snn

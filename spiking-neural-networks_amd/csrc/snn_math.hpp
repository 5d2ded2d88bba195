// Device-side scalar math of the lattice stepper (gfx950).
//
// The reference evaluates f32::exp / f32::powf through the platform libm
// (backend/src/neuron/ion_channels/mod.rs:224-228,234,270-271,280;
//  iterate_and_spike/mod.rs:149,1133; plasticity/mod.rs:52-54;
//  spike_train/mod.rs:85).  ocml's expf/powf are not bit-compatible with any CPU
// libm, so the stepper carries its own exp: range reduction by ln2, a degree-13
// Taylor polynomial by Horner's rule in binary64 with plain v_mul_f64/v_add_f64
// (this translation unit is compiled with -ffp-contract=off: no FMA anywhere),
// one rounding to binary32.  MI355X issues FP64 vector ops at half the FP32 rate,
// and the stepper is HBM-bound, so the f64 polynomial is free in practice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snn {

// exp of a binary64 argument, |x| < 700: the polynomial core shared by expf_portable and the hyperbolic functions
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

__device__ __forceinline__ float expf_portable(float x)
{
    if (!(x == x)) return x;
    if (x > 89.0f) return __builtin_inff();
    if (x < -104.0f) return 0.0f;
    return (float)exp_core((double)x);
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

// x.powf(n) for an integer literal n (the only exponents the generator accepts): square-and-multiply in binary64,
// one rounding to binary32; n < 0 -> reciprocal of the product
__device__ __forceinline__ float powif_portable(float x, int n)
{
    double b = (double)x, r = 1.0;
    int m = (n < 0) ? -n : n;
    while (m) {                      // square and multiply: x^3 = x * x^2, x^4 = (x^2)^2 as pow3f / pow4f form them
        if (m & 1) r = r * b;
        m >>= 1;
        if (m) b = b * b;
    }
    return (float)((n < 0) ? 1.0 / r : r);
}

// nb_macro's heaviside (lib.rs:9176-9178): `if x < 0 { 0 } else { x }`
__device__ __forceinline__ float heaviside_rs(float x) { return (x < 0.0f) ? 0.0f : x; }

// powf(x, 3.) / powf(x, 4.) of the Na / K channel currents (ion_channels/mod.rs:234, 280):
// exact square in binary64, one rounding each.
__device__ __forceinline__ float pow3f_portable(float x)
{
    const double d = (double)x;
    return (float)((d * d) * d);
}
__device__ __forceinline__ float pow4f_portable(float x)
{
    const double d = (double)x;
    const double d2 = d * d;
    return (float)(d2 * d2);
}

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

// xorshift32 of the reference's Poisson kernel (spike_train/mod.rs:380-388)
__device__ __forceinline__ uint32_t xorshift32(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

// DeltaDiracRefractoriness::get_effect (spike_train/mod.rs:67-88)
__device__ __forceinline__ float delta_dirac_effect(long long timestep, int last_firing_time,
                                                    float v_th, float v_resting, float k, float dt)
{
    const float a = v_th - v_resting;
    const float td = (float)(timestep - (long long)last_firing_time);
    return a * expf_portable((-1.0f / (k / dt)) * (td * td)) + v_resting;
}

// ExponentialDecayRefractoriness::get_effect (spike_train/mod.rs:164-178)
__device__ __forceinline__ float exponential_decay_effect(long long timestep, int last_firing_time,
                                                          float v_th, float v_resting, float k, float dt)
{
    const float a = v_th - v_resting;
    const float td = (float)(timestep - (long long)last_firing_time);
    return a * expf_portable((-1.0f / (k / dt)) * td) + v_resting;
}

// STDP::update_weight (plasticity/mod.rs:45-66): the delta added to the weight
__device__ __forceinline__ float stdp_delta(int t_pre, int t_post, float a_plus, float a_minus,
                                            float tau_plus, float tau_minus, float dt)
{
    if (t_pre < 0 || t_post < 0) return 0.0f;
    const float tp = (float)t_pre, tq = (float)t_post;
    if (tp < tq) return a_plus * expf_portable(-1.0f * __builtin_fabsf((tp - tq) * dt) / tau_plus);
    if (tp > tq) return -1.0f * a_minus * expf_portable(-1.0f * __builtin_fabsf((tq - tp) * dt) / tau_minus);
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
    return -x * expf_portable(dt / -l);
}

// NeurotransmitterKinetics::apply_t_change.  kind 0 Approximate (iterate_and_spike/mod.rs:193-196), 1 Destexhe
// (:148-150), 2 DiscreteSpike (:300-302), 3 ExponentialDecay (:350-354; `c` = decay_constant instead of the
// clearance constant).
__device__ __forceinline__ float nt_apply(int kind, float t, float t_max, float c, float v_p, float k_p,
                                          float voltage, uint32_t spiking, float dt)
{
    const float s = spiking ? 1.0f : 0.0f;
    if (kind == 1) return t_max / (1.0f + expf_portable(-(voltage - v_p) / k_p));
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

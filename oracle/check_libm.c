/*
 * oracle/check_libm.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Exhaustive pin of the oracle's expf / powf restatement (snn_oracle_math.h) against the libm the reference's
 * `f32::exp` / `f32::powf` resolve to (glibc's libm.so.6, default symbol versions expf@@GLIBC_2.27 /
 * powf@@GLIBC_2.27 -- the ones a Rust binary built on this image links):
 *
 *   expf   all 2^32 bit patterns of x
 *   pow3   all 2^32 bit patterns of x, y = 3.0f      (ion_channels/mod.rs:234)
 *   pow4   all 2^32 bit patterns of x, y = 4.0f      (ion_channels/mod.rs:280)
 *   powf   `samples` pseudo-random (x, y) bit patterns + a grid of special values
 *
 * Usage: check_libm <expf|pow3|pow4|powf|all> [stride] [samples]
 *   stride s > 1 visits every s-th bit pattern (quick mode).  Prints one line per function,
 *   `name checked=<n> mismatches=<m> nan_payload_only=<p>`, and up to 10 offending inputs; exit status 1 on a mismatch.
 * A NaN-vs-NaN pair with different payload/sign is counted apart (IEEE 754 leaves it open).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "snn_oracle_math.h"

typedef float (*f1_t)(float);
typedef float (*f2_t)(float, float);
/* volatile function pointers: the compiler can neither fold the libm calls nor substitute a builtin */
static volatile f1_t libm_expf = expf;
static volatile f2_t libm_powf = powf;

static int same(float a, float b, uint64_t *payload_only)
{
    const uint32_t ua = snn_o_asuint(a), ub = snn_o_asuint(b);
    if (ua == ub) return 1;
    if (a != a && b != b) { (*payload_only)++; return 1; }
    return 0;
}

static int check_unary(const char *name, int which, uint64_t stride)
{
    uint64_t bad = 0, payload = 0, checked = 0;
    const f1_t ref_exp = libm_expf;
    const f2_t ref_pow = libm_powf;
#pragma omp parallel for schedule(static) reduction(+ : bad, payload, checked)
    for (uint64_t blk = 0; blk < 4096; blk++) {
        for (uint64_t u = blk << 20; u < (blk + 1) << 20; u += stride) {
            const float x = snn_o_asfloat((uint32_t)u);
            float want, got;
            if (which == 0) { want = ref_exp(x); got = snn_o_expf(x); }
            else if (which == 1) { want = ref_pow(x, 3.0f); got = snn_o_pow3f(x); }
            else { want = ref_pow(x, 4.0f); got = snn_o_pow4f(x); }
            checked++;
            if (!same(want, got, &payload)) {
#pragma omp critical
                if (bad < 10)
                    fprintf(stderr, "%s(%a = 0x%08x): libm %a (0x%08x), oracle %a (0x%08x)\n", name, x, (uint32_t)u, want,
                            snn_o_asuint(want), got, snn_o_asuint(got));
                bad++;
            }
        }
    }
    printf("%s checked=%llu mismatches=%llu nan_payload_only=%llu\n", name, (unsigned long long)checked,
           (unsigned long long)bad, (unsigned long long)payload);
    return bad != 0;
}

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int check_powf_plane(uint64_t samples)
{
    static const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.0f, -2.0f, 0.5f, -0.5f, 3.0f, -3.0f, 4.0f, 1e-40f, -1e-40f,
                                    1e-45f, 0x1p-126f, 0x1.fffffep127f, -0x1.fffffep127f, 1.5f, -1.5f, 0x1p24f, 0x1p25f,
                                    -0x1p24f, 16777215.0f, -16777215.0f, 127.0f, 128.0f, -149.0f, -150.0f, 0.99999994f,
                                    1.0000001f, INFINITY, -INFINITY, NAN, -NAN};
    const int ns = (int)(sizeof special / sizeof special[0]);
    uint64_t bad = 0, payload = 0, checked = 0;
    const f2_t ref_pow = libm_powf;
    for (int a = 0; a < ns; a++)
        for (int b = 0; b < ns; b++) {
            const float want = ref_pow(special[a], special[b]), got = snn_o_powf(special[a], special[b]);
            checked++;
            if (!same(want, got, &payload)) {
                if (bad < 10) fprintf(stderr, "powf(%a, %a): libm %a, oracle %a\n", special[a], special[b], want, got);
                bad++;
            }
        }
#pragma omp parallel for schedule(static) reduction(+ : bad, payload, checked)
    for (uint64_t blk = 0; blk < 1024; blk++) {
        uint64_t seed = 0x5EEDull + blk;
        for (uint64_t n = 0; n < samples / 1024; n++) {
            const uint64_t bits = splitmix(&seed);
            float x = snn_o_asfloat((uint32_t)bits), y = snn_o_asfloat((uint32_t)(bits >> 32));
            if (n & 1) {   /* half of the samples where the result is finite and non-trivial */
                x = snn_o_asfloat(0x3f800000u + (int32_t)((bits & 0x7ffffff) - 0x4000000));      /* x in 2^-8 .. 2^8 */
                y = snn_o_asfloat(((uint32_t)(bits >> 32) & 0x83ffffffu) | 0x3c000000u);          /* |y| in 2^-7 .. 2^9 */
                if (n & 2) x = -x, y = (float)(int)y;
            }
            const float want = ref_pow(x, y), got = snn_o_powf(x, y);
            checked++;
            if (!same(want, got, &payload)) {
#pragma omp critical
                if (bad < 10) fprintf(stderr, "powf(%a, %a): libm %a, oracle %a\n", x, y, want, got);
                bad++;
            }
        }
    }
    printf("powf checked=%llu mismatches=%llu nan_payload_only=%llu\n", (unsigned long long)checked,
           (unsigned long long)bad, (unsigned long long)payload);
    return bad != 0;
}

int main(int argc, char **argv)
{
    const char *what = argc > 1 ? argv[1] : "all";
    const uint64_t stride = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
    const uint64_t samples = argc > 3 ? strtoull(argv[3], NULL, 10) : (1ull << 28);
    const int all = !strcmp(what, "all");
    int rc = 0;
    if (stride == 0) return 2;
    if (all || !strcmp(what, "expf")) rc |= check_unary("expf", 0, stride);
    if (all || !strcmp(what, "pow3")) rc |= check_unary("pow3", 1, stride);
    if (all || !strcmp(what, "pow4")) rc |= check_unary("pow4", 2, stride);
    if (all || !strcmp(what, "powf")) rc |= check_powf_plane(samples);
    return rc;
}

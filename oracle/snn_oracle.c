/*
 * oracle/snn_oracle.c -- TEST INFRASTRUCTURE ONLY.  See snn_oracle.h for scope,
 * parity status and the canonical choices.  Every function cites the reference
 * lines (relative to /root/reference/backend/src/) whose arithmetic it restates;
 * f32 operation order is kept literally (build: -ffp-contract=off, no fast-math).
 */
#include "snn_oracle.h"
#include "snn_oracle_math.h"

#include <stddef.h>
#include <stdlib.h>

static float custom_refractoriness_effect(const snn_o_net *n, uint32_t s);
typedef struct { snn_o_net *n; uint32_t q, spiking_prev; } program_ctx;   /* the neuron an on_electrochemical_iteration runs on */
static float program_run_ctx(const int32_t *c, const float *consts, uint32_t pc, float *slot, int apply_diffs,
                             const program_ctx *ctx);
static float program_run(const int32_t *c, const float *consts, uint32_t pc, float *slot, int apply_diffs)
{
    return program_run_ctx(c, consts, pc, slot, apply_diffs, NULL);
}

/* ---------- small helpers ---------- */

/* f32::max / f32::min as Rust defines them: a NaN operand yields the other one. */
static inline float o_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return (a > b) ? a : b;
}
static inline float o_min(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return (a < b) ? a : b;
}
static inline float o_abs(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u &= 0x7FFFFFFFu;
    memcpy(&x, &u, 4);
    return x;
}

/* spike_train/mod.rs:380-388 (the reference's GPU generator) */
uint32_t snn_o_xorshift32(uint32_t seed)
{
    uint32_t x = seed;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

/* DeltaDiracRefractoriness::get_effect, spike_train/mod.rs:67-88:
 *   a * ((-1. / (k / dt)) * time_difference.powf(2.)).exp() + v_resting
 * powf(2.) is folded to x*x by LLVM without fast-math. */
float snn_o_delta_dirac_effect(int64_t timestep, int32_t last_firing_time,
                               float v_th, float v_resting, float k, float dt)
{
    float a = v_th - v_resting;
    float td = (float)(timestep - (int64_t)last_firing_time);
    return a * snn_o_expf((-1.0f / (k / dt)) * (td * td)) + v_resting;
}

/* ExponentialDecayRefractoriness::get_effect, spike_train/mod.rs:164-178:
 *   a * ((-1. / (k / dt)) * time_difference).exp() + v_resting */
float snn_o_exponential_decay_effect(int64_t timestep, int32_t last_firing_time,
                                     float v_th, float v_resting, float k, float dt)
{
    float a = v_th - v_resting;
    float td = (float)(timestep - (int64_t)last_firing_time);
    return a * snn_o_expf((-1.0f / (k / dt)) * td) + v_resting;
}

/* STDP::update_weight, plasticity/mod.rs:45-66 (returns delta_w) */
float snn_o_stdp_delta(int32_t t_pre, int32_t t_post, float a_plus, float a_minus,
                       float tau_plus, float tau_minus, float dt)
{
    if (t_pre < 0 || t_post < 0) return 0.0f;          /* either side None */
    float tp = (float)t_pre, tq = (float)t_post;
    if (tp < tq)
        return a_plus * snn_o_expf(-1.0f * o_abs((tp - tq) * dt) / tau_plus);
    if (tp > tq)
        return -1.0f * a_minus * snn_o_expf(-1.0f * o_abs((tq - tp) * dt) / tau_minus);
    return 0.0f;
}

float snn_o_expf_export(float x) { return snn_o_expf(x); }
float snn_o_pow3f_export(float x) { return snn_o_pow3f(x); }
float snn_o_pow4f_export(float x) { return snn_o_pow4f(x); }
float snn_o_tanhf_export(float x) { return snn_o_tanhf(x); }
float snn_o_sinhf_export(float x) { return snn_o_sinhf(x); }
float snn_o_coshf_export(float x) { return snn_o_coshf(x); }
float snn_o_powif_export(float x, int n) { return snn_o_powif(x, n); }
float snn_o_sinf_export(float x) { return snn_o_sinf(x); }
float snn_o_cosf_export(float x) { return snn_o_cosf(x); }
float snn_o_tanf_export(float x) { return snn_o_tanf(x); }
float snn_o_powf_export(float x, float y) { return snn_o_powf(x, y); }

/* out[i] = f(the float whose bit pattern is first + i * stride): which = 0 expf, 1 powf(x, 3.), 2 powf(x, 4.),
 * 3 powf(x, y).  All cores; the GPU parity test walks the whole 2^32 pattern space in chunks with it. */
void snn_o_math_bits(int which, uint32_t first, uint32_t stride, uint64_t count, float y, float *out)
{
    /* (a handful of values is evaluated by the caller's thread: a team of as many threads as the box shows CPUs -- 256 on the
     * GPU boxes, of which a container may use 16 -- costs tens of milliseconds per region once the cores are oversubscribed) */
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (uint64_t i = 0; i < count; i++) {
        const float x = snn_o_asfloat(first + (uint32_t)i * stride);
        out[i] = which == 0 ? snn_o_expf(x) : which == 1 ? snn_o_pow3f(x) : which == 2 ? snn_o_pow4f(x) : snn_o_powf(x, y);
    }
}

/* ---------- synthetic data ---------- */

/* splitmix64 finaliser over (seed, index); upper 32 bits */
uint32_t snn_o_hash32(uint64_t seed, uint64_t index)
{
    uint64_t x = index + seed * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}

float snn_o_uniform(uint64_t seed, uint64_t index, float lo, float hi)
{
    float u = (float)(snn_o_hash32(seed, index) >> 8) * (1.0f / 16777216.0f);  /* [0,1) exact */
    return lo + (hi - lo) * u;
}

void snn_o_fill_graph(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                      uint64_t seed, float lo, float hi, int with_diagonal)
{
    for (uint32_t p = 0; p < n_tot; ++p)
        for (uint32_t q = 0; q < n_neurons; ++q) {
            size_t i = (size_t)p * n_neurons + q;
            int c = with_diagonal || p != q;
            connections[i] = (uint8_t)c;
            weights[i] = c ? snn_o_uniform(seed, i, lo, hi) : 0.0f;
        }
}

/* columns [col0, col0+ncols) of the same synthetic graph, stored [n_tot][ncols] */
void snn_o_fill_graph_window(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                             uint32_t col0, uint32_t ncols, uint64_t seed, float lo, float hi,
                             int with_diagonal)
{
#if defined(_OPENMP)
    #pragma omp parallel for schedule(static)
#endif
    for (int64_t p = 0; p < (int64_t)n_tot; ++p)
        for (uint32_t j = 0; j < ncols; ++j) {
            uint32_t q = col0 + j;
            size_t i = (size_t)p * ncols + j;
            int c = with_diagonal || (uint32_t)p != q;
            connections[i] = (uint8_t)c;
            weights[i] = c ? snn_o_uniform(seed, (uint64_t)p * n_neurons + q, lo, hi) : 0.0f;
        }
}

/* ---------- step 1: inputs ---------- */

/*
 * Postsynaptic neurons [q0, q0+nq), nq <= SNN_O_QBLOCK, advanced together so that each 64-byte line
 * of the row-major matrix is read once; per column the arithmetic is exactly the sequential chunked
 * sum of the header.
 *   electrical: Lattice::calculate_internal_electrical_input_from_positions  neuron/mod.rs:702-730
 *               LatticeNetwork::calculate_electrical_input_from_positions    neuron/mod.rs:2115-2167
 *               gap_junction mod.rs:54-60, spike_train_gap_junction mod.rs:119-137
 *   chemical:   calculate_*_neurotransmitter_input_from_positions mod.rs:733-754 / 2169-2210,
 *               weight_/aggregate_neurotransmitter_concentration(s) iterate_and_spike/mod.rs:2837-2866
 */
#define SNN_O_QBLOCK 16

static void inputs_block(snn_o_net *n, uint32_t q0, uint32_t nq)
{
    const uint32_t nn = n->n_neurons;
    const uint32_t n_tot = nn + n->n_cells;
    const size_t ld = n->w_ld ? n->w_ld : nn;
    const uint32_t col0 = n->w_col0;

    float vq[SNN_O_QBLOCK], gq[SNN_O_QBLOCK], sum[SNN_O_QBLOCK], part[SNN_O_QBLOCK];
    float tsum[SNN_O_K][SNN_O_QBLOCK], tpart[SNN_O_K][SNN_O_QBLOCK];
    uint32_t n_in[SNN_O_QBLOCK], tcnt[SNN_O_K][SNN_O_QBLOCK];
    for (uint32_t j = 0; j < nq; ++j) {
        vq[j] = n->current_voltage[q0 + j];
        gq[j] = n->gap_conductance[q0 + j];
        sum[j] = 0.0f; n_in[j] = 0;
        for (int k = 0; k < SNN_O_K; ++k) { tsum[k][j] = 0.0f; tcnt[k][j] = 0; }
    }

    for (uint32_t c0 = 0; c0 < n_tot; c0 += SNN_O_CHUNK) {
        uint32_t c1 = c0 + SNN_O_CHUNK;
        if (c1 > n_tot) c1 = n_tot;
        for (uint32_t j = 0; j < nq; ++j) {
            part[j] = 0.0f;
            for (int k = 0; k < SNN_O_K; ++k) tpart[k][j] = 0.0f;
        }
        for (uint32_t p = c0; p < c1; ++p) {
            const float *wrow = n->weights + (size_t)p * ld + (q0 - col0);
            const uint8_t *crow = n->connections + (size_t)p * ld + (q0 - col0);
            /* presynaptic side, once per row */
            int kind = 0;                 /* 0 neuron, 1 silent spike train, 2 fired spike train */
            float pv = 0.0f;
            uint32_t s = 0;
            if (p < nn) {
                pv = n->current_voltage[p];
            } else {
                s = p - nn;
                if (n->st_last_firing_time[s] < 0) { kind = 1; pv = n->st_v_resting[s]; }
                else {
                    kind = 2;
                    pv = (n->st_refractoriness && n->st_refractoriness[s] == 2) ? custom_refractoriness_effect(n, s)
                        : (n->st_refractoriness && n->st_refractoriness[s])
                        ? snn_o_exponential_decay_effect(n->clock, n->st_last_firing_time[s], n->st_v_th[s],
                                                         n->st_v_resting[s], n->st_k[s], n->st_dt[s])
                        : snn_o_delta_dirac_effect(n->clock, n->st_last_firing_time[s], n->st_v_th[s],
                                                   n->st_v_resting[s], n->st_k[s], n->st_dt[s]);
                }
            }
            if (n->electrical) {
                if (kind == 0) {
                    for (uint32_t j = 0; j < nq; ++j)
                        if (crow[j]) part[j] += (gq[j] * (pv - vq[j])) * wrow[j];
                } else if (kind == 1) {
                    for (uint32_t j = 0; j < nq; ++j)      /* no conductance factor, mod.rs:126-128 */
                        if (crow[j]) part[j] += pv * wrow[j];
                } else {
                    for (uint32_t j = 0; j < nq; ++j)
                        if (crow[j]) part[j] += (gq[j] * pv) * wrow[j];
                }
            }
            for (uint32_t j = 0; j < nq; ++j) n_in[j] += crow[j] ? 1u : 0u;
            if (n->chemical) {
                for (int k = 0; k < SNN_O_K; ++k) {
                    uint32_t flag; float t;
                    if (p < nn) { flag = n->nt_flags[(size_t)p * SNN_O_K + k]; t = n->nt_t[(size_t)p * SNN_O_K + k]; }
                    else { flag = n->st_nt_flags[(size_t)s * SNN_O_K + k]; t = n->st_nt_t[(size_t)s * SNN_O_K + k]; }
                    if (!flag) continue;
                    for (uint32_t j = 0; j < nq; ++j)
                        if (crow[j]) { tpart[k][j] += t * wrow[j]; ++tcnt[k][j]; }
                }
            }
        }
        for (uint32_t j = 0; j < nq; ++j) {
            sum[j] += part[j];
            for (int k = 0; k < SNN_O_K; ++k) tsum[k][j] += tpart[k][j];
        }
    }

    for (uint32_t j = 0; j < nq; ++j) {
        const uint32_t q = q0 + j;
        if (n->electrical) {
            float averager = (n_in[j] == 0) ? 1.0f : (float)n_in[j];   /* mod.rs:722-727 */
            n->input_current[q] = sum[j] / averager;
        } else {
            n->input_current[q] = 0.0f;                                /* mod.rs:929-931 */
        }
        if (n->chemical) {
            for (int k = 0; k < SNN_O_K; ++k) {
                n->input_count[(size_t)q * SNN_O_K + k] = (float)tcnt[k][j];
                n->input_t[(size_t)q * SNN_O_K + k] = tcnt[k][j] ? tsum[k][j] / (float)tcnt[k][j] : 0.0f;
            }
        }
    }
}

void snn_o_inputs_range(snn_o_net *n, uint32_t q0, uint32_t q1)
{
    const int64_t nblocks = ((int64_t)q1 - q0 + SNN_O_QBLOCK - 1) / SNN_O_QBLOCK;
#if defined(_OPENMP)
    int nt = n->n_threads > 1 ? n->n_threads : 1;
    #pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t b = 0; b < nblocks; ++b) {
        uint32_t b0 = q0 + (uint32_t)b * SNN_O_QBLOCK;
        uint32_t nq = (q1 - b0 < SNN_O_QBLOCK) ? (q1 - b0) : SNN_O_QBLOCK;
        inputs_block(n, b0, nq);
    }
}

void snn_o_inputs(snn_o_net *n) { snn_o_inputs_range(n, 0, n->n_neurons); }

/*
 * The same electrical input sums for postsynaptic neurons [q0, q1), arranged for memory bandwidth on many cores (the
 * all-core CPU baseline of bench.py; rayon's par_iter over postsynaptic neurons in the reference,
 * neuron/mod.rs:775-790): the matrix is cut into tiles of `block` adjacent columns x one reduction chunk of 256
 * presynaptic rows, every thread streams whole `block`-wide row segments (block = 1024 floats = one 4 KiB page per
 * row), the inner loop runs across the columns of the segment (one independent sequential sum per column -- SIMD
 * across columns does not change any column's order of additions), and the chunk partials are then added per
 * column in ascending chunk order from 0.0f.  Bit-identical to snn_o_inputs_range (tests/test_oracle_vs_numpy.py).
 * Restricted to what the baseline workload needs: electrical synapses, every presynaptic row a neuron; anything
 * else takes the general routine.
 */
void snn_o_inputs_tiled(snn_o_net *n, uint32_t q0, uint32_t q1, uint32_t block)
{
    const uint32_t nn = n->n_neurons;
    if (n->chemical || !n->electrical || n->n_cells || q1 <= q0 || block == 0) { snn_o_inputs_range(n, q0, q1); return; }
    const size_t ld = n->w_ld ? n->w_ld : nn;
    const uint32_t col0 = n->w_col0;
    const uint32_t ncols = q1 - q0;
    const int64_t nblk = (ncols + block - 1) / block;
    const int64_t nchunks = (nn + SNN_O_CHUNK - 1) / SNN_O_CHUNK;
    float *part = (float *)malloc((size_t)nchunks * ncols * sizeof(float));
    uint32_t *pcnt = (uint32_t *)malloc((size_t)nchunks * ncols * sizeof(uint32_t));
    if (!part || !pcnt) { free(part); free(pcnt); snn_o_inputs_range(n, q0, q1); return; }
#if defined(_OPENMP)
    int nt = n->n_threads > 1 ? n->n_threads : 1;
    #pragma omp parallel for collapse(2) schedule(static) num_threads(nt)
#endif
    for (int64_t b = 0; b < nblk; ++b)
        for (int64_t c = 0; c < nchunks; ++c) {
            const uint32_t j0 = (uint32_t)b * block;
            const uint32_t w = (ncols - j0 < block) ? ncols - j0 : block;
            float *acc = part + (size_t)c * ncols + j0;
            uint32_t *cnt = pcnt + (size_t)c * ncols + j0;
            const float *vq = n->current_voltage + q0 + j0, *gq = n->gap_conductance + q0 + j0;
            for (uint32_t j = 0; j < w; ++j) { acc[j] = 0.0f; cnt[j] = 0; }
            const uint32_t p1 = ((uint32_t)(c + 1) * SNN_O_CHUNK < nn) ? (uint32_t)(c + 1) * SNN_O_CHUNK : nn;
            for (uint32_t p = (uint32_t)c * SNN_O_CHUNK; p < p1; ++p) {
                const float *wrow = n->weights + (size_t)p * ld + (q0 - col0) + j0;
                const uint8_t *crow = n->connections + (size_t)p * ld + (q0 - col0) + j0;
                const float pv = n->current_voltage[p];
#if defined(_OPENMP)
                #pragma omp simd
#endif
                for (uint32_t j = 0; j < w; ++j) {
                    const float term = (gq[j] * (pv - vq[j])) * wrow[j];
                    acc[j] = crow[j] ? acc[j] + term : acc[j];
                    cnt[j] += crow[j] ? 1u : 0u;
                }
            }
        }
#if defined(_OPENMP)
    #pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t j = 0; j < (int64_t)ncols; ++j) {
        float sum = 0.0f;
        uint32_t n_in = 0;
        for (int64_t c = 0; c < nchunks; ++c) { sum += part[(size_t)c * ncols + j]; n_in += pcnt[(size_t)c * ncols + j]; }
        n->input_current[q0 + j] = sum / ((n_in == 0) ? 1.0f : (float)n_in);      /* mod.rs:722-727 */
    }
    free(part);
    free(pcnt);
}

/* first touch of a [n_tot][ncols] window by the threads that will stream it in snn_o_inputs_tiled (column blocks) */
void snn_o_fill_graph_window_blocked(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                                     uint32_t col0, uint32_t ncols, uint32_t block, uint64_t seed, float lo, float hi,
                                     int with_diagonal, int n_threads)
{
    const int64_t nblk = (ncols + block - 1) / block;
    const int64_t nchunks = (n_tot + SNN_O_CHUNK - 1) / SNN_O_CHUNK;
#if defined(_OPENMP)
    int nt = n_threads > 1 ? n_threads : 1;
    #pragma omp parallel for collapse(2) schedule(static) num_threads(nt)
#endif
    for (int64_t b = 0; b < nblk; ++b)
        for (int64_t c = 0; c < nchunks; ++c) {
            const uint32_t j0 = (uint32_t)b * block;
            const uint32_t j1 = (j0 + block < ncols) ? j0 + block : ncols;
            const uint32_t p1 = ((uint32_t)(c + 1) * SNN_O_CHUNK < n_tot) ? (uint32_t)(c + 1) * SNN_O_CHUNK : n_tot;
            for (uint32_t p = (uint32_t)c * SNN_O_CHUNK; p < p1; ++p)
                for (uint32_t j = j0; j < j1; ++j) {
                    const uint32_t q = col0 + j;
                    const size_t i = (size_t)p * ncols + j;
                    const int e = with_diagonal || p != q;
                    connections[i] = (uint8_t)e;
                    weights[i] = e ? snn_o_uniform(seed, (uint64_t)p * n_neurons + q, lo, hi) : 0.0f;
                }
        }
}

/*
 * Step 1 over a SPARSE graph given as CSR by postsynaptic neuron: row q holds the presynaptic indices
 * pre[row_ptr[q] .. row_ptr[q+1]) in ascending order with their weights -- every stored entry is a Some(w) edge of
 * the AdjacencyMatrix (graph/mod.rs:139-297), everything else None.  Same arithmetic as inputs_block: the canonical
 * chunk of an entry is pre / SNN_O_CHUNK, a chunk's partial starts at 0.0f, partials are added in ascending chunk order
 * (chunks without an entry would add a 0.0f partial: no change), n_in = the row length (mod.rs:722-729).
 * BASELINE configs[4] (4 x 512^2 neurons + Poisson cells) is only representable this way: dense it is 4.4 TB.
 */
void snn_o_inputs_csr(snn_o_net *n, const uint64_t *row_ptr, const uint32_t *pre, const float *w,
                      uint32_t q0, uint32_t q1)
{
    const uint32_t nn = n->n_neurons;
    /* presynaptic value of every spike-train cell at this clock, once (spike_train_gap_junction mod.rs:119-137) */
    float *cell_v = n->n_cells ? (float *)malloc(sizeof(float) * n->n_cells) : NULL;
    for (uint32_t s = 0; s < n->n_cells; ++s) {
        if (n->st_last_firing_time[s] < 0) { cell_v[s] = n->st_v_resting[s]; continue; }
        cell_v[s] = (n->st_refractoriness && n->st_refractoriness[s] == 2) ? custom_refractoriness_effect(n, s)
            : (n->st_refractoriness && n->st_refractoriness[s])
            ? snn_o_exponential_decay_effect(n->clock, n->st_last_firing_time[s], n->st_v_th[s], n->st_v_resting[s],
                                             n->st_k[s], n->st_dt[s])
            : snn_o_delta_dirac_effect(n->clock, n->st_last_firing_time[s], n->st_v_th[s], n->st_v_resting[s],
                                       n->st_k[s], n->st_dt[s]);
    }
#if defined(_OPENMP)
    int nt = n->n_threads > 1 ? n->n_threads : 1;
    #pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t qq = q0; qq < (int64_t)q1; ++qq) {
        const uint32_t q = (uint32_t)qq;
        const float vq = n->current_voltage[q], gq = n->gap_conductance[q];
        float sum = 0.0f, part = 0.0f;
        float tsum[SNN_O_K] = {0.0f, 0.0f, 0.0f}, tpart[SNN_O_K] = {0.0f, 0.0f, 0.0f};
        uint32_t tcnt[SNN_O_K] = {0, 0, 0};
        int64_t chunk = -1;
        for (uint64_t e = row_ptr[q]; e < row_ptr[q + 1]; ++e) {
            const uint32_t p = pre[e];
            const int64_t c = p / SNN_O_CHUNK;
            if (c != chunk) {
                if (chunk >= 0) {
                    sum += part;
                    for (int k = 0; k < SNN_O_K; ++k) tsum[k] += tpart[k];
                }
                part = 0.0f;
                for (int k = 0; k < SNN_O_K; ++k) tpart[k] = 0.0f;
                chunk = c;
            }
            if (n->electrical) {
                if (p < nn) part += (gq * (n->current_voltage[p] - vq)) * w[e];
                else if (n->st_last_firing_time[p - nn] < 0) part += cell_v[p - nn] * w[e];   /* no conductance factor */
                else part += (gq * cell_v[p - nn]) * w[e];
            }
            if (n->chemical) {
                for (int k = 0; k < SNN_O_K; ++k) {
                    uint32_t flag; float t;
                    if (p < nn) { flag = n->nt_flags[(size_t)p * SNN_O_K + k]; t = n->nt_t[(size_t)p * SNN_O_K + k]; }
                    else { flag = n->st_nt_flags[(size_t)(p - nn) * SNN_O_K + k]; t = n->st_nt_t[(size_t)(p - nn) * SNN_O_K + k]; }
                    if (!flag) continue;
                    tpart[k] += t * w[e];
                    ++tcnt[k];
                }
            }
        }
        if (chunk >= 0) {
            sum += part;
            for (int k = 0; k < SNN_O_K; ++k) tsum[k] += tpart[k];
        }
        const uint64_t n_in = row_ptr[q + 1] - row_ptr[q];
        n->input_current[q] = n->electrical ? sum / ((n_in == 0) ? 1.0f : (float)n_in) : 0.0f;
        if (n->chemical) {
            for (int k = 0; k < SNN_O_K; ++k) {
                n->input_count[(size_t)q * SNN_O_K + k] = (float)tcnt[k];
                n->input_t[(size_t)q * SNN_O_K + k] = tcnt[k] ? tsum[k] / (float)tcnt[k] : 0.0f;
            }
        }
    }
    free(cell_v);
}

/* ---------- step 2: neuron update ---------- */

/* NeurotransmitterKinetics::apply_t_change: Approximate iterate_and_spike/mod.rs:193-196,
 * Destexhe :148-150.  `spiking` is what NeurotransmittersIntermediate carries
 * (intermediate_delegate/mod.rs:17-23). */
/* exp_decay, iterate_and_spike/mod.rs:345-347 */
static inline float exp_decay(float x, float l, float dt)
{
    return -x * snn_o_expf(dt / -l);
}

static inline float nt_apply(int kind, float t, float t_max, float clearance, float v_p, float k_p,
                             float voltage, uint32_t spiking, float dt)
{
    const float s = spiking ? 1.0f : 0.0f;
    if (kind == SNN_O_NT_DESTEXHE)
        return t_max / (1.0f + snn_o_expf(-(voltage - v_p) / k_p));
    if (kind == SNN_O_NT_DISCRETE_SPIKE)            /* DiscreteSpikeNeurotransmitter :300-302 */
        return t_max * s;
    if (kind == SNN_O_NT_EXPONENTIAL_DECAY)         /* ExponentialDecayNeurotransmitter :350-354, clearance = decay_constant */
        t += exp_decay(t, clearance, dt) + (s * t_max);
    else
        t += dt * -clearance * t + (s * t_max);
    return o_min(t_max, o_max(t, 0.0f));
}

/* generated apply_t_change, build_test/nb_macro/src/lib.rs:6489-6498: cell i of `vars` ([nt_nvars][count]) */
static float custom_nt_apply(const snn_o_net *n, float *vars, size_t count, size_t i, float t, float voltage,
                             uint32_t spiking, float dt)
{
    float slot[5 + 8];
    slot[0] = t; slot[1] = spiking ? 1.0f : 0.0f; slot[2] = dt; slot[3] = voltage; slot[4] = 0.0f;
    for (uint32_t j = 0; j < n->nt_nvars; ++j) slot[5 + j] = vars[(size_t)j * count + i];
    program_run(n->nt_code, n->nt_consts, 0, slot, 1);
    for (uint32_t j = 0; j < n->nt_nvars; ++j) vars[(size_t)j * count + i] = slot[5 + j];
    return slot[0];
}

static inline void neuron_nt_update(snn_o_net *n, uint32_t q, float voltage, uint32_t spiking_prev)
{
    for (int k = 0; k < SNN_O_K; ++k) {
        size_t i = (size_t)q * SNN_O_K + k;
        if (!n->nt_flags || !n->nt_flags[i]) continue;
        if (n->nt_kind == SNN_O_NT_CUSTOM) {
            n->nt_t[i] = custom_nt_apply(n, n->nt_custom_vars, (size_t)n->n_neurons * SNN_O_K, i, n->nt_t[i], voltage,
                                         spiking_prev, n->dt[q]);
            continue;
        }
        n->nt_t[i] = nt_apply(n->nt_kind, n->nt_t[i], n->nt_t_max[i],
                              n->nt_clearance ? n->nt_clearance[i] : 0.0f,
                              n->nt_v_p ? n->nt_v_p[i] : 0.0f, n->nt_k_p ? n->nt_k_p[i] : 1.0f,
                              voltage, spiking_prev, n->dt[q]);
    }
}

/* Ionotropic::update_receptor_kinetics + set_receptor_currents, iterate_and_spike/mod.rs:1186-1284;
 * kinetics :404-406 (Destexhe) / :435-437 (Approximate); currents :1103-1105, 1132-1137, 1164-1166.
 * A type absent from the aggregated input (count 0) leaves r untouched. */
static inline void receptors_kinetics(snn_o_net *n, uint32_t q)
{
    const float dt = n->dt[q];
    if (n->model == SNN_O_CUSTOM && n->rx_ntypes && n->rx_multi) {
        /* a generated receptor set with several states per type (`receptors: a, b`, nb_macro lib.rs:7306-7316,
         * 7391-7404): apply_r_change(t, dt) of every state of a type present in the input AND in the set; the states and
         * their kinetics variables are set variables (slots 5..), t is slot 3, dt slot 2 */
        float slot[5 + 32];
        slot[0] = slot[1] = slot[4] = 0.0f; slot[2] = dt;
        for (uint32_t j = 0; j < n->rx_nvars; ++j) slot[5 + j] = n->rx_vars[(size_t)j * n->n_neurons + q];
        for (uint32_t k = 0; k < n->rx_ntypes; ++k) {
            size_t i = (size_t)q * SNN_O_K + k;
            if (!n->rc_flags[i] || n->input_count[i] == 0.0f) continue;
            slot[3] = n->input_t[i];
            program_run(n->rx_code, n->rx_consts, n->rx_kin_section[k], slot, 1);
        }
        for (uint32_t j = 0; j < n->rx_nvars; ++j) n->rx_vars[(size_t)j * n->n_neurons + q] = slot[5 + j];
        return;
    }
    for (int k = 0; k < SNN_O_K; ++k) {
        size_t i = (size_t)q * SNN_O_K + k;
        if (!n->rc_flags[i]) continue;
        if (n->input_count[i] != 0.0f) {
            float t = n->input_t[i];
            if (n->rc_kind == SNN_O_RC_CUSTOM) {
                /* generated apply_r_change, build_test/nb_macro/src/lib.rs:6778-6786 */
                const size_t count = (size_t)n->n_neurons * SNN_O_K;
                float slot[5 + 8];
                slot[0] = n->rc_r[i]; slot[1] = t; slot[2] = dt; slot[3] = 0.0f; slot[4] = 0.0f;
                for (uint32_t j = 0; j < n->rc_nvars; ++j) slot[5 + j] = n->rc_custom_vars[(size_t)j * count + i];
                program_run(n->rc_code, n->rc_consts, 0, slot, 1);
                for (uint32_t j = 0; j < n->rc_nvars; ++j) n->rc_custom_vars[(size_t)j * count + i] = slot[5 + j];
                n->rc_r[i] = slot[0];
            } else if (n->rc_kind == SNN_O_RC_DESTEXHE) {
                float r = n->rc_r[i];
                n->rc_r[i] = r + (n->rc_alpha[i] * t * (1.0f - r) - n->rc_beta[i] * r) * dt;
            } else if (n->rc_kind == SNN_O_RC_EXPONENTIAL_DECAY) {
                /* ExponentialDecayReceptor::apply_r_change :510-513; rc_alpha = r_max, rc_beta = decay_constant */
                float r = n->rc_r[i];
                r += exp_decay(r, n->rc_beta[i], dt) + t;
                n->rc_r[i] = o_min(n->rc_alpha[i], o_max(r, 0.0f));
            } else {
                n->rc_r[i] = t;
            }
        }
    }
}

static inline void receptors_set_currents(snn_o_net *n, uint32_t q, float v_old)
{
    if (n->model == SNN_O_CUSTOM && n->rx_ntypes) {
        /* generated receptor set, build_test/nb_macro/src/lib.rs:7512-7543: every receptor present iterates, in
         * declaration order, over the set's variables */
        float slot[5 + 32];
        slot[0] = v_old; slot[2] = slot[3] = slot[4] = 0.0f;
        for (uint32_t j = 0; j < n->rx_nvars; ++j) slot[5 + j] = n->rx_vars[(size_t)j * n->n_neurons + q];
        for (uint32_t k = 0; k < n->rx_ntypes; ++k) {
            size_t i = (size_t)q * SNN_O_K + k;
            if (!n->rc_flags[i]) continue;
            slot[1] = n->rc_r[i];
            program_run(n->rx_code, n->rx_consts, n->rx_section[k], slot, 0);
        }
        for (uint32_t j = 0; j < n->rx_nvars; ++j) n->rx_vars[(size_t)j * n->n_neurons + q] = slot[5 + j];
        return;
    }
    for (int k = 0; k < SNN_O_K; ++k) {
        size_t i = (size_t)q * SNN_O_K + k;
        if (!n->rc_flags[i]) continue;
        float r = n->rc_r[i];
        if (k == 1) {   /* NMDA */
            n->rc_current[i] = ((1.0f / (1.0f + ((snn_o_expf(-0.062f * v_old) * n->rc_mg[i]) / 3.75f))
                                 * n->rc_g[i]) * r) * (v_old - n->rc_e[i]);
        } else {
            n->rc_current[i] = (n->rc_g[i] * r) * (v_old - n->rc_e[i]);
        }
    }
}

/* Ionotropic::get_receptor_currents, iterate_and_spike/mod.rs:1286-1304 */
static inline void receptors_update(snn_o_net *n, uint32_t q, float v_old)
{
    receptors_kinetics(n, q);
    receptors_set_currents(n, q, v_old);
}

static inline float receptor_currents_scaled(const snn_o_net *n, uint32_t q, float dt, float c_m)
{
    float total = 0.0f;
    if (n->model == SNN_O_CUSTOM && n->rx_ntypes) {      /* lib.rs:7546-7566 */
        for (uint32_t k = 0; k < n->rx_ntypes; ++k) {
            if (n->rx_current_index[k] < 0 || !n->rc_flags[(size_t)q * SNN_O_K + k]) continue;
            total += n->rx_vars[(size_t)n->rx_current_index[k] * n->n_neurons + q];
        }
        return total * (dt / c_m);
    }
    if (n->rc_flags) {
        for (int k = 0; k < SNN_O_K; ++k) {
            size_t i = (size_t)q * SNN_O_K + k;
            if (n->rc_flags[i]) total += n->rc_current[i];
        }
    }
    return total * (dt / c_m);
}

static inline float receptor_currents(const snn_o_net *n, uint32_t q)
{
    return receptor_currents_scaled(n, q, n->dt[q], n->c_m[q]);
}

/* The firing-rate bookkeeping both BCM cells share (BCMIzhikevichNeuron::iterate_and_spike
 * integrate_and_fire/mod.rs:1458-1469 / :1484-1495, BCMPoissonNeuron::iterate spike_train/mod.rs:943-954): num_spikes
 * is never reset by the reference.  `rate_per_dt`: the electrical-only neuron path and the spike train divide by
 * (window * dt), the neuron's neurotransmission path by the window alone. */
static inline void bcm_window_update(float *clock, float window, float dt, uint32_t num_spikes, uint32_t period,
                                     float *current_activity, float *average_activity, int rate_per_dt)
{
    *clock += dt;
    if (*clock >= window) {
        *clock = 0.0f;
        *current_activity = rate_per_dt ? (float)num_spikes / (window * dt) : (float)num_spikes / window;
        *average_activity -= *average_activity / (float)period;
        *average_activity += *current_activity / (float)period;
    }
}

/* IzhikevichNeuron, integrate_and_fire/mod.rs:1222-1267 via impl_iterate_and_spike! :217-255 */
static uint32_t step_izhikevich(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], w = n->w_value[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float dv = (0.04f * (v * v) + 5.0f * v + 140.0f - w + i) * (dt / n->c_m[q]);
    float dw = (n->a[q] * (n->b[q] * v - w)) * (dt / n->tau_m[q]);

    float v_new, w_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }
    w_new = w + dw;

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->c[q];
        w_new += n->d[q];
    }
    n->current_voltage[q] = v_new;
    n->w_value[q] = w_new;
    return spike;
}

/* LeakyIntegrateAndFireNeuron, integrate_and_fire/mod.rs:173-215, handle_spiking :87-102 */
static uint32_t step_lif(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float dv = ((n->leak_constant[q] * (v - n->e_l[q])) +
                (n->integration_constant[q] * (i / n->g_l[q]))) * (dt / n->tau_m[q]);
    float v_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    float rc = n->refractory_count[q];
    if (rc > 0.0f) {
        v_new = n->v_reset[q];
        rc -= 1.0f;
    } else if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->v_reset[q];
        rc = n->tref[q] / dt;
    }
    n->refractory_count[q] = rc;
    n->current_voltage[q] = v_new;
    return spike;
}

/* QuadraticIntegrateAndFireNeuron, integrate_and_fire/mod.rs:324-365, handle_spiking :87-102 */
static uint32_t step_qif(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float dv = ((n->qif_alpha[q] * (v - n->v_reset[q]) * (v - n->qif_v_c[q])) +
                n->integration_constant[q] * i) * (dt / n->tau_m[q]);
    float v_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    float rc = n->refractory_count[q];
    if (rc > 0.0f) {
        v_new = n->v_reset[q];
        rc -= 1.0f;
    } else if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->v_reset[q];
        rc = n->tref[q] / dt;
    }
    n->refractory_count[q] = rc;
    n->current_voltage[q] = v_new;
    return spike;
}

/* SimpleLeakyIntegrateAndFire, integrate_and_fire/mod.rs:1577-1630 */
static uint32_t step_simple_lif(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float dv = (n->slif_g[q] * (v - n->slif_e[q]) + i) * dt;
    float v_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->v_reset[q];
    }
    n->current_voltage[q] = v_new;
    return spike;
}

/* AdaptiveLeakyIntegrateAndFireNeuron (exponential = 0) integrate_and_fire/mod.rs:1033-1049 and
 * AdaptiveExpLeakyIntegrateAndFireNeuron (exponential = 1) :1132-1155; adaptive_get_dw_change :1003-1010,
 * adaptive_handle_spiking :1014-1030 */
static uint32_t step_adaptive(snn_o_net *n, uint32_t q, int exponential)
{
    const float v = n->current_voltage[q], w = n->w_value[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float acc = n->leak_constant[q] * (v - n->e_l[q]);
    if (exponential)
        acc = acc + (n->slope_factor[q] * snn_o_expf((v - n->v_th[q]) / n->slope_factor[q]));
    float dv = (acc + (n->integration_constant[q] * (i / n->g_l[q])) - (w / n->g_l[q])) * (dt / n->c_m[q]);
    float dw = (n->adp_alpha[q] * (v - n->e_l[q]) - w) * (dt / n->tau_m[q]);

    float v_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }
    float w_new = w + dw;

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    float rc = n->refractory_count[q];
    if (rc > 0.0f) {
        v_new = n->v_reset[q];
        rc -= 1.0f;
    } else if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->v_reset[q];
        w_new += n->adp_beta[q];
        rc = n->tref[q] / dt;
    }
    n->refractory_count[q] = rc;
    n->current_voltage[q] = v_new;
    n->w_value[q] = w_new;
    return spike;
}

/* LeakyIzhikevichNeuron, integrate_and_fire/mod.rs:1336-1356 (dw and spike handling are Izhikevich's, :1222-1247).
 * `current_voltage.powf(2.0)` is taken as v * v (the correctly rounded square). */
static uint32_t step_leaky_izhikevich(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], w = n->w_value[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);

    float dv = (0.04f * (v * v) + 5.0f * v + 140.0f - w * (v - n->e_l[q]) + i) * (dt / n->c_m[q]);
    float dw = (n->a[q] * (n->b[q] * v - w)) * (dt / n->tau_m[q]);

    float v_new;
    if (n->chemical) {
        float neurotransmitter_dv = -receptor_currents(n, q);
        v_new = v + (dv + neurotransmitter_dv);
    } else {
        v_new = v + dv;
    }
    float w_new = w + dw;

    neuron_nt_update(n, q, v_new, spiking_prev);

    uint32_t spike = 0;
    if (v_new >= n->v_th[q]) {
        spike = 1;
        v_new = n->c[q];
        w_new += n->d[q];
    }
    n->current_voltage[q] = v_new;
    n->w_value[q] = w_new;
    return spike;
}

/* A generated model as a stack program (see snn_oracle.h).  Slots: 0 v, 1 i, 2 dt, 3 c_m, 4 gap_conductance,
 * 5.. model variables.  Values are float32; comparisons / logic leave 1.0f or 0.0f. */
enum { OP_END = 0, OP_CONST = 1, OP_LOAD = 2, OP_STORE = 3, OP_DIFF = 4, OP_NEG = 5, OP_NOT = 6, OP_ADD = 7, OP_SUB = 8,
       OP_MUL = 9, OP_DIV = 10, OP_EXP = 11, OP_EQ = 12, OP_NE = 13, OP_GE = 14, OP_LE = 15, OP_GT = 16, OP_LT = 17,
       OP_AND = 18, OP_OR = 19, OP_JZ = 20, OP_JMP = 21, OP_TANH = 22, OP_SINH = 23, OP_COSH = 24, OP_MIN = 25,
       OP_MAX = 26, OP_HEAVISIDE = 27, OP_POWI = 28, OP_MARK = 29, OP_FLUSH = 30, OP_RC_UPDATE = 31, OP_RC_SET = 32,
       OP_RC_GET = 33, OP_NT_APPLY = 34, OP_SIN = 35, OP_COS = 36, OP_TAN = 37, OP_ISNAN = 38, OP_POWF = 39,
       OP_RPOW = 40 };

static float program_run_ctx(const int32_t *c, const float *consts, uint32_t pc, float *slot, int apply_diffs,
                             const program_ctx *ctx)
{
    float stack[64], diff[32];
    uint32_t diff_slot[32];
    int sp = 0, nd = 0, mark = 0;
    for (;;) {
        int32_t op = c[pc++];
        if (op == OP_END) break;
        switch (op) {
        case OP_CONST: stack[sp++] = consts[c[pc++]]; break;
        case OP_LOAD:  stack[sp++] = slot[c[pc++]]; break;
        case OP_STORE: slot[c[pc++]] = stack[--sp]; break;
        case OP_DIFF:  diff[nd] = stack[--sp] * slot[2]; diff_slot[nd++] = (uint32_t)c[pc++]; break;   /* (expr) * dt */
        case OP_NEG:   stack[sp - 1] = -stack[sp - 1]; break;
        case OP_NOT:   stack[sp - 1] = (stack[sp - 1] != 0.0f) ? 0.0f : 1.0f; break;
        case OP_EXP:   stack[sp - 1] = snn_o_expf(stack[sp - 1]); break;
        case OP_TANH:  stack[sp - 1] = snn_o_tanhf(stack[sp - 1]); break;
        case OP_SINH:  stack[sp - 1] = snn_o_sinhf(stack[sp - 1]); break;
        case OP_COSH:  stack[sp - 1] = snn_o_coshf(stack[sp - 1]); break;
        case OP_SIN:   stack[sp - 1] = snn_o_sinf(stack[sp - 1]); break;
        case OP_COS:   stack[sp - 1] = snn_o_cosf(stack[sp - 1]); break;
        case OP_TAN:   stack[sp - 1] = snn_o_tanf(stack[sp - 1]); break;
        case OP_ISNAN: stack[sp - 1] = (stack[sp - 1] != stack[sp - 1]) ? 1.0f : 0.0f; break;
        case OP_HEAVISIDE: stack[sp - 1] = (stack[sp - 1] < 0.0f) ? 0.0f : stack[sp - 1]; break;   /* lib.rs:9176 */
        case OP_POWI:  stack[sp - 1] = snn_o_powif(stack[sp - 1], c[pc++]); break;
        /* an inlined ion channel's update_current (lib.rs:4043-4063): its own `x += dx` at the end of ITS body */
        /* the calls of an on_electrochemical_iteration, nb_macro lib.rs:2275-2293 */
        case OP_RC_UPDATE: receptors_kinetics(ctx->n, ctx->q); break;
        case OP_RC_SET:    receptors_set_currents(ctx->n, ctx->q, stack[--sp]); break;
        case OP_RC_GET:    { float cm = stack[--sp], step = stack[--sp];
                             stack[sp++] = receptor_currents_scaled(ctx->n, ctx->q, step, cm); } break;
        case OP_NT_APPLY:  neuron_nt_update(ctx->n, ctx->q, slot[0], ctx->spiking_prev); break;
        case OP_MARK:  mark = nd; break;
        case OP_FLUSH: for (int k = mark; k < nd; ++k) slot[diff_slot[k]] += diff[k]; nd = mark; break;
        case OP_JZ:    { uint32_t target = (uint32_t)c[pc++]; if (stack[--sp] == 0.0f) pc = target; } break;
        case OP_JMP:   pc = (uint32_t)c[pc]; break;
        default: {
            float b = stack[--sp], a = stack[--sp], r = 0.0f;
            switch (op) {
            case OP_ADD: r = a + b; break;
            case OP_SUB: r = a - b; break;
            case OP_MUL: r = a * b; break;
            case OP_DIV: r = a / b; break;
            case OP_EQ: r = (a == b); break;
            case OP_NE: r = (a != b); break;
            case OP_GE: r = (a >= b); break;
            case OP_LE: r = (a <= b); break;
            case OP_GT: r = (a > b); break;
            case OP_LT: r = (a < b); break;
            case OP_AND: r = (a != 0.0f && b != 0.0f); break;
            case OP_OR: r = (a != 0.0f || b != 0.0f); break;
            case OP_MIN: r = o_min(a, b); break;
            case OP_MAX: r = o_max(a, b); break;
            case OP_POWF: r = snn_o_powf(a, b); break;                 /* `a ^ b`: (a.powf(b)), nb_macro lib.rs:135 */
            case OP_RPOW: r = snn_o_powf(o_max(a, 0.0f), b); break;    /* `a r^ b`: (a.max(0.0f32).powf(b)), lib.rs:136 */
            }
            stack[sp++] = r;
        } }
    }
    if (apply_diffs)
        for (int k = 0; k < nd; ++k) slot[diff_slot[k]] += diff[k];    /* every `x += dx` after the last statement */
    return sp ? stack[sp - 1] : 0.0f;
}

static float custom_run(const snn_o_net *n, uint32_t pc, float *slot, int apply_diffs)
{
    return program_run(n->custom_code, n->custom_consts, pc, slot, apply_diffs);
}

/* generated NeuralRefractoriness::get_effect, build_test/nb_macro/src/lib.rs:5736-5750 */
static float custom_refractoriness_effect(const snn_o_net *n, uint32_t s)
{
    float slot[5 + 8];
    slot[0] = (float)(n->clock - (int64_t)n->st_last_firing_time[s]);      /* (timestep - last_firing_time) as f32 */
    slot[1] = n->st_v_th[s]; slot[2] = n->st_dt[s]; slot[3] = n->st_v_resting[s]; slot[4] = n->st_k[s];
    for (uint32_t k = 0; k < n->refr_nvars; ++k) slot[5 + k] = n->refr_vars[(size_t)k * n->n_cells + s];
    return program_run(n->refr_code, n->refr_consts, 0, slot, 0);
}

/* neuron_builder!-generated iterate_and_spike / iterate_with_neurotransmitter_and_spike,
 * build_test/nb_macro/src/lib.rs:2259-2345 (hand expansion: build_test/nb_macro/tests/lif_reference.rs) */
static uint32_t step_custom(snn_o_net *n, uint32_t q)
{
    const uint32_t nn = n->n_neurons, spiking_prev = n->is_spiking[q];
    float slot[5 + 32];
    slot[0] = n->current_voltage[q]; slot[1] = n->input_current[q]; slot[2] = n->dt[q]; slot[3] = n->c_m[q];
    slot[4] = n->gap_conductance[q];
    for (uint32_t k = 0; k < n->custom_nvars; ++k) slot[5 + k] = n->custom_vars[(size_t)k * nn + q];

    if (n->chemical && n->custom_has_chem) {
        /* on_electrochemical_iteration replaces the default sequence, lib.rs:2280-2316 */
        program_ctx ctx = { n, q, spiking_prev };
        program_run_ctx(n->custom_code, n->custom_consts, n->custom_chem_section, slot, 1, &ctx);
    } else {
        if (n->chemical) receptors_update(n, q, slot[0]);
        custom_run(n, n->custom_section[0], slot, 1);
        if (n->chemical) {
            /* the generated electrical iterate_and_spike is on_iteration + handle_spiking alone (lib.rs:2266-2272); the
             * transmitter release belongs to the neurotransmission form (lib.rs:2318-2328) */
            slot[0] -= receptor_currents(n, q);
            neuron_nt_update(n, q, slot[0], spiking_prev);
        }
    }
    uint32_t spike = custom_run(n, n->custom_section[1], slot, 0) != 0.0f;
    if (spike) custom_run(n, n->custom_section[2], slot, 0);

    n->current_voltage[q] = slot[0];
    for (uint32_t k = 0; k < n->custom_nvars; ++k) n->custom_vars[(size_t)k * nn + q] = slot[5 + k];
    return spike;
}

/* BasicGatingVariable::update, ion_channels/mod.rs:40-44 */
static inline float gate_update(float state, float alpha, float beta, float dt)
{
    float alpha_state = alpha * (1.0f - state);
    float beta_state = beta * state;
    return state + dt * (alpha_state - beta_state);
}

/* HodgkinHuxleyNeuron, hodgkin_huxley/mod.rs:156-241; channels ion_channels/mod.rs:219-316 */
static uint32_t step_hh(snn_o_net *n, uint32_t q)
{
    const float v = n->current_voltage[q], dt = n->dt[q];
    const float i = n->input_current[q];
    const uint32_t spiking_prev = n->is_spiking[q];

    if (n->chemical) receptors_update(n, q, v);            /* update_receptors :176-179 */

    /* update_gates :182-186, all at the old voltage */
    float m_a = 0.1f * ((v + 40.0f) / (1.0f - snn_o_expf(-(v + 40.0f) / 10.0f)));
    float m_b = 4.0f * snn_o_expf(-(v + 65.0f) / 18.0f);
    float h_a = 0.07f * snn_o_expf(-(v + 65.0f) / 20.0f);
    float h_b = 1.0f / (snn_o_expf(-(v + 35.0f) / 10.0f) + 1.0f);
    float m = gate_update(n->m_state[q], m_a, m_b, dt);
    float h = gate_update(n->h_state[q], h_a, h_b, dt);
    float i_na = snn_o_pow3f(m) * h * n->g_na[q] * (v - n->e_na[q]);

    float n_a = 0.01f * (v + 55.0f) / (1.0f - snn_o_expf(-(v + 55.0f) / 10.0f));
    float n_b = 0.125f * snn_o_expf(-(v + 65.0f) / 80.0f);
    float ng = gate_update(n->n_state[q], n_a, n_b, dt);
    float i_k = snn_o_pow4f(ng) * n->g_k[q] * (v - n->e_k[q]);

    float i_kl = n->g_k_leak[q] * (v - n->e_k_leak[q]);

    n->m_alpha[q] = m_a; n->m_beta[q] = m_b; n->h_alpha[q] = h_a; n->h_beta[q] = h_b;
    n->n_alpha[q] = n_a; n->n_beta[q] = n_b;
    n->m_state[q] = m; n->h_state[q] = h; n->n_state[q] = ng;
    n->na_current[q] = i_na; n->k_current[q] = i_k; n->k_leak_current[q] = i_kl;

    /* update_cell_voltage :156-166 (stored receptor currents are used even when chemical is off) */
    float i_ligand_gates = receptor_currents(n, q);
    float i_sum = i - (i_na + i_k + i_kl);
    float v_new = v + (dt * i_sum / n->c_m[q] - i_ligand_gates);

    neuron_nt_update(n, q, v_new, spiking_prev);            /* :169-171 */

    /* :207-220 */
    uint32_t increasing_right_now = v < v_new;
    uint32_t threshold_crossed = v_new > n->v_th[q];
    uint32_t spike = threshold_crossed && n->was_increasing[q] && !increasing_right_now;
    n->was_increasing[q] = increasing_right_now;
    n->current_voltage[q] = v_new;
    return spike;
}

/* Lattice::iterate* neuron/mod.rs:884-982, LatticeNetwork::iterate* :2420-2594 (neuron loop part) */
void snn_o_update_neurons(snn_o_net *n) { snn_o_update_neurons_range(n, 0, n->n_neurons); }

/* neurons [q0, q1) only: the per-shard half of a multi-GPU step (tests of the sharding protocol) */
void snn_o_update_neurons_range(snn_o_net *n, uint32_t q0, uint32_t q1)
{
    for (uint32_t q = q0; q < q1; ++q) {
        uint32_t spike;
        switch (n->model) {
        case SNN_O_LIF: spike = step_lif(n, q); break;
        case SNN_O_HH:  spike = step_hh(n, q); break;
        case SNN_O_QIF: spike = step_qif(n, q); break;
        case SNN_O_SIMPLE_LIF: spike = step_simple_lif(n, q); break;
        case SNN_O_ADAPTIVE_LIF: spike = step_adaptive(n, q, 0); break;
        case SNN_O_ADAPTIVE_EXP_LIF: spike = step_adaptive(n, q, 1); break;
        case SNN_O_LEAKY_IZHIKEVICH: spike = step_leaky_izhikevich(n, q); break;
        case SNN_O_CUSTOM: spike = step_custom(n, q); break;
        case SNN_O_BCM_IZHIKEVICH:
            /* activity bookkeeping first (previous step's spike flag), then the Izhikevich step */
            if (n->is_spiking[q]) n->bcm_num_spikes[q] += 1;
            bcm_window_update(&n->bcm_clock[q], n->bcm_window[q], n->dt[q], n->bcm_num_spikes[q], n->bcm_period[q],
                              &n->bcm_current_activity[q], &n->bcm_average_activity[q], !n->chemical);
            spike = step_izhikevich(n, q);
            break;
        default:        spike = step_izhikevich(n, q); break;
        }
        n->is_spiking[q] = spike;
        if (spike) n->last_firing_time[q] = (int32_t)n->clock;   /* mod.rs:964-966 / 2555-2557 */
    }
}

/* ---------- step 3: plasticity (deferred form) ---------- */

/*
 * LatticeNetwork::update_weights_from_neurons_{across,within}_lattices, neuron/mod.rs:2308-2417,
 * driven from :2573-2576; gating :2559-2562.  Every edge (p,q) incident to a gated spiking neuron
 * receives `w += delta(lft[p], lft[q])` with the plasticity of q's lattice -- once as an incoming
 * edge of q, once more as an outgoing edge of p when p is gated too (delta is 0 then: tp == tq).
 */
void snn_o_plasticity(snn_o_net *n) { snn_o_plasticity_cols(n, 0, n->n_neurons); }

/* the same, touching only weights whose postsynaptic column lies in [c0, c1) (one shard's columns) */
/* BCM::update_weight, plasticity/mod.rs:102-107 */
static inline float bcm_weight(const snn_o_net *n, uint32_t l, float w, float pre_activity, uint32_t post)
{
    float sliding_threshold = n->bcm_average_activity[post] / n->bcm_average_scalar[l];
    float activity_term = n->bcm_current_activity[post] * (n->bcm_current_activity[post] - sliding_threshold);
    float weight_decay = n->bcm_decay[l] * w;
    return w + (activity_term * pre_activity - weight_decay) * n->bcm_dt[l];
}

/* kind of the connection (presynaptic row p -> a neuron of lattice l): see snn_o_net::conn_kind */
static inline uint32_t conn_kind_of(const snn_o_net *n, uint32_t p, uint32_t l)
{
    if (!n->conn_kind) return 0;
    const uint32_t source = (p < n->n_neurons) ? n->lattice[p] : n->n_lattices + n->st_lattice[p - n->n_neurons];
    return n->conn_kind[(size_t)source * n->n_lattices + l];
}

/* lattice l is held by the network's reward_modulated_lattices map (see snn_oracle.h, rm_is_modulated) */
static inline int lattice_is_modulated(const snn_o_net *n, uint32_t l)
{
    return (n->rm_do_modulation && n->rm_do_modulation[l]) || (n->rm_is_modulated && n->rm_is_modulated[l]);
}

void snn_o_plasticity_cols(snn_o_net *n, uint32_t c0, uint32_t c1)
{
    const uint32_t nn = n->n_neurons;
    const uint32_t n_tot = nn + n->n_cells;
    /* the matrix may be a column window [w_col0, w_col0 + w_ld) (full-size teacher-forced checks); [c0, c1) lies in it */
    const size_t ld = n->w_ld ? n->w_ld : nn;
    const uint32_t col0 = n->w_col0;
    if (!n->do_plasticity) return;

    for (uint32_t j = 0; j < nn; ++j) {
        if (!(n->is_spiking[j] && n->do_plasticity[n->lattice[j]])) continue;
        if (lattice_is_modulated(n, n->lattice[j])) continue;          /* a RewardModulatedLattice has no STDP rule of its own */
        /* incoming edges of j */
        if (j >= c0 && j < c1) {
            uint32_t l = n->lattice[j];
            for (uint32_t p = 0; p < n_tot; ++p) {
                size_t i = (size_t)p * ld + (j - col0);
                if (!n->connections[i]) continue;
                if (conn_kind_of(n, p, l)) continue;         /* a connection of the reward-modulated network: snn_o_reward_cross */
                if (n->plasticity_kind && n->plasticity_kind[l]) {
                    float pre = (p < nn) ? n->bcm_current_activity[p] : n->st_bcm_current_activity[p - nn];
                    n->weights[i] = bcm_weight(n, l, n->weights[i], pre, j);
                    continue;
                }
                int32_t tp = (p < nn) ? n->last_firing_time[p] : n->st_last_firing_time[p - nn];
                n->weights[i] += snn_o_stdp_delta(tp, n->last_firing_time[j], n->stdp_a_plus[l],
                                                  n->stdp_a_minus[l], n->stdp_tau_plus[l],
                                                  n->stdp_tau_minus[l], n->stdp_dt[l]);
            }
        }
        /* outgoing edges of j */
        for (uint32_t r = c0; r < c1; ++r) {
            size_t i = (size_t)j * ld + (r - col0);
            if (!n->connections[i]) continue;
            uint32_t l = n->lattice[r];
            if (conn_kind_of(n, j, l)) continue;
            if (n->plasticity_kind && n->plasticity_kind[l]) {
                n->weights[i] = bcm_weight(n, l, n->weights[i], n->bcm_current_activity[j], r);
                continue;
            }
            n->weights[i] += snn_o_stdp_delta(n->last_firing_time[j], n->last_firing_time[r],
                                              n->stdp_a_plus[l], n->stdp_a_minus[l],
                                              n->stdp_tau_plus[l], n->stdp_tau_minus[l], n->stdp_dt[l]);
        }
    }
}

/* ---------- reward modulation ---------- */


/* RewardModulatedSTDP::update, plasticity/mod.rs:199-201, on every modulated lattice */
void snn_o_apply_reward(snn_o_net *n, float reward)
{
    if (!n->rm_do_modulation) return;
    for (uint32_t l = 0; l < n->n_lattices; ++l) {
        if (!lattice_is_modulated(n, l)) continue;                  /* (a paused modulator still takes the reward, :5287-5291) */
        n->rm_dopamine[l] = n->rm_dopamine[l] * snn_o_expf(-n->rm_dt[l] / n->rm_tau_d[l]) + n->rm_tau_d[l] * reward;
    }
}

/* RewardModulatedLattice::update_weights_from_neurons (neuron/mod.rs:3022-3054) with
 * RewardModulatedSTDP::update_weight (plasticity/mod.rs:203-237).  do_update is always true (:239-241), so every
 * step every internal edge (p,q) of a modulated lattice is visited exactly twice: as an outgoing edge of p and as
 * an incoming edge of q.  The reference does this inside its neuron loop (HashSet order), where the two visits may
 * see different last_firing_times; the deferred form used here (all neurons of the step updated first -- the same
 * choice as for STDP, and what the reference's network form does) makes both visits see the same delta:
 *   visit 1: dw = 0 + delta; counter 0 -> 1;                                  weight += c * dopamine
 *   visit 2: dw += delta; c = c * exp(-dt / tau_c) + tau_c * dw; counter, dw -> 0;  weight += c * dopamine */
void snn_o_reward_modulation(snn_o_net *n) { snn_o_reward_modulation_cols(n, 0, n->n_neurons); }

void snn_o_reward_modulation_cols(snn_o_net *n, uint32_t c0, uint32_t c1)
{
    const uint32_t nn = n->n_neurons;
    if (!n->rm_do_modulation || !n->traces) return;
    for (uint32_t p = 0; p < nn; ++p) {
        const uint32_t l = n->lattice[p];
        if (!n->rm_do_modulation[l]) continue;
        const float dopamine = n->rm_dopamine[l], dt = n->rm_dt[l], tau_c = n->rm_tau_c[l];
        const float decay = snn_o_expf(-dt / tau_c);
        for (uint32_t q = c0; q < c1; ++q) {
            size_t i = (size_t)p * nn + q;
            if (n->lattice[q] != l || !n->connections[i]) continue;
            float delta_w = snn_o_stdp_delta(n->last_firing_time[p], n->last_firing_time[q], n->rm_a_plus[l],
                                             n->rm_a_minus[l], n->rm_tau_plus[l], n->rm_tau_minus[l], dt);
            float w = n->weights[i], c = n->traces[i];
            float dw = 0.0f;
            dw += delta_w;
            w += c * dopamine;
            dw += delta_w;
            c = c * decay + tau_c * dw;
            w += c * dopamine;
            n->weights[i] = w;
            n->traces[i] = c;
        }
    }
}

/* RewardModulatedLatticeNetwork: the connections BETWEEN lattices that carry a RewardModulatedConnection (conn_kind 1 =
 * RewardModulatedWeight, 2 = Weight).  post_neuron_update_step (neuron/mod.rs:5030-5043) visits, in this order,
 *   1. every SPIKING neuron of a plain lattice with do_plasticity: update_weights_from_neurons_across_lattices (:4707-4802),
 *   2. EVERY neuron of a reward-modulated lattice (RewardModulatedSTDP::do_update is always true, plasticity/mod.rs:239-241):
 *      update_weights_from_neurons_across_reward_lattices (:4855-4977),
 * lattices in layout order here (the reference walks a HashMap: its order between lattices is unspecified), neurons by index.
 * A visit of neuron x handles first the connections p -> x (incoming half), then the connections x -> o (outgoing half).
 *
 * incoming p -> x:
 *   Weight, x plain (:4721-4741):      x's lattice's STDP adds its delta for (p, x);
 *   Weight, x modulated (:4869-4883):  p's lattice's STDP -- the PRESYNAPTIC lattice's -- when p sits in a plain lattice, else nothing;
 *   RewardModulatedWeight, x plain (:4742-4756):      the modulator of p's lattice visits the TraceRSTDP once;
 *   RewardModulatedWeight, x modulated (:4885-4921):  the modulator of x's lattice visits it once (p: any lattice or spike train).
 * outgoing x -> o: the reference looks up the REVERSE connection o -> x (`lookup_weight(&output_pos, pos)`, :4768-4771 and
 * :4929-4932), unwraps it, updates that COPY with (pre = x, post = o) and stores it as the connection x -> o:
 *   Weight, x plain (:4774-4787):      copy.weight + delta of x's lattice's STDP;
 *   Weight, x modulated (:4935-4950):  copy.weight + delta of o's lattice's STDP when o sits in a plain lattice, else untouched;
 *   RewardModulatedWeight, x plain (:4788-4801):      one visit by the modulator of o's lattice on the copy;
 *   RewardModulatedWeight, x modulated (:4952-4972):  one visit by the modulator of x's lattice on the copy;
 *   weight, trace, dw and counter of x -> o are all REPLACED by the copy's.
 * One visit (RewardModulatedSTDP::update_weight, plasticity/mod.rs:203-237): dw += delta; counter 0: counter = 1; counter 1: the
 * trace folds dw in, dw = 0, counter = 0; then weight += c * dopamine.  dw (`pending`) and the counter (`edge_counter`) are per
 * connection and live across steps.
 * The reference panics (unwrap of None) outside this domain; snn_o_reward_cross_check names the case and the stepper refuses it:
 *   1 a connection u -> v of a visited lattice without its reverse v -> u, or with a reverse of another kind (:4768-4771, :4929-4932);
 *   2 RewardModulatedWeight where the visit has no modulator: both lattices plain and one of them plastic (:4743, :4789),
 *     or from a spike train into a plastic plain lattice (:4743);
 *   3 Weight between a plastic plain lattice and a reward-modulated one (:4729-4733 takes the source for a spike train, :4778
 *     looks the target up among the plain lattices);
 *   4 a plain lattice with the BCM rule on such a connection (the reference's network is generic over ONE plasticity rule).
 * Connections of kind 0 are left to the plain network's rule (snn_o_plasticity_cols), which skips kinds 1 and 2. */
static inline int cross_visited(const snn_o_net *n, uint32_t l)
{
    return lattice_is_modulated(n, l) ? (n->rm_do_modulation[l] != 0) : (n->do_plasticity && n->do_plasticity[l]);
}

int snn_o_reward_cross_check(const snn_o_net *n)
{
    const uint32_t nn = n->n_neurons, n_tot = nn + n->n_cells, nl = n->n_lattices;
    if (!n->conn_kind || !n->rm_do_modulation) return 0;
    for (uint32_t p = 0; p < n_tot; ++p)
        for (uint32_t q = 0; q < nn; ++q) {
            if (!n->connections[(size_t)p * nn + q]) continue;
            const uint32_t lq = n->lattice[q], kind = conn_kind_of(n, p, lq);
            if (kind == 0) continue;
            const int mod_q = lattice_is_modulated(n, lq), plastic_q = !mod_q && cross_visited(n, lq);
            if (plastic_q && n->plasticity_kind && n->plasticity_kind[lq]) return 4;
            if (p >= nn) {                                           /* spike train -> q: incoming half of q only */
                if (kind == 1 && plastic_q) return 2;
                continue;
            }
            const uint32_t lp = n->lattice[p];
            if (lp == lq) continue;
            const int mod_p = lattice_is_modulated(n, lp), plastic_p = !mod_p && cross_visited(n, lp);
            if (plastic_p && n->plasticity_kind && n->plasticity_kind[lp]) return 4;
            if (kind == 1 && !mod_p && !mod_q && (plastic_p || plastic_q)) return 2;
            if (kind == 2 && ((plastic_p && mod_q) || (plastic_q && mod_p))) return 3;
            if (cross_visited(n, lp)) {                              /* p's outgoing half needs q -> p */
                if (!n->connections[(size_t)q * nn + p] || n->conn_kind[(size_t)lq * nl + lp] != kind) return 1;
            }
        }
    return 0;
}

/* one RewardModulatedSTDP::update_weight with the modulator of lattice m */
static void cross_trace_visit(const snn_o_net *n, uint32_t m, int32_t t_pre, int32_t t_post, float *w, float *c, float *dw, uint8_t *counter)
{
    const float dt = n->rm_dt[m], tau_c = n->rm_tau_c[m];
    *dw += snn_o_stdp_delta(t_pre, t_post, n->rm_a_plus[m], n->rm_a_minus[m], n->rm_tau_plus[m], n->rm_tau_minus[m], dt);
    if (*counter == 0) {
        *counter = 1;
    } else {
        *c = *c * snn_o_expf(-dt / tau_c) + tau_c * *dw;
        *counter = 0;
        *dw = 0.0f;
    }
    *w += *c * n->rm_dopamine[m];
}

static inline float cross_stdp(const snn_o_net *n, uint32_t l, int32_t t_pre, int32_t t_post)
{
    return snn_o_stdp_delta(t_pre, t_post, n->stdp_a_plus[l], n->stdp_a_minus[l], n->stdp_tau_plus[l], n->stdp_tau_minus[l], n->stdp_dt[l]);
}

static void cross_visit(snn_o_net *n, uint32_t x)
{
    const uint32_t nn = n->n_neurons, n_tot = nn + n->n_cells, lx = n->lattice[x];
    const int mod_x = lattice_is_modulated(n, lx);
    const int32_t tx = n->last_firing_time[x];
    for (uint32_t p = 0; p < n_tot; ++p) {                           /* incoming half */
        const size_t i = (size_t)p * nn + x;
        if (!n->connections[i]) continue;
        const uint32_t kind = conn_kind_of(n, p, lx);
        if (kind == 0 || (p < nn && n->lattice[p] == lx)) continue;
        const int32_t tp = (p < nn) ? n->last_firing_time[p] : n->st_last_firing_time[p - nn];
        if (kind == 2) {
            if (!mod_x) n->weights[i] += cross_stdp(n, lx, tp, tx);
            else if (p < nn && !lattice_is_modulated(n, n->lattice[p])) n->weights[i] += cross_stdp(n, n->lattice[p], tp, tx);
        } else {
            const uint32_t m = mod_x ? lx : n->lattice[p];
            cross_trace_visit(n, m, tp, tx, &n->weights[i], &n->traces[i], &n->pending[i], &n->edge_counter[i]);
        }
    }
    for (uint32_t o = 0; o < nn; ++o) {                              /* outgoing half */
        const size_t f = (size_t)x * nn + o, r = (size_t)o * nn + x;
        const uint32_t lo = n->lattice[o];
        if (!n->connections[f] || lo == lx) continue;
        const uint32_t kind = conn_kind_of(n, x, lo);
        if (kind == 0 || !n->connections[r]) continue;               /* (no reverse: outside the domain, see the check) */
        const int32_t to = n->last_firing_time[o];
        if (kind == 2) {
            if (!mod_x) n->weights[f] = n->weights[r] + cross_stdp(n, lx, tx, to);
            else if (!lattice_is_modulated(n, lo)) n->weights[f] = n->weights[r] + cross_stdp(n, lo, tx, to);
        } else {
            float w = n->weights[r], c = n->traces[r], dw = n->pending[r];
            uint8_t counter = n->edge_counter[r];
            cross_trace_visit(n, mod_x ? lx : lo, tx, to, &w, &c, &dw, &counter);
            n->weights[f] = w; n->traces[f] = c; n->pending[f] = dw; n->edge_counter[f] = counter;
        }
    }
}

void snn_o_reward_cross(snn_o_net *n)
{
    const uint32_t nn = n->n_neurons;
    if (!n->conn_kind || !n->rm_do_modulation) return;
    {
        uint32_t any = 0;       /* no connection of these kinds: a plain network, nothing is visited */
        for (size_t i = 0; i < (size_t)(n->n_lattices + n->n_st_lattices) * n->n_lattices; ++i) any |= n->conn_kind[i];
        if (!any) return;
    }
    for (uint32_t x = 0; x < nn; ++x) {
        const uint32_t l = n->lattice[x];
        if (!lattice_is_modulated(n, l) && n->do_plasticity && n->do_plasticity[l] && n->is_spiking[x]) cross_visit(n, x);
    }
    for (uint32_t x = 0; x < nn; ++x)
        if (n->rm_do_modulation[n->lattice[x]]) cross_visit(n, x);      /* (modulated AND do_modulation, :5113) */
}

/* ---------- step 6: spike trains ---------- */

/* SpikeTrainLattice::iterate neuron/mod.rs:1377-1393; PoissonNeuron (GPU form) spike_train/mod.rs:411-435;
 * RateSpikeTrain::iterate spike_train/mod.rs:1016-1031 */
void snn_o_spike_trains(snn_o_net *n)
{
    for (uint32_t s = 0; s < n->n_cells; ++s) {
        uint32_t spike;
        float custom_v = 0.0f;
        if (n->st_kind == SNN_O_ST_CUSTOM) {
            /* generated SpikeTrain::iterate, build_test/nb_macro/src/lib.rs:4884-4891 */
            float slot[5 + 16];
            slot[0] = n->st_current_voltage[s]; slot[1] = n->st_is_spiking[s] ? 1.0f : 0.0f; slot[2] = n->st_dt[s];
            slot[3] = n->st_v_resting[s]; slot[4] = n->st_v_th[s];
            for (uint32_t k = 0; k < n->st_custom_nvars; ++k) slot[5 + k] = n->st_custom_vars[(size_t)k * n->n_cells + s];
            program_run(n->st_custom_code, n->st_custom_consts, 0, slot, 1);
            for (uint32_t k = 0; k < n->st_custom_nvars; ++k) n->st_custom_vars[(size_t)k * n->n_cells + s] = slot[5 + k];
            spike = slot[1] != 0.0f;
            custom_v = slot[0];
        } else if (n->st_kind == SNN_O_ST_POISSON || n->st_kind == SNN_O_ST_BCM_POISSON) {
            uint32_t new_seed = snn_o_xorshift32(n->st_seed[s]);
            n->st_seed[s] = new_seed;
            float random_number = (float)new_seed / 4294967296.0f;   /* (float)seed / 0xFFFFFFFF */
            spike = random_number < n->st_chance_of_firing[s];
        } else if (n->st_kind == SNN_O_ST_PRESET) {
            /* PresetSpikeTrain::iterate spike_train/mod.rs:803-827; an empty list never fires */
            float clock = n->st_step[s] + n->st_dt[s];
            uint32_t f0 = n->st_firing_ptr[s], len = n->st_firing_ptr[s + 1] - f0;
            spike = len != 0 && clock > n->st_firing_times[f0 + n->st_counter[s]];
            if (spike) {
                clock = 0.0f;
                n->st_counter[s] += 1;
                if (n->st_counter[s] == len) n->st_counter[s] = 0;
            }
            n->st_step[s] = clock;
        } else {
            float step = n->st_step[s] + n->st_dt[s];
            spike = (n->st_rate[s] != 0.0f) && (step >= n->st_rate[s]);
            if (spike) step = 0.0f;
            n->st_step[s] = step;
        }
        float v = (n->st_kind == SNN_O_ST_CUSTOM) ? custom_v : (spike ? n->st_v_th[s] : n->st_v_resting[s]);
        if (n->st_kind == SNN_O_ST_BCM_POISSON) {
            /* BCMPoissonNeuron::iterate spike_train/mod.rs:931-954: activity = voltage change, replaced by the firing
             * rate at the end of a window */
            n->st_bcm_current_activity[s] = v - n->st_current_voltage[s];
            if (spike) n->st_bcm_num_spikes[s] += 1;
            bcm_window_update(&n->st_bcm_clock[s], n->st_bcm_window[s], n->st_dt[s], n->st_bcm_num_spikes[s],
                              n->st_bcm_period[s], &n->st_bcm_current_activity[s], &n->st_bcm_average_activity[s], 1);
        }
        n->st_current_voltage[s] = v;
        n->st_is_spiking[s] = spike;
        if (n->st_nt_flags) {
            for (int k = 0; k < SNN_O_K; ++k) {
                size_t i = (size_t)s * SNN_O_K + k;
                if (!n->st_nt_flags[i]) continue;
                if (n->nt_kind == SNN_O_NT_CUSTOM) {
                    n->st_nt_t[i] = custom_nt_apply(n, n->st_nt_custom_vars, (size_t)n->n_cells * SNN_O_K, i, n->st_nt_t[i],
                                                    v, spike, n->st_dt[s]);
                    continue;
                }
                n->st_nt_t[i] = nt_apply(n->nt_kind, n->st_nt_t[i], n->st_nt_t_max[i],
                                         n->st_nt_clearance ? n->st_nt_clearance[i] : 0.0f,
                                         n->st_nt_v_p ? n->st_nt_v_p[i] : 0.0f,
                                         n->st_nt_k_p ? n->st_nt_k_p[i] : 1.0f,
                                         v, spike, n->st_dt[s]);
            }
        }
        if (spike) n->st_last_firing_time[s] = (int32_t)n->st_clock[n->st_lattice[s]];
    }
    for (uint32_t l = 0; l < n->n_st_lattices; ++l) n->st_clock[l] += 1;
}

/* ---------- reduced histories ---------- */

/* AverageVoltageHistory::update neuron/mod.rs:311-317: sum / length; EEGHistory::update :262-277:
 * (1 / (4 * PI * conductivity * distance)) * sum(V - reference_voltage).  Chunked canonical order. */
static void lattice_summaries(snn_o_net *n, uint64_t it)
{
    const float pi = 3.14159274101257324f;       /* std::f32::consts::PI */
    for (uint32_t l = 0; l < n->n_lattices; ++l) {
        const uint32_t first = n->lattice_first[l], count = n->lattice_count[l];
        float tot = 0.0f, tot_e = 0.0f;
        for (uint32_t c0 = 0; c0 < count; c0 += SNN_O_CHUNK) {
            uint32_t c1 = c0 + SNN_O_CHUNK;
            if (c1 > count) c1 = count;
            float part = 0.0f, part_e = 0.0f;
            for (uint32_t i = c0; i < c1; ++i) {
                const float v = n->current_voltage[first + i];
                part += v;
                part_e += v - n->eeg_reference_voltage;
            }
            tot += part;
            tot_e += part_e;
        }
        if (n->avg_history) n->avg_history[it * n->n_lattices + l] = tot / (float)count;
        if (n->eeg_history)
            n->eeg_history[it * n->n_lattices + l] =
                (1.0f / (4.0f * pi * n->eeg_conductivity * n->eeg_distance)) * tot_e;
    }
}

/* ---------- whole loop ---------- */

/* run_lattice_* neuron/mod.rs:1035-1088, run_lattices_* :2598-2651; (false,false) is a no-op :1217 */
void snn_o_run(snn_o_net *n, uint64_t iterations)
{
    if (!n->electrical && !n->chemical) return;
    for (uint64_t it = 0; it < iterations; ++it) {
        if (n->rewards) snn_o_apply_reward(n, n->rewards[it]);
        if (n->n_neurons) {
            snn_o_inputs(n);
            snn_o_update_neurons(n);
            snn_o_plasticity(n);
            snn_o_reward_modulation(n);
            snn_o_reward_cross(n);
            if (n->voltage_history)
                memcpy(n->voltage_history + (size_t)it * n->n_neurons, n->current_voltage,
                       sizeof(float) * n->n_neurons);
            if (n->spike_history)
                for (uint32_t q = 0; q < n->n_neurons; ++q)
                    n->spike_history[(size_t)it * n->n_neurons + q] = (uint8_t)n->is_spiking[q];
            if (n->spike_counts)
                for (uint32_t q = 0; q < n->n_neurons; ++q) n->spike_counts[q] += n->is_spiking[q] ? 1u : 0u;
            if (n->avg_history || n->eeg_history) lattice_summaries(n, it);
        }
        n->clock += 1;
        if (n->n_cells) {
            snn_o_spike_trains(n);
            if (n->st_voltage_history)
                memcpy(n->st_voltage_history + (size_t)it * n->n_cells, n->st_current_voltage,
                       sizeof(float) * n->n_cells);
        }
    }
}

/* the same loop over a sparse graph (snn_o_inputs_csr); no plasticity, histories optional */
void snn_o_run_csr(snn_o_net *n, const uint64_t *row_ptr, const uint32_t *pre, const float *w, uint64_t iterations)
{
    if (!n->electrical && !n->chemical) return;
    for (uint64_t it = 0; it < iterations; ++it) {
        if (n->n_neurons) {
            snn_o_inputs_csr(n, row_ptr, pre, w, 0, n->n_neurons);
            snn_o_update_neurons(n);
            if (n->voltage_history)
                memcpy(n->voltage_history + (size_t)it * n->n_neurons, n->current_voltage, sizeof(float) * n->n_neurons);
            if (n->spike_history)
                for (uint32_t q = 0; q < n->n_neurons; ++q)
                    n->spike_history[(size_t)it * n->n_neurons + q] = (uint8_t)n->is_spiking[q];
        }
        n->clock += 1;
        if (n->n_cells) snn_o_spike_trains(n);
    }
}

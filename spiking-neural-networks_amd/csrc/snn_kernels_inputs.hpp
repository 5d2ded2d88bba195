// k_inputs_dense -- the dominant kernel: ONE streaming pass over the dense synapse matrix per
// time-step that produces, for every local postsynaptic neuron, the chunk partials of
//   * the gap-junction input   Lattice::calculate_internal_electrical_input_from_positions
//                              (neuron/mod.rs:702-730) / LatticeNetwork::calculate_electrical_input_
//                              from_positions (:2115-2167), gap_junction (:54-60),
//                              spike_train_gap_junction (:119-137)
//   * the neurotransmitter input per type (neuron/mod.rs:733-754, 2169-2210;
//                              iterate_and_spike/mod.rs:2837-2866)
// It supersedes the reference's four OpenCL kernels calculate_internal_electrical_inputs,
// get_neurotransmitter_inputs, calculate_network_electrical_inputs and
// calculate_network_chemical_inputs (neuron/gpu_lattices/mod.rs:60-137, 1244-1382), which each make
// their own pass over connections u32[N^2] + weights f32[N^2] with one work-item per column.
//
// Mapping (wave64, gfx950): workgroup = 256 threads = 4 waves; a workgroup owns CHUNK = 256
// consecutive presynaptic rows x 1024 consecutive postsynaptic columns; every lane owns 4 adjacent
// columns, so one wave-row is ONE 1 KiB global_load_dwordx4 and a workgroup-row is 4 KiB contiguous.
// blockIdx.x = column tile (fastest in dispatch order), blockIdx.y = row chunk: the workgroups in
// flight at any moment sweep whole matrix rows, i.e. long contiguous HBM bursts.  The presynaptic
// values of the chunk (voltages, spike-train values, neurotransmitter concentrations and flags) are
// staged once in LDS and read back as wave-uniform broadcasts; the per-lane postsynaptic voltage and
// gap conductance stay in registers.  No reuse of W exists (0.5 flop/byte), so there is nothing for
// MFMA here: the kernel is priced against HBM bandwidth.
//
// Arithmetic contract (bit-exact with the oracle): inside a chunk each column's sum is a strictly
// ascending sequential f32 accumulation from 0.0f of `term * weight` (no FMA: the TU is built with
// -ffp-contract=off); absent edges (NaN sentinel) are skipped, not added as zeros.
#pragma once
#include "snn_layout.hpp"

namespace snn {

struct InputsArgs {
    const float *W;
    uint32_t ld;            // floats per matrix row (multiple of 64)
    uint32_t n_loc;         // local postsynaptic columns
    uint32_t q0;            // global neuron index of local column 0
    uint32_t n_neurons;
    uint32_t n_tot;
    const float *xbuf;
    XLayout xl;
    const float *gap_conductance;
    // spike-train cells
    const float *st_value;
    const int32_t *st_last_firing_time;
    const float *st_nt_t;          // [3][c_pad]
    const uint32_t *st_nt_flags;   // [3][c_pad]
    uint32_t c_pad;
    // neuron neurotransmitter flags [3][n_pad]
    const uint32_t *nt_flags;
    uint32_t n_pad;
    // outputs
    float *part_i;          // [n_chunks][ld]
    float *part_t;          // [3][n_chunks][ld]
    uint32_t n_chunks;
};

// kind word per staged presynaptic row: bits 0..1 = 0 neuron | 1 spike train that never fired |
// 2 spike train that fired; bits 8..10 = carries neurotransmitter type k
constexpr uint32_t KIND_NEURON = 0, KIND_ST_SILENT = 1, KIND_ST_FIRED = 2;

typedef float v4f __attribute__((ext_vector_type(4)));

// W is read exactly once per step: stream it past the caches (global_load_dwordx4 ... nt).
// Measured on MI355X at 256x256 (same box, A/B): nt 6.18 TB/s vs default cache policy 5.75 TB/s.
__device__ __forceinline__ v4f load_w4(const v4f *p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float acc_if_edge(float acc, float term, float w)
{
    // `w == w` is false exactly for the NaN sentinel of an absent edge
    return (w == w) ? acc + term * w : acc;
}

template <bool ELEC, bool CHEM>
__global__ __launch_bounds__(256) void k_inputs_dense(const InputsArgs a)
{
    __shared__ float s_val[CHUNK];
    __shared__ uint32_t s_kind[CHUNK];
    __shared__ float s_t[CHEM ? K_TYPES : 1][CHUNK];

    const uint32_t chunk = blockIdx.y;
    const uint32_t p0 = chunk * CHUNK;
    const uint32_t rows = min((uint32_t)CHUNK, a.n_tot - p0);
    const uint32_t tid = threadIdx.x;

    // ---- stage the chunk's presynaptic values in LDS (one coalesced read per array) ----
    if (tid < rows) {
        const uint32_t p = p0 + tid;
        float val;
        uint32_t kind;
        if (p < a.n_neurons) {
            val = a.xbuf[a.xl.at(p, PLANE_V)];
            kind = KIND_NEURON;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    const uint32_t f = a.nt_flags[(size_t)k * a.n_pad + p];
                    kind |= f ? (0x100u << k) : 0u;
                    s_t[k][tid] = a.xbuf[a.xl.at(p, PLANE_T0 + k)];
                }
            }
        } else {
            const uint32_t s = p - a.n_neurons;
            val = a.st_value[s];
            kind = (a.st_last_firing_time[s] < 0) ? KIND_ST_SILENT : KIND_ST_FIRED;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    const uint32_t f = a.st_nt_flags[(size_t)k * a.c_pad + s];
                    kind |= f ? (0x100u << k) : 0u;
                    s_t[k][tid] = a.st_nt_t[(size_t)k * a.c_pad + s];
                }
            }
        }
        s_val[tid] = val;
        s_kind[tid] = kind;
    }
    __syncthreads();

    const uint32_t ql = blockIdx.x * TILE_POSTS + tid * 4;   // first of this lane's 4 local columns
    if (ql >= a.ld) return;

    // ---- this lane's postsynaptic voltage / conductance, kept in registers ----
    float vq[4], gq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = ql + j;
        if (ELEC && q < a.n_loc) {
            vq[j] = a.xbuf[a.xl.at(a.q0 + q, PLANE_V)];
            gq[j] = a.gap_conductance[a.q0 + q];
        } else {
            vq[j] = 0.0f;
            gq[j] = 0.0f;
        }
    }

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float tacc[CHEM ? K_TYPES : 1][4];
#pragma unroll
    for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) tacc[k][j] = 0.0f;

    const v4f *wrow = reinterpret_cast<const v4f *>(a.W + (size_t)p0 * a.ld + ql);
    const size_t ld4 = a.ld / 4;

    // Rows are consumed in batches of ROW_BATCH: all loads of a batch are issued before the first use,
    // so every wave keeps ROW_BATCH KiB of HBM reads in flight regardless of the branches in the body.
    constexpr uint32_t ROW_BATCH = 8;

    const bool plain = !CHEM && (p0 + rows <= a.n_neurons);   // workgroup-uniform
    if (plain) {
        // all presynaptic rows are neurons, electrical only: the C1/C2 inner loop
        auto body = [&](uint32_t r, const v4f w) {
            const float vp = s_val[r];
            acc[0] = acc_if_edge(acc[0], gq[0] * (vp - vq[0]), w.x);
            acc[1] = acc_if_edge(acc[1], gq[1] * (vp - vq[1]), w.y);
            acc[2] = acc_if_edge(acc[2], gq[2] * (vp - vq[2]), w.z);
            acc[3] = acc_if_edge(acc[3], gq[3] * (vp - vq[3]), w.w);
        };
        uint32_t r = 0;
        for (; r + ROW_BATCH <= rows; r += ROW_BATCH) {
            v4f wb[ROW_BATCH];
#pragma unroll
            for (uint32_t u = 0; u < ROW_BATCH; ++u) wb[u] = load_w4(wrow + (size_t)(r + u) * ld4);
#pragma unroll
            for (uint32_t u = 0; u < ROW_BATCH; ++u) body(r + u, wb[u]);
        }
        for (; r < rows; ++r) body(r, load_w4(wrow + (size_t)r * ld4));
    } else {
        auto body = [&](uint32_t r, const v4f w4) {
            const float w[4] = {w4.x, w4.y, w4.z, w4.w};
            const uint32_t kind = __builtin_amdgcn_readfirstlane(s_kind[r]);
            if (ELEC) {
                const float vp = s_val[r];
                const uint32_t src = kind & 3u;
                if (src == KIND_NEURON) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * (vp - vq[j]), w[j]);
                } else if (src == KIND_ST_SILENT) {
                    // never fired: v_resting without the conductance factor (neuron/mod.rs:126-128)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = acc_if_edge(acc[j], vp, w[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * vp, w[j]);
                }
            }
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    if (kind & (0x100u << k)) {
                        const float t = s_t[k][r];
#pragma unroll
                        for (int j = 0; j < 4; ++j) tacc[k][j] = acc_if_edge(tacc[k][j], t, w[j]);
                    }
                }
            }
        };
        uint32_t r = 0;
        for (; r + ROW_BATCH <= rows; r += ROW_BATCH) {
            v4f wb[ROW_BATCH];
#pragma unroll
            for (uint32_t u = 0; u < ROW_BATCH; ++u) wb[u] = load_w4(wrow + (size_t)(r + u) * ld4);
#pragma unroll
            for (uint32_t u = 0; u < ROW_BATCH; ++u) body(r + u, wb[u]);
        }
        for (; r < rows; ++r) body(r, load_w4(wrow + (size_t)r * ld4));
    }

    if (ELEC) {
        float4 *dst = reinterpret_cast<float4 *>(a.part_i + (size_t)chunk * a.ld + ql);
        *dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) {
            float4 *dst = reinterpret_cast<float4 *>(a.part_t + ((size_t)k * a.n_chunks + chunk) * a.ld + ql);
            *dst = make_float4(tacc[k][0], tacc[k][1], tacc[k][2], tacc[k][3]);
        }
    }
}

// Static per-column counts, recomputed when the graph or the neurotransmitter flags change:
//   avg[q]        = max(1, #{p : edge (p,q)})                    neuron/mod.rs:722-727
//   tcount[k][q]  = #{p : edge (p,q) and p carries type k}       iterate_and_spike/mod.rs:2847-2853
// One thread per local column, rows strided over blockIdx.y, integer atomics (order independent).
struct CountArgs {
    const float *W;
    uint32_t ld, n_loc, n_neurons, n_tot;
    const uint32_t *nt_flags; uint32_t n_pad;
    const uint32_t *st_nt_flags; uint32_t c_pad;
    uint32_t *n_in;        // [ld]
    uint32_t *tcount;      // [3][ld]
    uint32_t rows_per_block;
};

__global__ __launch_bounds__(256) void k_graph_count(const CountArgs a)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r0 = blockIdx.y * a.rows_per_block;
    const uint32_t r1 = min(a.n_tot, r0 + a.rows_per_block);
    if (q >= a.n_loc) return;
    uint32_t cnt = 0, tc[K_TYPES] = {0, 0, 0};
    for (uint32_t p = r0; p < r1; ++p) {
        const float w = a.W[(size_t)p * a.ld + q];
        if (w == w) {
            ++cnt;
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                const uint32_t f = (p < a.n_neurons) ? a.nt_flags[(size_t)k * a.n_pad + p]
                                                     : a.st_nt_flags[(size_t)k * a.c_pad + (p - a.n_neurons)];
                tc[k] += f ? 1u : 0u;
            }
        }
    }
    if (cnt) atomicAdd(&a.n_in[q], cnt);
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k)
        if (tc[k]) atomicAdd(&a.tcount[(size_t)k * a.ld + q], tc[k]);
}

} // namespace snn

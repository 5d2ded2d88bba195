// Sparse (CSR-by-postsynaptic-neuron) form of the hot path, for connectivity that cannot be held as a dense
// N x N matrix (BASELINE configs[4]: 4 x 512^2 neurons + Poisson cells would need 4.4 TB dense).  The
// reference has no sparse GPU form (its AdjacencyList, graph/mod.rs:974-1118, is CPU only); the arithmetic is
// the same canonical chunked ascending sum as the dense kernel -- a row's entries are sorted by presynaptic
// index and a partial is flushed whenever the 256-index chunk changes -- so dense and CSR handles of the same
// graph produce bit-identical results.
//
// Layout (local postsynaptic rows only): ptr[n_loc+1] u32, pre[nnz] u32, w[nnz] f32, post[nnz] u32 (local
// row of every edge) and the transpose index t_ptr[n_tot+1], t_edge[nnz] (edges grouped by presynaptic cell,
// used by the outgoing side of STDP).  8 B per synapse are streamed per step (index + weight); the
// presynaptic state is gathered from the exchanged planes, which stay L2-resident (20 B per neuron).
#pragma once
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_misc.hpp"

namespace snn {

struct CsrGraph {
    const uint32_t *ptr;
    const uint32_t *pre;
    float *w;
    const uint32_t *post;
    const uint32_t *t_ptr;
    const uint32_t *t_edge;
    uint32_t n_loc;
};

struct CsrInputsArgs {
    CsrGraph g;
    InputsArgs in;      // presynaptic state pointers / sizes; W, ld unused except ld = partial row stride
};

// Workgroup = 256 consecutive rows, whose edges are one contiguous segment of the CSR arrays.
// Phase 1 (edge-parallel): the segment is read with coalesced, independent loads (index, weight) plus ONE
// gather per edge of the presynaptic value (a neuron's voltage or a spike-train cell's gap-junction value,
// selected by pointer, not by branch); value, weight and a meta word (chunk id, source kind, transmitter
// flags) go to LDS.  Phase 2 (row-parallel): each thread, holding its own voltage and conductance in
// registers, forms `term * weight` for its row's edges from LDS in ascending order with the canonical chunk
// flush.  The arithmetic per edge and the order of the adds are exactly those of the dense kernel; only the
// memory access pattern differs.  Segments that do not fit the LDS tile (very long rows) are read straight
// from global memory by the row's thread.
template <bool ELEC, bool CHEM>
__global__ __launch_bounds__(256) void k_inputs_csr(const CsrInputsArgs a)
{
    constexpr uint32_t CAP = CHEM ? 2048 : 4096;            // edges per workgroup tile: 48 KiB of LDS
    constexpr uint32_t EDGE_BATCH = CAP / 256;              // independent edges per thread: a full tile in one round trip
    constexpr uint32_t META_CELL = 1u << 28, META_SILENT = 1u << 29;
    __shared__ float s_v[CAP];                              // presynaptic value
    __shared__ float s_w[CAP];
    __shared__ uint32_t s_meta[CAP];                        // chunk id | type flags << 24 | kind bits
    __shared__ float s_t[CHEM ? K_TYPES : 1][CHEM ? CAP : 1];

    const InputsArgs &in = a.in;
    const uint32_t q_first = blockIdx.x * 256;
    const uint32_t q_last = min(a.g.n_loc, q_first + 256);
    const uint32_t q = q_first + threadIdx.x;
    const uint32_t e_base = a.g.ptr[q_first], e_end = a.g.ptr[q_last];
    const uint32_t len = e_end - e_base;
    const bool tiled = len <= CAP;                          // workgroup-uniform

    // Branch-free on purpose: every load is unconditional (the inapplicable source's index is clamped to
    // 0), so the EDGE_BATCH edges a thread handles per round trip have all their loads in flight together.
    auto fetch = [&](uint32_t e, float &v, float &w, float (&t)[K_TYPES], uint32_t &meta) {
        const uint32_t p = a.g.pre[e];
        w = a.g.w[e];
        const bool is_cell = p >= in.n_neurons;
        const uint32_t pn = is_cell ? 0u : p;                    // neuron index (clamped)
        const uint32_t s = is_cell ? p - in.n_neurons : 0u;      // spike-train cell index (clamped)
        meta = (p / CHUNK) | (is_cell ? META_CELL : 0u);
        v = 0.0f;
        if (ELEC) {
            const float *src = is_cell ? in.st_value + s : in.xbuf + in.xl.at(pn, PLANE_V);
            v = *src;
            const int32_t slft = in.st_last_firing_time[s];
            meta |= (is_cell && slft < 0) ? META_SILENT : 0u;
        }
        if (CHEM) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                const uint32_t *fsrc = is_cell ? in.st_nt_flags + (size_t)k * in.c_pad + s
                                               : in.nt_flags + (size_t)k * in.n_pad + pn;
                const float *tsrc = is_cell ? in.st_nt_t + (size_t)k * in.c_pad + s
                                            : in.xbuf + in.xl.at(pn, PLANE_T0 + k);
                const bool f = *fsrc != 0;
                meta |= f ? (0x1000000u << k) : 0u;
                t[k] = *tsrc;
            }
        }
    };

    if (tiled) {
        for (uint32_t i0 = threadIdx.x; i0 < len; i0 += 256 * EDGE_BATCH) {
            float v[EDGE_BATCH], w[EDGE_BATCH], t[EDGE_BATCH][K_TYPES];
            uint32_t meta[EDGE_BATCH];
#pragma unroll
            for (uint32_t u = 0; u < EDGE_BATCH; ++u)              // clamped: out-of-range results are dropped
                fetch(e_base + min(i0 + u * 256, len - 1), v[u], w[u], t[u], meta[u]);
#pragma unroll
            for (uint32_t u = 0; u < EDGE_BATCH; ++u) {
                const uint32_t i = i0 + u * 256;
                if (i < len) {
                    s_v[i] = v[u];
                    s_w[i] = w[u];
                    s_meta[i] = meta[u];
                    if (CHEM) {
#pragma unroll
                        for (int k = 0; k < K_TYPES; ++k) s_t[k][i] = t[u][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (q >= a.g.n_loc) return;

    const uint32_t e0 = a.g.ptr[q], e1 = a.g.ptr[q + 1];
    const float vq = ELEC ? in.xbuf[in.xl.at(in.q0 + q, PLANE_V)] : 0.0f;
    const float gq = ELEC ? in.gap_conductance[in.q0 + q] : 0.0f;
    float sum = 0.0f, part = 0.0f;
    float tsum[K_TYPES] = {0.0f, 0.0f, 0.0f}, tpart[K_TYPES] = {0.0f, 0.0f, 0.0f};
    uint32_t cur_chunk = 0xFFFFFFFFu;
    for (uint32_t e = e0; e < e1; ++e) {
        float v, w, t[K_TYPES];
        uint32_t meta;
        if (tiled) {
            const uint32_t i = e - e_base;
            v = s_v[i];
            w = s_w[i];
            meta = s_meta[i];
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) t[k] = s_t[k][i];
            }
        } else {
            fetch(e, v, w, t, meta);
        }
        const uint32_t chunk = meta & 0xFFFFFFu;
        if (chunk != cur_chunk) {        // flush the finished chunk's partial (canonical two-level order)
            if (cur_chunk != 0xFFFFFFFFu) {
                sum += part;
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) tsum[k] += tpart[k];
            }
            part = 0.0f;
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) tpart[k] = 0.0f;
            cur_chunk = chunk;
        }
        if (ELEC) {
            // gap_junction neuron/mod.rs:54-60; spike_train_gap_junction :119-137 (never fired: v_resting
            // without the conductance factor)
            const float term = (meta & META_CELL) ? ((meta & META_SILENT) ? v : gq * v) : gq * (v - vq);
            part += term * w;
        }
        if (CHEM) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k)
                if (meta & (0x1000000u << k)) tpart[k] += t[k] * w;
        }
    }
    if (cur_chunk != 0xFFFFFFFFu) {
        sum += part;
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) tsum[k] += tpart[k];
    }
    // Chunks without edges contribute +0.0f partials in the dense form; x + 0.0f == x for every x this sum
    // can hold (it starts at +0.0f, so it is never -0.0f): skipping them is exact.
    if (ELEC) in.part_i[q] = sum;
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) in.part_t[(size_t)k * in.ld + q] = tsum[k];
    }
}

// static counts of a CSR graph: n_in = row length, tcount[k] = entries whose presynaptic cell carries type k
struct CsrCountArgs {
    CsrGraph g;
    uint32_t n_neurons, ld;
    const uint32_t *nt_flags; uint32_t n_pad;
    const uint32_t *st_nt_flags; uint32_t c_pad;
    uint32_t *n_in, *tcount;
};

__global__ __launch_bounds__(256) void k_csr_count(const CsrCountArgs a)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.g.n_loc) return;
    const uint32_t e0 = a.g.ptr[q], e1 = a.g.ptr[q + 1];
    uint32_t tc[K_TYPES] = {0, 0, 0};
    for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t p = a.g.pre[e];
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) {
            const uint32_t f = (p < a.n_neurons) ? a.nt_flags[(size_t)k * a.n_pad + p]
                                                 : a.st_nt_flags[(size_t)k * a.c_pad + (p - a.n_neurons)];
            tc[k] += f ? 1u : 0u;
        }
    }
    a.n_in[q] = e1 - e0;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) a.tcount[(size_t)k * a.ld + q] = tc[k];
}

// STDP on CSR: incoming edges of a listed local neuron are its row; outgoing edges of a listed neuron
// are the transpose list of that presynaptic index.
struct CsrStdpArgs {
    CsrGraph g;
    StdpArgs s;
};

__global__ __launch_bounds__(64) void k_stdp_csr_in(const CsrStdpArgs a)
{
    const uint32_t count = *a.s.spike_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t j = a.s.spike_list[i];
        if (j < a.s.q0 || j >= a.s.q0 + a.s.n_loc) continue;
        const uint32_t r = j - a.s.q0;
        const float *prm = a.s.stdp + 5 * a.s.lattice_slot[j];
        const int32_t tj = a.s.last_firing_time[j];
        for (uint32_t e = a.g.ptr[r] + threadIdx.x; e < a.g.ptr[r + 1]; e += 64) {
            const uint32_t p = a.g.pre[e];
            const int32_t tp = (p < a.s.n_neurons) ? a.s.last_firing_time[p] : a.s.st_last_firing_time[p - a.s.n_neurons];
            a.g.w[e] = a.g.w[e] + stdp_delta(tp, tj, prm[0], prm[1], prm[2], prm[3], prm[4]);
        }
    }
}

__global__ __launch_bounds__(64) void k_stdp_csr_out(const CsrStdpArgs a)
{
    const uint32_t count = *a.s.spike_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t j = a.s.spike_list[i];
        const int32_t tj = a.s.last_firing_time[j];
        for (uint32_t t = a.g.t_ptr[j] + threadIdx.x; t < a.g.t_ptr[j + 1]; t += 64) {
            const uint32_t e = a.g.t_edge[t];
            const uint32_t gr = a.s.q0 + a.g.post[e];
            const float *prm = a.s.stdp + 5 * a.s.lattice_slot[gr];
            a.g.w[e] = a.g.w[e] + stdp_delta(tj, a.s.last_firing_time[gr], prm[0], prm[1], prm[2], prm[3], prm[4]);
        }
    }
}

} // namespace snn

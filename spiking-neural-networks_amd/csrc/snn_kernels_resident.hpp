// k_step_resident -- the whole neuron step (synaptic inputs + neuron update) of a SMALL dense lattice in one
// launch.  At the sizes the reference's users run on a CPU (5x5 .. 64x64) the two-kernel step is bound by
// latency, not by HBM: measured on MI355X at 32x32, k_inputs_dense 8.4 us (8 dependent L2 round trips per lane for
// the 256 sequential rows of a chunk) + k_update 3.7 us = the 12.7 us step.  Here
//   * a workgroup owns 64 postsynaptic columns and ALL their chunks: wavefront c sums chunk c (lane = column), so
//     the chunk partials meet in LDS and the neuron update of the 64 neurons follows in the same launch -- no
//     second kernel, and no device-scope fence (a ticket scheme across workgroups was tried first: the L2
//     write-back / invalidate that cross-XCD visibility needs on this chip costs more than the second launch);
//   * a lane keeps TWO batches of 32 rows in registers, so the next batch is in flight while the current one is
//     summed (the sum itself stays strictly sequential: canonical order); the presynaptic state of a 64-row block
//     is loaded one row per lane and broadcast with v_readlane -- no LDS staging, no barrier inside the sum;
//   * other workgroups may still be reading S(t) while a finished tile writes S(t+1): the exchanged state is read
//     from a shadow copy and written to the handle's exchange buffer AND the other shadow (UpdateArgs xout /
//     xout2); the host flips the shadows every step.
// Arithmetic and update code are those of k_inputs_dense / k_update: results are bit-identical.
#pragma once
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_update.hpp"

namespace snn {

// One wavefront per chunk: 256 threads = one wave per SIMD, so the 2 x 64 rows a lane keeps in registers fit without
// scratch.  A 16-wavefront variant (1024 threads, 2 x 32 rows per lane) was measured and dropped: 18.0 us per step
// at 48x48 against 14.4 us for the two kernels, no difference at 64x64.
constexpr uint32_t RESIDENT_MAX_CHUNKS = 4;

struct ResidentArgs {
    InputsArgs in;              // in.xbuf = the shadow holding S(t); part_i / part_t unused
    UpdateArgs up;              // up.n.xbuf = the same shadow; up.xout = exchange buffer; up.xout2 = other shadow
};

// ... or the chunk partials of this workgroup in LDS
struct LdsSums {
    const float (*pi)[64];
    const float (*pt)[RESIDENT_MAX_CHUNKS][64];
    uint32_t n_chunks, lane;
    __device__ __forceinline__ float elec() const
    {
        float s = 0.0f;
        for (uint32_t c = 0; c < n_chunks; ++c) s += pi[c][lane];
        return s;
    }
    __device__ __forceinline__ float chem(int k) const
    {
        float s = 0.0f;
        for (uint32_t c = 0; c < n_chunks; ++c) s += pt[k][c][lane];
        return s;
    }
};

template <int MODEL, bool ELEC, bool CHEM>
__global__ __launch_bounds__(64 * RESIDENT_MAX_CHUNKS) void k_step_resident(const ResidentArgs a)
{
    constexpr uint32_t B = 32;                       // rows per register batch
    __shared__ float s_pi[RESIDENT_MAX_CHUNKS][64];
    __shared__ float s_pt[CHEM ? K_TYPES : 1][RESIDENT_MAX_CHUNKS][64];

    const InputsArgs &in = a.in;
    const uint32_t lane = threadIdx.x & 63u;
    // blockDim.x = 64 * n_chunks; readfirstlane tells the compiler the chunk is wave-uniform, so that row
    // addresses live in scalar registers (saddr + lane offset loads) instead of one 64-bit VGPR pair per row
    const uint32_t chunk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x;
    const uint32_t ql = tile * 64 + lane;                    // W rows are padded to a multiple of 64 columns (NaN)
    const bool col = ql < in.n_loc;
    const float vq = (ELEC && col) ? in.xbuf[in.xl.at(in.q0 + ql, PLANE_V)] : 0.0f;
    const float gq = (ELEC && col) ? in.gap_conductance[in.q0 + ql] : 0.0f;

    float acc = 0.0f, tacc[CHEM ? K_TYPES : 1];
#pragma unroll
    for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k) tacc[k] = 0.0f;

    const uint32_t p0 = chunk * CHUNK;
    const uint32_t rows = min((uint32_t)CHUNK, in.n_tot - p0);
    const size_t ld = in.ld;

    // One 64-row block of the chunk: the presynaptic state of row b0 + lane (one row per lane, broadcast later with
    // v_readlane) and this lane's 64 weights.  Block b+1 is loaded before block b is summed.
    struct Block {
        float val, tval[CHEM ? K_TYPES : 1];
        uint32_t kind;
        float wa[B], wb[B];
    };
    auto load_block = [&](Block &blk, uint32_t b0) {
        const uint32_t rb = min(64u, rows - b0);             // rows of this block (wave-uniform)
        blk.val = 0.0f;
        blk.kind = KIND_NEURON;
#pragma unroll
        for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k) blk.tval[k] = 0.0f;
        if (lane < rb) {
            const uint32_t p = p0 + b0 + lane;
            if (p < in.n_neurons) {
                blk.val = in.xbuf[in.xl.at(p, PLANE_V)];
                if (CHEM) {
#pragma unroll
                    for (int k = 0; k < K_TYPES; ++k) {
                        blk.kind |= in.nt_flags[(size_t)k * in.n_pad + p] ? (0x100u << k) : 0u;
                        blk.tval[k] = in.xbuf[in.xl.at(p, PLANE_T0 + k)];
                    }
                }
            } else {
                const uint32_t s = p - in.n_neurons;
                blk.val = in.st_value[s];
                blk.kind = (in.st_last_firing_time[s] < 0) ? KIND_ST_SILENT : KIND_ST_FIRED;
                if (CHEM) {
#pragma unroll
                    for (int k = 0; k < K_TYPES; ++k) {
                        blk.kind |= in.st_nt_flags[(size_t)k * in.c_pad + s] ? (0x100u << k) : 0u;
                        blk.tval[k] = in.st_nt_t[(size_t)k * in.c_pad + s];
                    }
                }
            }
        }
        // quad-row order: one dwordx4 = 4 consecutive rows of this lane's column; + ql is in bounds for padding columns
        // too; row groups past the end of the matrix = absent edges (the padding rows inside the last group are NaN)
        const v4f *units = reinterpret_cast<const v4f *>(in.W) + (size_t)((p0 + b0) >> 2) * ld + ql;
        const v4f none = {quiet_nan(), quiet_nan(), quiet_nan(), quiet_nan()};
#pragma unroll
        for (uint32_t g = 0; g < B / 4; ++g) {
            const v4f x = (4 * g < rb) ? units[(size_t)g * ld] : none;
            blk.wa[4 * g] = x.x; blk.wa[4 * g + 1] = x.y; blk.wa[4 * g + 2] = x.z; blk.wa[4 * g + 3] = x.w;
        }
#pragma unroll
        for (uint32_t g = 0; g < B / 4; ++g) {
            const v4f x = (B + 4 * g < rb) ? units[(size_t)(B / 4 + g) * ld] : none;
            blk.wb[4 * g] = x.x; blk.wb[4 * g + 1] = x.y; blk.wb[4 * g + 2] = x.z; blk.wb[4 * g + 3] = x.w;
        }
    };

    Block cur, nxt;
    load_block(cur, 0);
    for (uint32_t b0 = 0; b0 < rows; b0 += 64) {
        const uint32_t rb = min(64u, rows - b0);
        if (b0 + 64 < rows) load_block(nxt, b0 + 64);
        const float val = cur.val;
        const uint32_t kind = cur.kind;
        const float (&tval)[CHEM ? K_TYPES : 1] = cur.tval;
        const float (&wa)[B] = cur.wa;
        const float (&wb)[B] = cur.wb;

        // The per-row work is a dependent chain on one wavefront (4 clocks per VALU instruction), so the common
        // shapes of a 64-row block get straight-line bodies: every row a neuron, and every / no row carrying a
        // transmitter type (wave-uniform ballots).  Rows past the end of the chunk hold the absent-edge sentinel.
        const unsigned long long live = (rb == 64) ? ~0ull : ((1ull << rb) - 1ull);
        const bool all_neurons = (__ballot((kind & 3u) == KIND_NEURON) & live) == live;
        bool all_k[CHEM ? K_TYPES : 1], any_k[CHEM ? K_TYPES : 1];
#pragma unroll
        for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k) {
            const unsigned long long has = CHEM ? (__ballot((kind & (0x100u << k)) != 0) & live) : 0ull;
            all_k[k] = has == live;
            any_k[k] = has != 0ull;
        }
        auto bcast = [&](float x, uint32_t r) {
            return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), r));
        };
        // A row's product does not depend on the running sum, so the 32 products of a batch are formed first
        // (independent instructions the scheduler can interleave) and only `sum += product` is serial.  An
        // absent edge contributes +0.0f instead of being skipped: the sum starts at +0.0f and therefore never
        // holds -0.0f, so x + 0.0f == x bit for bit.
        auto sweep = [&](float &sum, auto term, auto present) {
            float pr[B];
#pragma unroll
            for (uint32_t u = 0; u < B; ++u) {
                const float p = term(u) * wa[u];             // formed unconditionally (NaN for an absent edge), selected below
                pr[u] = (wa[u] == wa[u] && present(u)) ? p : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < B; ++u) sum += pr[u];
#pragma unroll
            for (uint32_t u = 0; u < B; ++u) {
                const float p = term(B + u) * wb[u];
                pr[u] = (wb[u] == wb[u] && present(B + u)) ? p : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < B; ++u) sum += pr[u];
        };
        auto always = [](uint32_t) { return true; };
        if (ELEC) {
            if (all_neurons) {                               // gap_junction neuron/mod.rs:54-60
                sweep(acc, [&](uint32_t r) { return gq * (bcast(val, r) - vq); }, always);
            } else {                                         // + spike_train_gap_junction :119-137
                sweep(acc, [&](uint32_t r) {
                    const uint32_t src = __builtin_amdgcn_readlane(kind, r) & 3u;
                    const float vp = bcast(val, r);
                    return (src == KIND_NEURON) ? gq * (vp - vq) : ((src == KIND_ST_SILENT) ? vp : gq * vp);
                }, always);
            }
        }
        if (CHEM) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                if (all_k[k])
                    sweep(tacc[k], [&](uint32_t r) { return bcast(tval[k], r); }, always);
                else if (any_k[k])
                    sweep(tacc[k], [&](uint32_t r) { return bcast(tval[k], r); },
                          [&](uint32_t r) { return (__builtin_amdgcn_readlane(kind, r) & (0x100u << k)) != 0; });
            }
        }
        cur = nxt;
    }

    if (ELEC) s_pi[chunk][lane] = acc;
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) s_pt[k][chunk][lane] = tacc[k];
    }
    __syncthreads();
    if (chunk != 0) return;

    // ---- wavefront 0: second level of the canonical sum + the neuron update of these 64 columns ----
    uint32_t spike = 0;
    if (col) spike = update_neuron<MODEL>(a.up, ql, LdsSums{s_pi, s_pt, in.n_chunks, lane});
    if (a.up.spike_row) {
        const unsigned long long word = __ballot(spike != 0);
        if (lane == 0) a.up.spike_row[(a.up.q0 + ql) >> 6] = word;
    }
}

} // namespace snn

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
#include <cstddef>
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
    warm_kernel_arguments<sizeof(ResidentArgs)>();
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
    if (col) spike = update_neuron<MODEL, LdsSums, CHEM>(a.up, ql, LdsSums{s_pi, s_pt, in.n_chunks, lane});
    if (a.up.spike_row) {
        const unsigned long long word = __ballot(spike != 0);
        if (lane == 0) a.up.spike_row[(a.up.q0 + ql) >> 6] = word;
    }
}

// k_step_resident_q -- the same step with a chunk's 256 rows spread over FOUR wavefronts (round 6).  In k_step_resident one
// wavefront walks a chunk's rows alone: broadcast, difference, two products, the absent-edge select and the add are a dependent
// instruction stream of about a hundred clocks per row and synapse kind -- 16x16 with gap junctions and one transmitter type:
// 17.9 us per step of which the sums are 13.  Here wavefront (chunk c, quarter r) owns rows [256 c + 64 r, + 64): it forms the 64
// products of a plane on its own (independent instructions), and only the adds are taken in turn -- quarter 0 from 0.0f, quarter
// 1 continuing from the running sum quarter 0 left in LDS, ... : the canonical ascending order, add for add.  The planes (gap
// junctions, then every live transmitter type) are staggered: in turn t quarter r adds plane t - r and then forms the products
// of its next plane while the other quarters add.  4 + planes - 1 turns, a workgroup barrier between two turns.
// Workgroup = 64 columns x 4 quarters x chunks: up to 512 threads (two chunks: 512 presynaptic rows); larger networks keep
// k_step_resident.  Registers: 64 weights + 64 products per lane.
template <int MODEL, bool ELEC, bool CHEM>
__global__ __launch_bounds__(512) void k_step_resident_q(const ResidentArgs a)
{
    constexpr uint32_t R = 64;                        // rows per wavefront
    warm_kernel_arguments<sizeof(ResidentArgs)>();
    __shared__ float s_pi[RESIDENT_MAX_CHUNKS][64];
    __shared__ float s_pt[CHEM ? K_TYPES : 1][RESIDENT_MAX_CHUNKS][64];

    const InputsArgs &in = a.in;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunk = wave >> 2, quarter = wave & 3u;
    const uint32_t ql = blockIdx.x * 64 + lane;              // W rows are padded to a multiple of 64 columns (NaN)
    const bool col = ql < in.n_loc;
    const float vq = (ELEC && col) ? in.xbuf[in.xl.at(in.q0 + ql, PLANE_V)] : 0.0f;
    const float gq = (ELEC && col) ? in.gap_conductance[in.q0 + ql] : 0.0f;

#ifdef SNN_LAB_TIMING
    const unsigned long long lab_t0 = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t c0 = chunk * CHUNK;
    const uint32_t chunk_rows = min((uint32_t)CHUNK, in.n_tot - c0);
    const uint32_t p0 = c0 + quarter * R;
    const uint32_t rb = chunk_rows > quarter * R ? min(R, chunk_rows - quarter * R) : 0u;      // rows of this wavefront (wave-uniform)

    // Every load of the prologue is requested before any of them is looked at: ONE round trip to memory (a launch starts with
    // cold caches).  The row state used to be loaded and tested inside branches -- neuron or spike-train row, which transmitter
    // types -- whose waits the compiler keeps inside them, so the weights, requested after the branches, were a second round trip:
    // 5 400 + 3 900 shader clocks before the first turn at 16x16 + AMPA.  Here the branches choose ADDRESSES (a lane without a
    // row reads row 0; a neuron row reads word 0 of the exchange buffer where a spike-train row reads its last firing time).
    const bool has_row = lane < rb;
    const uint32_t p = has_row ? p0 + lane : 0u;
    const bool is_neuron = p < in.n_neurons;
    const uint32_t s = is_neuron ? 0u : p - in.n_neurons;
    const float raw_val = *(is_neuron ? in.xbuf + in.xl.at(p, PLANE_V) : in.st_value + s);
    const int32_t raw_lft = *(is_neuron ? reinterpret_cast<const int32_t *>(in.xbuf) : in.st_last_firing_time + s);
    uint32_t raw_flag[CHEM ? K_TYPES : 1];
    float raw_t[CHEM ? K_TYPES : 1];
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) {
            raw_flag[k] = *(is_neuron ? in.nt_flags + (size_t)k * in.n_pad + p : in.st_nt_flags + (size_t)k * in.c_pad + s);
            raw_t[k] = *(is_neuron ? in.xbuf + in.xl.at(p, PLANE_T0 + k) : in.st_nt_t + (size_t)k * in.c_pad + s);
        }
    }
    // what the neuron update at the end of this launch will read (update_touch_load; a column past the end reads column 0's):
    // requested by the wavefront that will update, alone -- by all eight the 30 extra loads each kept the CU's address unit busy
    // for longer than the update saved (session 8c: 8 800 clocks to the first ballot instead of 5 400)
    UpdateTouch touch{};
    if (wave == 0) update_touch_load<MODEL, CHEM>(a.up, col ? ql : 0u, touch);
    // this lane's 64 weights
    float w[R];
    {
        // quad-row order: one dwordx4 = 4 consecutive rows of this lane's column; row groups past the end of the matrix = absent edges
        const v4f *units = reinterpret_cast<const v4f *>(in.W) + (size_t)(p0 >> 2) * in.ld + ql;
        const v4f none = {quiet_nan(), quiet_nan(), quiet_nan(), quiet_nan()};
#pragma unroll
        for (uint32_t g = 0; g < R / 4; ++g) {
            const v4f x = (4 * g < rb) ? units[(size_t)g * in.ld] : none;
            w[4 * g] = x.x; w[4 * g + 1] = x.y; w[4 * g + 2] = x.z; w[4 * g + 3] = x.w;
        }
    }
    // the presynaptic state of row p0 + lane (one row per lane, broadcast with v_readlane)
    float val = has_row ? raw_val : 0.0f, tval[CHEM ? K_TYPES : 1];
    uint32_t kind = (!has_row || is_neuron) ? KIND_NEURON : ((raw_lft < 0) ? KIND_ST_SILENT : KIND_ST_FIRED);
#pragma unroll
    for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k) {
        tval[k] = (CHEM && has_row) ? raw_t[k] : 0.0f;
        if (CHEM) kind |= (has_row && raw_flag[k]) ? (0x100u << k) : 0u;
    }
    const uint32_t touched = wave == 0 ? update_touch_fold<MODEL>(a.up, touch) : 0u;
    const unsigned long long live = (rb == 64) ? ~0ull : ((1ull << rb) - 1ull);
    const bool all_neurons = (__ballot((kind & 3u) == KIND_NEURON) & live) == live;
    auto bcast = [&](float x, uint32_t r) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), r)); };

    // plane ids of the launch (uniform): 0 = gap junctions, 1 + k = transmitter type k (the live ones)
    uint32_t plane_id[1 + K_TYPES], n_planes = 0;
    if (ELEC) plane_id[n_planes++] = 0u;
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k)
            if (a.up.live_mask >> k & 1u) plane_id[n_planes++] = 1u + (uint32_t)k;
    }
    // the 64 products of one plane: formed unconditionally (NaN for an absent edge), an absent edge or a row without the plane
    // contributes +0.0f (the sum starts at +0.0f and never holds -0.0f: x + 0.0f == x bit for bit), as in k_step_resident
    // TWO product buffers (round 6, after the phase clocks of the first form: with one buffer every turn carried somebody's
    // "add, then form the next plane" -- 2 400 clocks -- behind the barrier; 16x16 + AMPA spent 10 400 clocks in its five turns):
    // the products of the first two planes are formed by all quarters at once BEFORE the first turn, the turns are adds alone; a
    // third or fourth plane is formed into the buffer its plane-minus-two has just left.
    // (the second buffer IS the weights' registers: with at most two planes nothing needs a weight once both sets of products
    // exist; three or four planes -- NMDA / GABA next to AMPA -- keep the one-buffer schedule)
    float prA[R];
    bool zeroA = false, zeroB = false;                       // a buffer's products are all +0.0f (wave-uniform): its adds are skipped
    bool all_k[CHEM ? K_TYPES : 1], any_k[CHEM ? K_TYPES : 1];
#pragma unroll
    for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k) {
        const unsigned long long has = CHEM ? (__ballot((kind & (0x100u << k)) != 0) & live) : 0ull;
        all_k[k] = has == live && rb != 0u;
        any_k[k] = has != 0ull;
    }
    auto form = [&](uint32_t id, auto &pr, bool &pr_zero) {
        pr_zero = false;
        if (id == 0u) {
            if (rb == 0u) { pr_zero = true; return; }        // a quarter past the end of the chunk
            if (all_neurons) {                               // gap_junction neuron/mod.rs:54-60
#pragma unroll
                for (uint32_t u = 0; u < R; ++u) {
                    const float p = (gq * (bcast(val, u) - vq)) * w[u];
                    pr[u] = (w[u] == w[u]) ? p : 0.0f;
                }
            } else {                                         // + spike_train_gap_junction :119-137
#pragma unroll
                for (uint32_t u = 0; u < R; ++u) {
                    const uint32_t src = __builtin_amdgcn_readlane(kind, u) & 3u;
                    const float vp = bcast(val, u);
                    const float term = (src == KIND_NEURON) ? gq * (vp - vq) : ((src == KIND_ST_SILENT) ? vp : gq * vp);
                    const float p = term * w[u];
                    pr[u] = (w[u] == w[u]) ? p : 0.0f;
                }
            }
        } else if (CHEM) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                if (id != 1u + (uint32_t)k) continue;
                if (!any_k[k]) { pr_zero = true; continue; }         // no row of this wavefront carries the type: nothing to add
                if (all_k[k]) {
#pragma unroll
                    for (uint32_t u = 0; u < R; ++u) {
                        const float p = bcast(tval[k], u) * w[u];
                        pr[u] = (w[u] == w[u]) ? p : 0.0f;
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < R; ++u) {
                        const float p = bcast(tval[k], u) * w[u];
                        const bool has = (__builtin_amdgcn_readlane(kind, u) & (0x100u << k)) != 0;
                        pr[u] = (w[u] == w[u] && has) ? p : 0.0f;
                    }
                }
            }
        }
    };
#ifdef SNN_LAB_TIMING
    const unsigned long long lab_t1 = __builtin_amdgcn_s_memtime();      // loads issued; the transmitter flags waited for (ballots)
#endif
    asm volatile("" :: "v"(touched));                        // (landed: the weights were requested before it)
    const bool two_buffers = n_planes <= 2u;
    if (n_planes) form(plane_id[0], prA, zeroA);
    if constexpr (CHEM) {
        if (two_buffers && n_planes > 1u) form(plane_id[1], w, zeroB);          // in place: w[u] becomes the product of row u
    }
#ifdef SNN_LAB_TIMING
    const unsigned long long lab_t2 = __builtin_amdgcn_s_memtime();      // the first two planes' products formed (= the weights have landed)
#endif
    const uint32_t turns = n_planes ? 4u + n_planes - 1u : 0u;
    auto take_turn = [&](uint32_t pi, auto &pr, bool &pr_zero) {
        const uint32_t id = plane_id[pi];
        float *slot = id == 0u ? &s_pi[chunk][lane] : &s_pt[CHEM ? id - 1u : 0u][chunk][lane];
        float sum = quarter ? *slot : 0.0f;
        if (!pr_zero) {
#pragma unroll
            for (uint32_t u = 0; u < R; ++u) sum += pr[u];
        }
        if (!pr_zero || quarter == 0u) *slot = sum;
    };
    if (two_buffers) {
        // (a loop of its own: no register array changes inside it -- with the one-buffer schedule in the same loop the compiler
        // carried both arrays around the back edge, 128 moves per turn)
        for (uint32_t t = 0; t < turns; ++t) {
            const uint32_t pi = t - quarter;                 // (wraps below zero: not this wavefront's turn yet)
            if (pi == 0u) take_turn(0u, prA, zeroA);
            if constexpr (CHEM) {
                if (pi == 1u && n_planes > 1u) take_turn(1u, w, zeroB);
            }
            __syncthreads();
        }
    } else {
        for (uint32_t t = 0; t < turns; ++t) {
            const uint32_t pi = t - quarter;
            if (pi < n_planes) {
                take_turn(pi, prA, zeroA);                    // one buffer: add, then form the next plane's products
                if (pi + 1u < n_planes) form(plane_id[pi + 1u], prA, zeroA);
            }
            __syncthreads();
        }
    }
    if (CHEM) {
        // the planes of types nobody releases hold zeros (k_step_resident leaves its accumulators at 0.0f for them)
        if (quarter == 3u) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k)
                if (!(a.up.live_mask >> k & 1u)) s_pt[k][chunk][lane] = 0.0f;
        }
        __syncthreads();
    }
    if (wave != 0) return;
#ifdef SNN_LAB_TIMING
    const unsigned long long lab_t3 = __builtin_amdgcn_s_memtime();      // turns over
#endif

    // ---- wavefront 0: second level of the canonical sum + the neuron update of these 64 columns ----
    uint32_t spike = 0;
    if (col) spike = update_neuron<MODEL, LdsSums, CHEM>(a.up, ql, LdsSums{s_pi, s_pt, in.n_chunks, lane});
    if (a.up.spike_row) {
        const unsigned long long word = __ballot(spike != 0);
        if (lane == 0) a.up.spike_row[(a.up.q0 + ql) >> 6] = word;
    }
#ifdef SNN_LAB_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long lab_t4 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0 && (a.up.clock % 500) == 250)
        printf("k_step_resident_q clock %lld planes %u: issue %llu, loads+form %llu, turns %llu, update %llu (shader clocks)\n", a.up.clock, n_planes,
               lab_t1 - lab_t0, lab_t2 - lab_t1, lab_t3 - lab_t2, lab_t4 - lab_t3);
#endif
}

// k_run_resident -- MANY steps of a small dense lattice in ONE launch (electrical synapses, neurons only; first for <= 1024
// of them: BASELINE configs[0], the 32 x 32 lattice; up to 4096 with the row groups described further down).  The
// one-launch step above is a chain of dependent L2 round trips (9.5 us per launch, of which the canonical 256-long add
// chain itself is well under 1 us); here
//   * a workgroup = 64 postsynaptic columns x all rows, 16 wavefronts x 64 rows, and keeps its slab of W in REGISTERS
//     for the whole run (64 words per lane, absent edges as weight 0; wavefront 0, which also updates, keeps its in LDS);
//   * per step every workgroup needs every neuron's new voltage: the owner publishes it as an 8-byte {voltage, step
//     tag} granule (agent-scope relaxed store = write-through) into the slot of the step's parity, every thread polls
//     ONE granule (agent-scope relaxed loads bypass L1) until the tag is the step's -- measured on MI355X: 1.0 us per
//     all-to-all of 1024 granules among 16 workgroups (profiles/experiments/granule_exchange_probe.hip), against
//     1.45 us for a kernel boundary alone.  Two slots suffice: a workgroup can only publish step t + 2 after it has seen
//     every workgroup's step t + 1, which each of them publishes after it finished reading step t.  A lattice of <= 64
//     neurons is one workgroup and keeps its voltages in LDS;
//   * the canonical sum keeps its order: the 4 wavefronts of a chunk take turns (the running sum passes through LDS), a
//     turn's products are formed two rows per (packed) instruction, batch by batch, ahead of their adds; wavefront w sits
//     on SIMD w % 4, and turn = (w % 4 - chunk) & 3 puts the four wavefronts of a turn on four different SIMDs;
//   * an absent edge contributes product(term, 0) = +-0 instead of being skipped -- exact as long as the term is finite
//     (the sum never holds -0, see above); a step in which some voltage is not a small finite number re-reads the matrix
//     (absent edge = NaN) and skips explicitly instead (workgroup-uniform).
// Arithmetic, order of operations and the update code are those of k_step_resident / k_inputs_dense + k_update.
//
// Lattices of 1025 .. 4096 neurons (33 x 33 .. 64 x 64) take R = 2 .. 4 ROW GROUPS of 1024 rows per column tile: workgroup
// (tile, group) holds the 64 x 1024 slab of its group and sums the group's four chunks; the groups other than 0 publish
// their chunk sums as granules, group 0 collects them (one wavefront per remote chunk), adds all chunk sums in ascending
// order and updates.  A workgroup polls the 1024 voltages of its own rows and the 64 of its columns.  64 x 64 fills the
// device: 64 tiles x 4 groups = 256 workgroups of 1024 threads, one per CU.
constexpr uint32_t RUN_RESIDENT_GROUP_ROWS = 1024;             // rows (and threads) of a workgroup
constexpr uint32_t RUN_RESIDENT_MAX_GROUPS = 4;
constexpr uint32_t RUN_RESIDENT_MAX_NEURONS = RUN_RESIDENT_GROUP_ROWS * RUN_RESIDENT_MAX_GROUPS;
constexpr uint32_t RUN_RESIDENT_MAX_TILES = RUN_RESIDENT_MAX_NEURONS / 64;
constexpr uint32_t RUN_RESIDENT_MAX_ALL_CHUNKS = RUN_RESIDENT_MAX_NEURONS / CHUNK;
constexpr uint32_t RUN_RESIDENT_SPIN_LIMIT = 1u << 24;
constexpr uint32_t RUN_GRANULE_PLANES = 1 + K_TYPES;

struct ResidentRunArgs {
    InputsArgs in;                  // W, ld, n_loc, n_tot, xbuf = the exchange buffer itself, gap conductances
    UpdateArgs up;                  // in place: n.xbuf = xout = exchange buffer, xout2 = null; history rows of the FIRST step
    uint32_t steps;
    uint32_t vhist_stride, raster_stride;     // elements between consecutive history rows
    unsigned long long *granules;   // [2][RUN_GRANULE_PLANES][RUN_RESIDENT_MAX_NEURONS] {value bits, tag << 32}: plane 0 the voltage,
                                    // plane 1 + j what the neuron releases of live transmitter type j (CHEM)
    unsigned long long *partials;   // [2][tiles][groups - 1][4 chunks][64] {chunk sum bits, tag << 32}; groups > 1 only
    uint32_t n_groups;              // row groups per column tile; gridDim.x = tiles * n_groups
    uint32_t tag_base;              // tag of the state after step s (1-based) = tag_base + s
    uint32_t *failed;               // host-visible word, set when a poll gave up (workgroups not co-resident)
    uint32_t spin_limit;            // polls before a waiter gives up (RUN_RESIDENT_SPIN_LIMIT; option "run_resident_spin_limit")
    uint32_t fault_step;            // test hook (option "run_resident_fault_step"): workgroup 0 does not publish the state after
                                    // this step (1-based; 0 = off), so every reader of its columns times out
    unsigned long long *timing;     // null, or [gridDim.x][4] shader-clock totals of workgroup phases (SNN_AMD_RUN_TIMING=1)
    // spike-train cells (Poisson or Rate, no transmitters): rows n_neurons .. n_tot - 1.  Nothing about a cell depends on the
    // rest of the network, so every workgroup advances the cells among ITS rows by itself (thread = row; the cell's state lives
    // in the workgroup's LDS); the workgroups of column tile 0 write histories and the final state.
    CellArrays cells;
    int st_kind;
    const long long *lattice_clock; // [n_st_lattices] clocks at the start of this run call
    long long step_offset0;         // steps done since then when the launch starts
    long long view_clock0;          // network clock of the launch's first input calculation
    float *st_vhist_row;            // the cells' voltage history row of the FIRST step or null
    uint32_t st_vhist_stride;
    // chemical synapses (CHEM variants): the transmitter types some neuron releases, ascending; a neuron's concentration of a
    // type it does not release travels as 0 (the presence test of the canonical sum becomes a zero product)
    uint32_t n_live, live_type[K_TYPES];
    // STDP inside the run (STDP variants: electrical synapses, neurons only, one row group): the rule per lattice
    // ([n_lattices][PL_STRIDE], at most RUN_STDP_MAX_LATTICES), who is plastic, the lattice of every neuron, and where the
    // workgroups leave their weights when the run has completed (the layout of W; the host copies it over W on success, so a
    // run that gives up leaves W as it was)
    const float *stdp_table;
    const uint32_t *stdp_on;
    const uint32_t *stdp_lattice;
    uint32_t stdp_lattices;
    float *w_out;
};
constexpr uint32_t RUN_STDP_MAX_LATTICES = 4;

typedef float v2f __attribute__((ext_vector_type(2)));

// the granules of the state a run starts from (slot 0, tag = tag_base): voltages, and per live transmitter type what each neuron
// releases of it
__global__ __launch_bounds__(256) void k_run_resident_seed(const float *xbuf, XLayout xl, uint32_t n, unsigned long long *granules, uint32_t tag,
                                                           const uint32_t *nt_flags, uint32_t n_pad, uint32_t n_live, uint32_t live0, uint32_t live1, uint32_t live2)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    granules[i] = ((unsigned long long)tag << 32) | __float_as_uint(xbuf[xl.at(i, PLANE_V)]);
    const uint32_t live[K_TYPES] = {live0, live1, live2};
    for (uint32_t j = 0; j < n_live; ++j) {
        const float t = nt_flags[(size_t)live[j] * n_pad + i] ? xbuf[xl.at(i, PLANE_T0 + live[j])] : 0.0f;
        granules[(size_t)(1 + j) * RUN_RESIDENT_MAX_NEURONS + i] = ((unsigned long long)tag << 32) | __float_as_uint(t);
    }
}

// Order matters: on gfx950 LDS addresses above 64 KB are several times slower than the first 64 KB (measured here: the same
// broadcast reads of a 64-row chain took 1.56 us from an array at 96 KB and 0.52 us from one at 4 KB).  What the turns read row by
// row and batch by batch comes first; what a thread touches once per step (the cells' own state) lies above the line.
struct ResidentRunShared {
    // ---- below 64 KB: read inside the chains ----
    float v[RUN_RESIDENT_GROUP_ROWS];            // S(t): the voltages of this workgroup's rows
    float vcol[64];                              // ... and of its columns
    // chemical synapses: per live transmitter type the concentrations of the rows, the running sum of a chunk between turns,
    // and -- by TYPE, as LdsSums reads them -- the finished chunk sums (one row group: at most 4 chunks)
    float t[K_TYPES][RUN_RESIDENT_GROUP_ROWS];
    float hand_t[K_TYPES][RESIDENT_MAX_CHUNKS][64];
    float pt[K_TYPES][RESIDENT_MAX_CHUNKS][64];
    float hand[RESIDENT_MAX_CHUNKS][64];         // running sum of a chunk, from one wavefront's turn to the next
    float pi[RUN_RESIDENT_MAX_ALL_CHUNKS][64];   // finished chunk sums: own group's at 4 * group .., group 0 also the collected ones
    v4f w0[16][64];                              // wavefront 0's weights (its registers belong to the update)
    uint32_t ok[16], plain[16];                  // per wavefront: all its granules arrived / all its values small and finite
    uint32_t w_finite[16];                       // per wavefront: every weight it holds is finite
    uint32_t gave_up;                            // sticky: some poll of chunk sums gave up
    uint32_t block_fired[16], block_silent[16];  // per 64-row block of the workgroup: all its cells have fired / none has
    // wavefront 0, CHEM with the update in registers: the per-neuron constants of the receptors and transmitters, [slot][lane]
    // (per type k: alpha, beta, g, e of the receptor, t_max, clearance, v_p, k_p of the transmitter at 8 * k ..; slot 24: NMDA mg)
    float chem_par[25][64];
    float chem_state[9][64];                     // ... and their state: r, current, t of type k at 3 * k ..
    uint32_t chem_cnt[K_TYPES][64];              // ... and the receptor's input count per type
    // STDP variants (no cells: the rows' arrays of the cells serve -- cell_s: the column deltas of the lattice in hand, cell_f:
    // every row's last firing time (int32 bits), cell_n: 1 where the row's lattice is plastic, kind: 1 where a plastic row spiked in
    // the previous step): the rule of each
    // column's lattice, the table of all rules, per lattice the columns of this tile that belong to it, the plastic columns that
    // spiked in the previous step
    float stdp_par[5][64];
    float stdp_tab[RUN_STDP_MAX_LATTICES][PL_STRIDE];
    uint32_t stdp_latmask[RUN_STDP_MAX_LATTICES][2], stdp_colmask[2];
    // networks with cells: per row of the workgroup, its kind and this step's (s, f) of a row (see the step loop) and n: 1 for a
    // neuron's row, 0 for a cell's
    float cell_s[RUN_RESIDENT_GROUP_ROWS], cell_f[RUN_RESIDENT_GROUP_ROWS];
    float cell_n[RUN_RESIDENT_GROUP_ROWS];
    uint32_t kind[RUN_RESIDENT_GROUP_ROWS];      // KIND_NEURON / KIND_ST_SILENT / KIND_ST_FIRED (read where a value is not finite)
    // ---- above 64 KB: a cell's own state, touched by its thread once per step ----
    uint32_t cell_word[RUN_RESIDENT_GROUP_ROWS]; // Poisson: seed; Rate: step (float bits)
    int32_t cell_lft[RUN_RESIDENT_GROUP_ROWS];
    float cell_presyn[RUN_RESIDENT_GROUP_ROWS], cell_v[RUN_RESIDENT_GROUP_ROWS];
    // a cell's parameters, read once: chance of firing | rate, v_th, v_resting, dt, k, refractoriness kind, and the clock of its
    // lattice at the launch's first step (64 bits in two words) -- read from the arrays every step they were a chain of
    // dependent global loads per cell
    float cell_par[5][RUN_RESIDENT_GROUP_ROWS];
    uint32_t cell_refr[RUN_RESIDENT_GROUP_ROWS], cell_clock_lo[RUN_RESIDENT_GROUP_ROWS], cell_clock_hi[RUN_RESIDENT_GROUP_ROWS];
    uint32_t cell_spiking[RUN_RESIDENT_GROUP_ROWS];
};
static_assert(offsetof(ResidentRunShared, cell_n) <= 65536, "the arrays the chains read must lie in the first 64 KB of LDS");


// The step loop of one wavefront.  UPDATER = wavefront 0, which also owns the neuron update of the workgroup's 64 columns: it
// keeps its weights in LDS and, for Izhikevich neurons without transmitters, the neurons' state in registers for the whole
// run (same expressions as update_neuron, integrate_and_fire/mod.rs:217-255); the other wavefronts keep 64 weights per lane
// in registers.  Every wavefront of a workgroup passes the same workgroup barriers per step (one behind the polls, one per turn --
// four, fewer for networks under 256 rows --; group 0 of a multi-group tile one more).
template <int MODEL, bool UPDATER, bool REGISTERS, bool CELLS, bool CHEM, bool STDP = false, bool LEND = false>
__device__ __forceinline__ void run_resident_steps(const ResidentRunArgs &a, ResidentRunShared &sh, const uint32_t wave)
{
    const InputsArgs &in = a.in;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t n_groups = a.n_groups, tile = blockIdx.x / n_groups, group = blockIdx.x % n_groups;
    const uint32_t n_tiles = gridDim.x / n_groups;
    const uint32_t chunk_local = wave >> 2, chunk = 4u * group + chunk_local;
    const uint32_t turn = ((wave & 3u) - chunk_local) & 3u;      // which quarter of the chunk, and when
    const uint32_t n_tot = in.n_tot, steps = a.steps, tag_base = a.tag_base, n_chunks = in.n_chunks;
    // LEND (the host launches it for networks of one chunk at most, <= 256 rows, with gap junctions AND transmitters): the
    // wavefronts of the idle chunks 1 + j take the chain of live type j over chunk 0's rows -- in the same slot as the
    // gap-junction chain of those rows, on another SIMD ((wave & 3) differs by the chunk index) -- so all chains of a turn run
    // side by side: 4 slots instead of 4 + n_live
    const bool shadow = LEND && chunk_local >= 1u && chunk_local <= a.n_live;
    const uint32_t row0 = (shadow ? 0u : chunk) * CHUNK + turn * 64u, group_row0 = group * RUN_RESIDENT_GROUP_ROWS;
    const bool rows_live = row0 < n_tot;
    const bool last_of_chunk = turn == 3u || row0 + 64u >= n_tot;
    const bool updates = group == 0u;                            // this workgroup's wavefront 0 updates the tile's neurons
    const bool cols_in_rows = (tile * 64u) / RUN_RESIDENT_GROUP_ROWS == group;   // its columns are among its rows: no second poll
    const bool alone = gridDim.x == 1u;                          // <= 64 neurons: the only workgroup
    const uint32_t n_turns = n_tot >= CHUNK ? 4u : (n_tot + 63u) / 64u;   // fewer than 256 rows in all: fewer turns (and barriers)
    float v_mine = 0.0f;                                         // wavefront 0, alone: the voltage this lane's neuron was left with
    const uint32_t ql = tile * 64u + lane;
    const bool col = ql < in.n_loc;
    const float gq = col ? uload_vector(in.uni, NP_GAP, in.gap_conductance, in.q0 + ql) : 0.0f;
    const bool gq_small = fabsf(gq) <= 1e15f;
    const unsigned long long *granules = a.granules;

    // this lane's 64 weights (quad-row units: 4 consecutive rows of one column per load); absent edge: weight 0
    float w[UPDATER ? 1 : 64];
    uint32_t ex_lo = 0u, ex_hi = 0u;             // STDP: bit k = this lane's column has an edge from row row0 + k
    bool w_all_finite = true;
    const v4f *units = reinterpret_cast<const v4f *>(in.W) + (size_t)(row0 >> 2) * in.ld + ql;
    {
#pragma unroll
        for (uint32_t g = 0; g < 16; ++g) {
            v4f x = {quiet_nan(), quiet_nan(), quiet_nan(), quiet_nan()};
            if (row0 + 4 * g < n_tot) x = units[(size_t)g * in.ld];
            float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const bool edge = e[k] == e[k];
                if (STDP && edge) { if (g < 8) ex_lo |= 1u << (4 * g + k); else ex_hi |= 1u << (4 * (g - 8) + k); }
                e[k] = edge ? e[k] : 0.0f;
                w_all_finite = w_all_finite && fabsf(e[k]) <= 3.0e38f;
                if (!UPDATER) w[4 * g + k] = e[k];
            }
            if (UPDATER) sh.w0[g][lane] = v4f{e[0], e[1], e[2], e[3]};      // read back by this lane only
        }
    }
    // CHEM: a concentration that travels as 0 (the neuron does not release the type) stands for a skipped term only while its
    // product with the weight is a zero: every weight finite (wave-uniform, fixed for the run)
    if (CHEM) {
        const bool all = __all(w_all_finite);
        if (lane == 0) sh.w_finite[wave] = all;
    }
    const uint32_t n_live = CHEM ? a.n_live : 0u;
    // the transmitter types this lane's neuron releases (UPDATER, CHEM): bit k
    uint32_t my_nt_mask = 0u;
    if (CHEM && UPDATER && updates && col && a.up.has_nt) {
        const uint32_t q = a.up.rows.global_of(ql);
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) my_nt_mask |= a.up.n.nt_flags[(size_t)k * a.up.n.n_pad + q] ? (1u << k) : 0u;
    }
    float t_mine[K_TYPES] = {0.0f, 0.0f, 0.0f};     // wavefront 0, alone: what this lane's neuron released (by live slot)

    // Neuron state in registers for the whole run (REGISTERS: the host launches that variant for Izhikevich, leaky, quadratic and
    // simple leaky integrate-and-fire lattices without transmitters and without the BCM extension).  The expressions are
    // update_neuron's (integrate_and_fire/mod.rs:217-255, 173-215, 324-365, 1577-1630); what they read every step is read once.
    constexpr bool in_registers = UPDATER && REGISTERS;
    static_assert(!REGISTERS || MODEL == 0 || MODEL == 1 || MODEL == 3 || MODEL == 4, "no register-resident update for this model");
    float nv = 0.0f, n2 = 0.0f, n_div = 1.0f;        // voltage; Izhikevich w / refractory_count; the averager's divisor
    float pr[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // the model's parameters (see below)
    uint32_t n_spikes = 0, last_spike = 0;
    // ... and, CHEM (Izhikevich only): per type the receptor (r, current, g, e, kinetics parameters, input count) and the
    // transmitter (t, its kinetics parameters); c_flags bit k: receptor present, bit 8 + k: the neuron releases the type
    // (the constants of the kinetics wait in LDS: ResidentRunShared::chem_par)
    float c_dt = 0.0f;
    uint32_t c_flags = 0u;
    if (in_registers && updates && col) {
        const NeuronArrays &n = a.up.n;
        const uint32_t q = a.up.rows.global_of(ql);
        nv = n.xbuf[n.xl.at(q, PLANE_V)];
        const uint32_t cnt = a.up.n_in[ql];
        n_div = cnt == 0 ? 1.0f : (float)cnt;
        const float dt = uload_vector(n.uni, NP_DT, n.dt, q);
        if (MODEL == 0) {
            n2 = n.w_value[q];
            pr[0] = dt / uload_vector(n.uni, NP_C_M, n.c_m, q);
            pr[1] = dt / uload_vector(n.uni, NP_TAU_M, n.tau_m, q);
            pr[2] = uload_vector(n.uni, NP_A, n.a, q); pr[3] = uload_vector(n.uni, NP_B, n.b, q);
            pr[4] = uload_vector(n.uni, NP_C, n.c, q); pr[5] = uload_vector(n.uni, NP_D, n.d, q);
            pr[6] = uload_vector(n.uni, NP_V_TH, n.v_th, q);
        } else if (MODEL == 1) {
            n2 = n.refractory_count[q];
            pr[0] = n.leak_constant[q]; pr[1] = n.e_l[q]; pr[2] = n.integration_constant[q]; pr[3] = n.g_l[q];
            pr[4] = dt / n.tau_m[q]; pr[5] = n.v_reset[q]; pr[6] = n.v_th[q]; pr[7] = n.tref[q] / dt;
        } else if (MODEL == 3) {
            n2 = n.refractory_count[q];
            pr[0] = n.qif_alpha[q]; pr[1] = n.v_reset[q]; pr[2] = n.qif_v_c[q]; pr[3] = n.integration_constant[q];
            pr[4] = dt / n.tau_m[q]; pr[6] = n.v_th[q]; pr[7] = n.tref[q] / dt;
        } else {
            pr[0] = n.slif_g[q]; pr[1] = n.slif_e[q]; pr[2] = dt; pr[5] = n.v_reset[q]; pr[6] = n.v_th[q];
        }
        last_spike = reinterpret_cast<const uint32_t *>(n.xbuf)[n.xl.at(q, PLANE_SPIKE)];
        if (CHEM) {
            // Izhikevich with chemical synapses, built-in kinetics: receptors and transmitters in registers too (ChemStep's
            // arithmetic, snn_kernels_update.hpp; iterate_and_spike/mod.rs:1186-1304, 148-196)
            c_dt = dt;
            sh.chem_par[24][lane] = n.rc_mg[(size_t)1 * n.n_pad + q];
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                const size_t i = (size_t)k * n.n_pad + q;
                c_flags |= n.rc_flags[i] ? (1u << k) : 0u;
                c_flags |= (a.up.has_nt && n.nt_flags[i]) ? (0x100u << k) : 0u;
                sh.chem_cnt[k][lane] = a.up.tcount[(size_t)k * a.up.ld + ql];
                sh.chem_state[3 * k][lane] = n.rc_r[i];
                sh.chem_state[3 * k + 1][lane] = n.rc_current[i];
                sh.chem_state[3 * k + 2][lane] = n.xbuf[n.xl.at(q, PLANE_T0 + k)];
                sh.chem_par[8 * k + 0][lane] = n.rc_alpha[i]; sh.chem_par[8 * k + 1][lane] = n.rc_beta[i];
                sh.chem_par[8 * k + 2][lane] = n.rc_g[i]; sh.chem_par[8 * k + 3][lane] = n.rc_e[i];
                sh.chem_par[8 * k + 4][lane] = n.nt_t_max[i]; sh.chem_par[8 * k + 5][lane] = n.nt_clearance[i];
                sh.chem_par[8 * k + 6][lane] = n.nt_v_p[i]; sh.chem_par[8 * k + 7][lane] = n.nt_k_p[i];
            }
        }
    }

    // this thread's row as a spike-train cell (CELLS: the variant launched for networks with cells)
    const uint32_t n_neurons = in.n_neurons, my_row = group_row0 + tid;
    const bool is_cell = CELLS && my_row >= n_neurons && my_row < n_tot;
    const bool all_cells = CELLS && rows_live && row0 >= n_neurons;                       // this wavefront's rows: cells only
    const bool mixed = CELLS && rows_live && row0 < n_neurons && row0 + 64u > n_neurons;   // ... neurons and cells
    const bool writes_cells = is_cell && tile == 0u;
    const uint32_t cell = is_cell ? my_row - n_neurons : 0u;
    if (is_cell) {
        const CellArrays &c = a.cells;
        sh.cell_par[0][tid] = a.st_kind == 1 ? uload_vector(c.uni, CP_CHANCE, c.chance_of_firing, cell) : c.rate[cell];
        sh.cell_par[1][tid] = uload_vector(c.uni, CP_V_TH, c.v_th, cell);
        sh.cell_par[2][tid] = uload_vector(c.uni, CP_V_RESTING, c.v_resting, cell);
        sh.cell_par[3][tid] = uload_vector(c.uni, CP_DT, c.dt, cell);
        sh.cell_par[4][tid] = uload_vector(c.uni, CP_K, c.k, cell);
        sh.cell_refr[tid] = uload_vector(c.uni, CP_REFR, c.refractoriness, cell);
        const long long clock0 = a.lattice_clock[c.lattice_slot[cell]] + a.step_offset0;
        sh.cell_clock_lo[tid] = (uint32_t)(unsigned long long)clock0;
        sh.cell_clock_hi[tid] = (uint32_t)((unsigned long long)clock0 >> 32);
        sh.cell_word[tid] = a.st_kind == 1 ? a.cells.seed[cell] : __float_as_uint(a.cells.step[cell]);
        sh.cell_lft[tid] = a.cells.last_firing_time[cell];
        sh.cell_presyn[tid] = a.cells.presyn_value[cell];
        sh.cell_v[tid] = a.cells.current_voltage[cell];
        sh.cell_spiking[tid] = a.cells.is_spiking[cell];
    }

    // STDP inside the run: this thread's ROW (its last firing time, whether its lattice is plastic) and, for the wavefront
    // that updates, this lane's COLUMN; the rule of every column's lattice and the table of all rules go to LDS
    // (the row's two values live in LDS -- cell_f: last firing time, cell_n: 1 where the row's lattice is plastic -- the registers
    // are the weights')
    bool col_plastic = false;
    uint32_t spk_mine = 0u;                      // alone: whether this lane's neuron spiked in the previous step (wavefront 0)
    if (STDP) {
        int32_t lft_row = -1;
        bool row_plastic = false;
        if (my_row < n_neurons) {
            lft_row = a.up.n.last_firing_time[my_row];
            row_plastic = a.stdp_on[a.stdp_lattice[my_row]] != 0u;
        }
        reinterpret_cast<int32_t *>(sh.cell_f)[tid] = lft_row;
        sh.cell_n[tid] = row_plastic ? 1.0f : 0.0f;
        sh.kind[tid] = 0u;
        if (tid < a.stdp_lattices * (uint32_t)PL_STRIDE) (&sh.stdp_tab[0][0])[tid] = a.stdp_table[tid];
        const uint32_t lat = col ? a.stdp_lattice[in.q0 + ql] : 0u;
        col_plastic = col && a.stdp_on[lat] != 0u;
        if (wave == 1u || alone) {
#pragma unroll
            for (int i = 0; i < 5; ++i) sh.stdp_par[i][lane] = a.stdp_table[(size_t)lat * PL_STRIDE + i];
            for (uint32_t l = 0; l < a.stdp_lattices; ++l) {
                const unsigned long long m = __ballot(col && lat == l);
                if (lane == 0) { sh.stdp_latmask[l][0] = (uint32_t)m; sh.stdp_latmask[l][1] = (uint32_t)(m >> 32); }
            }
        }
        if (tid == 0) { sh.stdp_colmask[0] = 0u; sh.stdp_colmask[1] = 0u; }
    }

    unsigned long long spent[4] = {0, 0, 0, 0}, mark = a.timing ? clock64() : 0;
    auto lap = [&](int phase) {
        if (!UPDATER || !a.timing) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // keeps the phases' code on its side of the mark
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = clock64();
        __builtin_amdgcn_sched_barrier(0);
        spent[phase] += now - mark;
        mark = now;
    };

    const uint32_t tid_outer = tid;
    bool completed = true;                       // (a poll that gives up ends the loop early: the weights then stay where they are)
    for (uint32_t s = 0; s < steps; ++s) {
        // What a step derives from the thread's index is derived again in every step (the index passes through an opaque
        // instruction): kept across the loop, these values and the 64-bit addresses built on them cost the registers the
        // weights need, and the compiler spilled them to scratch inside the loop.
        uint32_t tid_l = tid_outer;
        asm volatile("" : "+v"(tid_l));
        const uint32_t tid = tid_l, lane = tid & 63u, ql = tile * 64u + lane, my_row = group_row0 + tid;
        const bool col = ql < in.n_loc;
        const bool is_cell = CELLS && my_row >= n_neurons && my_row < n_tot;
        const bool writes_cells = is_cell && tile == 0u;
        const uint32_t cell = is_cell ? my_row - n_neurons : 0u;
        const v4f *units = reinterpret_cast<const v4f *>(in.W) + (size_t)(row0 >> 2) * in.ld + ql;
        // (1) S(t): every neuron's voltage, one granule per thread (those of the first step come from k_run_resident_seed:
        // the exchange buffer itself is updated in place by workgroups that are already a step ahead)
        const unsigned long long *slot = granules + (size_t)(s & 1u) * RUN_GRANULE_PLANES * RUN_RESIDENT_MAX_NEURONS;
        const uint32_t tag = tag_base + s;
        auto poll = [&](const unsigned long long *g, bool &arrived) {
            unsigned long long x;
            uint32_t spins = 0;
            do {
                x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((uint32_t)(x >> 32) != tag && ++spins < a.spin_limit);
            arrived = arrived && (uint32_t)(x >> 32) == tag;
            return __uint_as_float((uint32_t)x);
        };
        float v = 0.0f, v_col = 0.0f;
        float tv[K_TYPES] = {0.0f, 0.0f, 0.0f};      // CHEM: this row's concentrations, by live slot
        bool arrived = true;
        uint32_t spk_row = 0u;                       // STDP: this thread's row spiked in the previous step
        if (alone && s != 0) {
            if (STDP && UPDATER) spk_row = spk_mine;
            // a lattice of <= 64 neurons is ONE workgroup: wavefront 0 left the new voltages in sh.v itself (below), nothing
            // travels through memory; it also knows whether they are all small finite numbers
            v = UPDATER ? v_mine : 0.0f;
            if (CHEM && UPDATER) { tv[0] = t_mine[0]; tv[1] = t_mine[1]; tv[2] = t_mine[2]; }
        } else {
            if (CHEM) {
                // the voltage and the live concentrations of this row: 1 + n_live granules, each with its own tag, requested
                // together (one round trip, not one per plane)
                if (my_row < n_neurons) {
                    unsigned long long x[RUN_GRANULE_PLANES];
                    uint32_t spins = 0;
                    bool all;
                    do {
                        all = true;
#pragma unroll
                        for (uint32_t j = 0; j < RUN_GRANULE_PLANES; ++j)
                            if (j <= n_live) x[j] = __hip_atomic_load(slot + (size_t)j * RUN_RESIDENT_MAX_NEURONS + my_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (uint32_t j = 0; j < RUN_GRANULE_PLANES; ++j)
                            if (j <= n_live) all = all && (uint32_t)(x[j] >> 32) == tag;
                    } while (!all && ++spins < a.spin_limit);
                    arrived = arrived && all;
                    v = __uint_as_float((uint32_t)x[0]);
#pragma unroll
                    for (uint32_t j = 0; j < K_TYPES; ++j)
                        if (j < n_live) tv[j] = __uint_as_float((uint32_t)x[1 + j]);
                }
            } else if (STDP && my_row < n_neurons) {
                // bit 31 of the tag word: the neuron spiked in the step that produced this voltage
                unsigned long long x;
                uint32_t spins = 0;
                do {
                    x = __hip_atomic_load(slot + my_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (((uint32_t)(x >> 32) & 0x7FFFFFFFu) != tag && ++spins < a.spin_limit);
                arrived = arrived && ((uint32_t)(x >> 32) & 0x7FFFFFFFu) == tag;
                spk_row = (uint32_t)(x >> 63);
                v = __uint_as_float((uint32_t)x);
            } else if (my_row < n_neurons) {
                v = poll(slot + my_row, arrived);
            }
            if (!cols_in_rows && wave == 1 && col) v_col = poll(slot + in.q0 + ql, arrived);   // the columns' voltages, by one wavefront
            if (!CELLS || my_row < n_neurons) sh.v[tid] = v;
            if (CHEM) {
                // (a spike-train cell's row carries 0: cells with transmitters keep the network on the per-step forms)
#pragma unroll
                for (uint32_t j = 0; j < K_TYPES; ++j)
                    if (j < n_live) sh.t[j][tid] = tv[j];
            }
        }
        if (CELLS && my_row >= n_neurons) {
            // spike_train_gap_junction (neuron/mod.rs:119-137): a cell's row carries its gap-junction value x; one that never
            // fired enters as x, the others as g * x.  Both as  s + g * f  with (s, f) = (x, 0) resp. (0, x): exact while x is
            // finite (+-0 is added to a sum that never holds -0), and the same two packed instructions for every row.
            const bool silent = is_cell && sh.cell_lft[tid] < 0;
            v = is_cell ? sh.cell_presyn[tid] : 0.0f;
            sh.v[tid] = v;
            sh.kind[tid] = !is_cell ? KIND_NEURON : (silent ? KIND_ST_SILENT : KIND_ST_FIRED);
            sh.cell_s[tid] = silent ? v : 0.0f;
            sh.cell_f[tid] = silent ? 0.0f : v;
            sh.cell_n[tid] = 0.0f;
        } else if (CELLS) {
            // a neuron's row in the same form (the block that holds both kinds):  s + g * (f - n * vq)  with (s, f, n) = (0, v, 1)
            sh.kind[tid] = KIND_NEURON;
            sh.cell_s[tid] = 0.0f;
            sh.cell_f[tid] = v;
            sh.cell_n[tid] = 1.0f;
        }
        if (CELLS) {
            // rows past the end carry weight 0 and value 0: they fit both uniform forms
            const bool cell_here = my_row >= n_neurons && my_row < n_tot;
            const bool fired_here = cell_here && sh.cell_lft[tid] >= 0;
            const bool fired_all = __all(!cell_here || fired_here), silent_all = __all(!cell_here || !fired_here);
            if (lane == 0) { sh.block_fired[wave] = fired_all; sh.block_silent[wave] = silent_all; }
        }
        lap(0);
        if (!cols_in_rows && wave == 1) sh.vcol[lane] = v_col;
        if (STDP && s != 0u) {
            if (spk_row) reinterpret_cast<int32_t *>(sh.cell_f)[tid] = (int32_t)(a.up.clock + (long long)s - 1);
            sh.kind[tid] = (spk_row && sh.cell_n[tid] != 0.0f) ? 1u : 0u;
        }
        const bool all_arrived = __all(arrived), all_plain = __all(fabsf(v) <= 1e15f && fabsf(v_col) <= 1e15f && gq_small &&
                                                                   (!CHEM || (fabsf(tv[0]) <= 1e15f && fabsf(tv[1]) <= 1e15f && fabsf(tv[2]) <= 1e15f)));
        if (lane == 0) { sh.ok[wave] = all_arrived; sh.plain[wave] = all_plain; }
        __syncthreads();
        const uint32_t flag_ok = sh.ok[lane & 15u], flag_plain = sh.plain[lane & 15u];
        if (!__all(flag_ok != 0) || *const_cast<volatile uint32_t *>(&sh.gave_up)) { completed = false; break; }   // workgroup-uniform: some poll gave up
        const bool plain = __all(flag_plain != 0);
        const bool plain_t = CHEM && plain && __all(sh.w_finite[lane & 15u] != 0u);
        lap(1);

        // (1b) STDP of the previous step (the deferred form, as k_spike_compact + k_stdp_columns + k_stdp_rows leave it): for
        // every plastic neuron j that spiked, the incoming edges p -> j gain delta(last_firing_time[p], now) under the rule of
        // j's lattice, then the outgoing edges j -> r gain delta(now, last_firing_time[r]) under the rule of r's lattice
        // (plasticity/mod.rs:45-66, neuron/mod.rs:2308-2417); the weights are this workgroup's registers / LDS.
        if (STDP && s != 0u) {
            const int32_t t_now = (int32_t)(a.up.clock + (long long)s - 1);
            uint32_t zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
            auto exists = [&](uint32_t k) { return ((k < 32u ? ex_lo >> k : ex_hi >> (k - 32u)) & 1u) != 0u; };
            const uint32_t cm0 = __builtin_amdgcn_readfirstlane(sh.stdp_colmask[0]), cm1 = __builtin_amdgcn_readfirstlane(sh.stdp_colmask[1]);
            if ((cm0 | cm1) != 0u) {                                   // workgroup-uniform: a plastic column of this tile spiked
                for (uint32_t l = 0; l < a.stdp_lattices; ++l) {
                    const uint32_t m0 = cm0 & __builtin_amdgcn_readfirstlane(sh.stdp_latmask[l][0]);
                    const uint32_t m1 = cm1 & __builtin_amdgcn_readfirstlane(sh.stdp_latmask[l][1]);
                    if ((m0 | m1) == 0u) continue;
                    // the column deltas of lattice l: one per row, computed by the row's thread
                    sh.cell_s[tid] = stdp_delta(reinterpret_cast<const int32_t *>(sh.cell_f)[tid], t_now, sh.stdp_tab[l][0], sh.stdp_tab[l][1], sh.stdp_tab[l][2], sh.stdp_tab[l][3], sh.stdp_tab[l][4]);
                    __syncthreads();
                    const bool mine = ((lane < 32u ? m0 >> lane : m1 >> (lane - 32u)) & 1u) != 0u;
                    if (rows_live) {
                        const v4f *dc = reinterpret_cast<const v4f *>(sh.cell_s + (row0 - group_row0) + zero);
#pragma unroll
                        for (uint32_t g = 0; g < 16; ++g) {
                            const v4f d = dc[g];
                            const float dd[4] = {d.x, d.y, d.z, d.w};
                            if (UPDATER) {
                                v4f x = sh.w0[g][lane];
                                float xx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                                for (uint32_t k = 0; k < 4; ++k) xx[k] = (mine && exists(4 * g + k)) ? xx[k] + dd[k] : xx[k];
                                sh.w0[g][lane] = v4f{xx[0], xx[1], xx[2], xx[3]};
                            } else {
#pragma unroll
                                for (uint32_t k = 0; k < 4; ++k) {
                                    const uint32_t r = UPDATER ? 0u : 4 * g + k;
                                    w[r] = (mine && exists(4 * g + k)) ? w[r] + dd[k] : w[r];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __syncthreads();                                    // the deltas are done with (another lattice may follow)
                }
            }
            if (rows_live) {
                const unsigned long long rm = __ballot(sh.kind[(row0 - group_row0) + lane] != 0u);    // my rows that spiked (plastic)
                if (rm != 0ull) {
                    const int32_t lft_col = reinterpret_cast<const int32_t *>(sh.cell_f)[(ql - group_row0) & (RUN_RESIDENT_GROUP_ROWS - 1u)];
                    const float drow = stdp_delta(t_now, lft_col, sh.stdp_par[0][lane], sh.stdp_par[1][lane], sh.stdp_par[2][lane],
                                                  sh.stdp_par[3][lane], sh.stdp_par[4][lane]);
#pragma unroll
                    for (uint32_t g = 0; g < 16; ++g) {
                        if (((rm >> (4 * g)) & 0xFull) == 0ull) continue;                              // wave-uniform
                        if (UPDATER) {
                            v4f x = sh.w0[g][lane];
                            float xx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) xx[k] = (((rm >> (4 * g + k)) & 1ull) != 0ull && exists(4 * g + k)) ? xx[k] + drow : xx[k];
                            sh.w0[g][lane] = v4f{xx[0], xx[1], xx[2], xx[3]};
                        } else {
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                const uint32_t r = UPDATER ? 0u : 4 * g + k;
                                if ((rm >> (4 * g + k)) & 1ull) w[r] = exists(4 * g + k) ? w[r] + drow : w[r];
                            }
                        }
                    }
                }
            }
        }

        // (2) the canonical chunk sums, the wavefronts of a chunk in turn
        const float vq = cols_in_rows ? sh.v[(ql - group_row0) & (RUN_RESIDENT_GROUP_ROWS - 1u)] : sh.vcol[lane];
        // With chemical synapses the chains of a wavefront are STAGGERED over the slots: slot t = its gap-junction chain's turn,
        // slot t + 1 + j its chain of live type j -- while the next wavefront of the chunk (another SIMD) already sums its
        // gap-junction rows.  The hand-over of every chain still runs turn 0 -> 3, one barrier apart; 4 + n_live slots instead
        // of 4 x (1 + n_live) chains one after the other.
        // (the variants that also carry spike-train cells have no register to spare for it: their chains stay in one slot)
        constexpr bool STAGGER = CHEM && !CELLS;
        const uint32_t chem_off = (STAGGER && a.up.electrical) ? 1u : 0u;
        const uint32_t n_slots = (STAGGER && !LEND) ? n_turns + chem_off + n_live - 1u : n_turns;
#pragma unroll 1
        for (uint32_t t = 0; t < n_slots; ++t) {
            if (rows_live) {
              if (t == turn && (!CHEM || a.up.electrical) && !shadow) {
                float acc = (turn != 0) ? sh.hand[chunk_local][lane] : 0.0f;
                // wave-uniform address = LDS broadcast; the offset is laundered through a vector register so that the values
                // stay in vector registers (as scalars every one of them costs a v_readlane plus its wait states)
                uint32_t zero;
                asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
                const v4f *pre = reinterpret_cast<const v4f *>(sh.v + (row0 - group_row0) + zero);
                // 16 rows per batch: the batch's LDS reads (4 x 16 B of voltages, for the updater 4 x 16 B of weights too) are
                // issued together and the next batch's before this one is summed -- one read at a time, each waited for,
                // costs an LDS round trip per 4 rows
                auto load_batch = [&](uint32_t b, v4f (&vp)[4], v4f (&wr)[4]) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        vp[k] = pre[4 * b + k];
                        if (UPDATER) wr[k] = sh.w0[4 * b + k][lane];
                    }
                };
                const uint4 *kinds = reinterpret_cast<const uint4 *>(sh.kind + (CELLS ? row0 - group_row0 : 0u) + zero);
                auto term_of = [&](uint32_t kind, float vp) {    // gap_junction neuron/mod.rs:54-60, spike_train_gap_junction :119-137
                    return kind == KIND_NEURON ? gq * (vp - vq) : (kind == KIND_ST_SILENT ? vp : gq * vp);
                };
                if (plain && mixed) {
                    // the one block that holds neurons and cells:  s + g * (f - n * vq), times the weight -- (0, v, 1) is the
                    // neuron's g * (v - vq), (0, x, 0) and (x, 0, 0) the two kinds of cell; every step exact while all values are
                    // finite (1 * vq == vq, 0 * vq == +-0, +-0 added to a sum that never holds -0).  Batches of 8 rows: at most one
                    // wavefront of a workgroup takes this form, and with 16 rows in flight its three operand streams pushed the
                    // weights out of the registers (spills inside the step loop).
                    const v4f *cs = reinterpret_cast<const v4f *>(sh.cell_s + (row0 - group_row0) + zero);
                    const v4f *cf = reinterpret_cast<const v4f *>(sh.cell_f + (row0 - group_row0) + zero);
                    const v4f *cn = reinterpret_cast<const v4f *>(sh.cell_n + (row0 - group_row0) + zero);
                    const v2f gq2 = {gq, gq}, vq2 = {vq, vq};
#pragma unroll
                    for (uint32_t b = 0; b < 8; ++b) {
                        v4f xs[2], xf[2], xn[2], wr[2];
#pragma unroll
                        for (uint32_t k = 0; k < 2; ++k) {
                            xs[k] = cs[2 * b + k];
                            xf[k] = cf[2 * b + k];
                            xn[k] = cn[2 * b + k];
                            if (UPDATER) wr[k] = sh.w0[2 * b + k][lane];
                        }
                        v2f d[4];
#pragma unroll
                        for (uint32_t k = 0; k < 2; ++k) {
                            d[2 * k] = v2f{xn[k].x, xn[k].y} * vq2;
                            d[2 * k + 1] = v2f{xn[k].z, xn[k].w} * vq2;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 2; ++k) {
                            d[2 * k] = v2f{xf[k].x, xf[k].y} - d[2 * k];
                            d[2 * k + 1] = v2f{xf[k].z, xf[k].w} - d[2 * k + 1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) d[k] = gq2 * d[k];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 2; ++k) {
                            d[2 * k] = v2f{xs[k].x, xs[k].y} + d[2 * k];
                            d[2 * k + 1] = v2f{xs[k].z, xs[k].w} + d[2 * k + 1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 2; ++k) {
                            const uint32_t r = 8 * b + 4 * k;
                            if (!UPDATER) wr[k] = v4f{w[UPDATER ? 0 : r], w[UPDATER ? 0 : r + 1], w[UPDATER ? 0 : r + 2], w[UPDATER ? 0 : r + 3]};
                            d[2 * k] = d[2 * k] * v2f{wr[k].x, wr[k].y};
                            d[2 * k + 1] = d[2 * k + 1] * v2f{wr[k].z, wr[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) { acc += d[k].x; acc += d[k].y; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (plain && all_cells && (sh.block_fired[(row0 - group_row0) >> 6] || sh.block_silent[(row0 - group_row0) >> 6])) {
                    // rows that are all cells of ONE kind (after a warm-up: all have fired): g * x resp. x, times the weight
                    const bool fired = sh.block_fired[(row0 - group_row0) >> 6] != 0u;
                    const v2f g2 = fired ? v2f{gq, gq} : v2f{1.0f, 1.0f};         // 1 * x == x exactly
                    v4f vp[4], wr[4];
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        load_batch(b, vp, wr);
                        v2f d[8];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            d[2 * k] = g2 * v2f{vp[k].x, vp[k].y};
                            d[2 * k + 1] = g2 * v2f{vp[k].z, vp[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            const uint32_t r = 16 * b + 4 * k;
                            if (!UPDATER) wr[k] = v4f{w[UPDATER ? 0 : r], w[UPDATER ? 0 : r + 1], w[UPDATER ? 0 : r + 2], w[UPDATER ? 0 : r + 3]};
                            d[2 * k] = d[2 * k] * v2f{wr[k].x, wr[k].y};
                            d[2 * k + 1] = d[2 * k + 1] * v2f{wr[k].z, wr[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) { acc += d[k].x; acc += d[k].y; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (plain && all_cells) {
                    // rows that are all cells: s + g * f, times the weight -- staged like the neurons' products below
                    const v4f *cs = reinterpret_cast<const v4f *>(sh.cell_s + (row0 - group_row0) + zero);
                    const v4f *cf = reinterpret_cast<const v4f *>(sh.cell_f + (row0 - group_row0) + zero);
                    const v2f gq2 = {gq, gq};
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        v4f xs[4], xf[4], wr[4];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            xs[k] = cs[4 * b + k];
                            xf[k] = cf[4 * b + k];
                            if (UPDATER) wr[k] = sh.w0[4 * b + k][lane];
                        }
                        v2f d[8];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            d[2 * k] = gq2 * v2f{xf[k].x, xf[k].y};
                            d[2 * k + 1] = gq2 * v2f{xf[k].z, xf[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            d[2 * k] = v2f{xs[k].x, xs[k].y} + d[2 * k];
                            d[2 * k + 1] = v2f{xs[k].z, xs[k].w} + d[2 * k + 1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            const uint32_t r = 16 * b + 4 * k;
                            if (!UPDATER) wr[k] = v4f{w[UPDATER ? 0 : r], w[UPDATER ? 0 : r + 1], w[UPDATER ? 0 : r + 2], w[UPDATER ? 0 : r + 3]};
                            d[2 * k] = d[2 * k] * v2f{wr[k].x, wr[k].y};
                            d[2 * k + 1] = d[2 * k + 1] * v2f{wr[k].z, wr[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) { acc += d[k].x; acc += d[k].y; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (plain) {
                    // A lone wavefront per SIMD hides no latency by itself: the products of a batch are formed stage by stage
                    // (8 independent packed instructions back to back, the scheduler held to that order), then its 16 adds --
                    // measured 900-1000 clocks per 64 rows against 1600-1800 with product and add interleaved row by row
                    // (profiles/experiments/chain_turn_probe.hip).
                    const v2f vq2 = {vq, vq}, gq2 = {gq, gq};
                    // (the variant for networks with cells has no registers to spare for the next batch's voltages: it would spill
                    // weights to scratch inside this loop, 3 700 instead of 1 300 clocks per turn)
                    constexpr bool AHEAD = !CELLS || UPDATER || REGISTERS;
                    v4f vp[4], wr[4], vp_next[4], wr_next[4];
                    load_batch(0, vp, wr);
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        if (AHEAD && b < 3) load_batch(b + 1, vp_next, wr_next);
                        v2f d[8];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {                   // gap_junction neuron/mod.rs:54-60 ...
                            d[2 * k] = v2f{vp[k].x, vp[k].y} - vq2;
                            d[2 * k + 1] = v2f{vp[k].z, vp[k].w} - vq2;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) d[k] = gq2 * d[k];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {                   // ... times the weight
                            const uint32_t r = 16 * b + 4 * k;
                            if (!UPDATER) wr[k] = v4f{w[UPDATER ? 0 : r], w[UPDATER ? 0 : r + 1], w[UPDATER ? 0 : r + 2], w[UPDATER ? 0 : r + 3]};
                            d[2 * k] = d[2 * k] * v2f{wr[k].x, wr[k].y};
                            d[2 * k + 1] = d[2 * k + 1] * v2f{wr[k].z, wr[k].w};
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) { acc += d[k].x; acc += d[k].y; }
                        __builtin_amdgcn_sched_barrier(0);
                        if (AHEAD) {
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) { vp[k] = vp_next[k]; wr[k] = wr_next[k]; }
                        } else if (b < 3) {
                            load_batch(b + 1, vp, wr);
                        }
                    }
                } else if (STDP) {
                    // (as below, but the matrix in memory is stale while the run updates its weights: the resident weights and this
                    // lane's existence bits instead)
#pragma unroll
                    for (uint32_t g = 0; g < 16; ++g) {
                        if (row0 + 4 * g >= n_tot) continue;
                        const v4f x = pre[g];
                        const float e[4] = {x.x, x.y, x.z, x.w};
                        float ww[4];
                        if (UPDATER) { const v4f y = sh.w0[g][lane]; ww[0] = y.x; ww[1] = y.y; ww[2] = y.z; ww[3] = y.w; }
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            const float wj = UPDATER ? ww[j] : w[UPDATER ? 0 : 4 * g + j];
                            const bool edge = ((4 * g + j < 32u ? ex_lo >> (4 * g + j) : ex_hi >> (4 * g + j - 32u)) & 1u) != 0u;
                            const float p = term_of(KIND_NEURON, e[j]) * wj;
                            acc += edge ? p : 0.0f;
                        }
                    }
                } else {
                    // some voltage is huge, infinite or NaN: absent edges are skipped explicitly -- the weights come from the
                    // matrix again (cache resident), where an absent edge is the NaN sentinel
#pragma unroll 1
                    for (uint32_t g = 0; g < 16 && row0 + 4 * g < n_tot; ++g) {
                        const v4f x = pre[g], y = units[(size_t)g * in.ld];
                        const uint4 kd = CELLS ? kinds[g] : make_uint4(KIND_NEURON, KIND_NEURON, KIND_NEURON, KIND_NEURON);
                        const float e[4] = {x.x, x.y, x.z, x.w}, ww[4] = {y.x, y.y, y.z, y.w};
                        const uint32_t kk[4] = {kd.x, kd.y, kd.z, kd.w};
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            const float p = term_of(kk[j], e[j]) * ww[j];
                            acc += (ww[j] == ww[j]) ? p : 0.0f;
                        }
                    }
                }
                if (last_of_chunk) sh.pi[chunk][lane] = acc;
                else sh.hand[chunk_local][lane] = acc;
              }
              if (CHEM) {
                // weighted transmitter concentrations (weight_neurotransmitter_concentration, iterate_and_spike/mod.rs:2837-2866): per live
                // type the same ascending chain over this wavefront's rows, t * w, the running sum handed on through LDS
#pragma unroll 1
                for (uint32_t jj = 0; jj < (STAGGER ? 1u : n_live); ++jj) {
                    // the live type this wavefront sums in this slot (wraps when it is none); a lent wavefront: its own type, in turn
                    const uint32_t j = LEND ? ((shadow && t == turn) ? chunk_local - 1u : 0xFFFFFFFFu) : (STAGGER ? t - turn - chem_off : jj);
                    if (STAGGER ? j >= n_live : t != turn) continue;
                    const uint32_t cl = shadow ? 0u : chunk_local;            // (a lent wavefront sums chunk 0)
                    float acc_t = (turn != 0) ? sh.hand_t[j][cl][lane] : 0.0f;
                    uint32_t zero;
                    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
                    const v4f *tp = reinterpret_cast<const v4f *>(sh.t[j] + (row0 - group_row0) + zero);
                    if (plain_t) {
                        // (batches of 8 rows where the wavefront also carries spike-train cells: the registers are the weights')
                        constexpr uint32_t Q = (CELLS && !UPDATER) ? 2u : 4u;          // 16-byte units (4 rows each) per batch
#pragma unroll
                        for (uint32_t b = 0; b < 16u / Q; ++b) {
                            v4f x[Q], wr[Q];
#pragma unroll
                            for (uint32_t k = 0; k < Q; ++k) {
                                x[k] = tp[Q * b + k];
                                if (UPDATER) wr[k] = sh.w0[Q * b + k][lane];
                            }
                            v2f d[2 * Q];
#pragma unroll
                            for (uint32_t k = 0; k < Q; ++k) {
                                const uint32_t r = 4 * Q * b + 4 * k;
                                if (!UPDATER) wr[k] = v4f{w[UPDATER ? 0 : r], w[UPDATER ? 0 : r + 1], w[UPDATER ? 0 : r + 2], w[UPDATER ? 0 : r + 3]};
                                d[2 * k] = v2f{x[k].x, x[k].y} * v2f{wr[k].x, wr[k].y};
                                d[2 * k + 1] = v2f{x[k].z, x[k].w} * v2f{wr[k].z, wr[k].w};
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (uint32_t k = 0; k < 2 * Q; ++k) { acc_t += d[k].x; acc_t += d[k].y; }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        // some value is not a small finite number: terms are skipped explicitly -- weights from the matrix again
                        // (absent edge = NaN), the presence of the type from the flag planes
                        const uint32_t type = a.live_type[j];
#pragma unroll 1
                        for (uint32_t g = 0; g < 16 && row0 + 4 * g < n_tot; ++g) {
                            const v4f x = tp[g], y = units[(size_t)g * in.ld];
                            const float e[4] = {x.x, x.y, x.z, x.w}, ww[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                            for (uint32_t jj = 0; jj < 4; ++jj) {
                                const uint32_t p = row0 + 4 * g + jj;
                                const bool has = p < n_neurons && in.nt_flags[(size_t)type * in.n_pad + p] != 0u;
                                const float pr = e[jj] * ww[jj];
                                acc_t += (has && ww[jj] == ww[jj]) ? pr : 0.0f;
                            }
                        }
                    }
                    if (last_of_chunk) sh.pt[a.live_type[j]][cl][lane] = acc_t;
                    else sh.hand_t[j][cl][lane] = acc_t;
                }
              }
            }
            __syncthreads();
        }
        // (2a) this thread's cell advances (PoissonNeuron::iterate spike_train/mod.rs:411-435, RateSpikeTrain::iterate :1016-1031,
        // as k_spike_trains): spike, voltage, firing time, and the gap-junction value the NEXT step's inputs read
        if (is_cell) {
            const float par = sh.cell_par[0][tid], c_v_th = sh.cell_par[1][tid], c_v_resting = sh.cell_par[2][tid];
            const float c_dt = sh.cell_par[3][tid], c_k = sh.cell_par[4][tid];
            uint32_t spike;
            if (a.st_kind == 1) {
                const uint32_t seed = xorshift32(sh.cell_word[tid]);
                sh.cell_word[tid] = seed;
                const float random_number = (float)seed / 4294967296.0f;   // (float) seed / 0xFFFFFFFF
                spike = random_number < par;
            } else {
                float step = __uint_as_float(sh.cell_word[tid]) + c_dt;
                spike = (par != 0.0f) && (step >= par);
                if (spike) step = 0.0f;
                sh.cell_word[tid] = __float_as_uint(step);
            }
            const float cv = spike ? c_v_th : c_v_resting;
            sh.cell_v[tid] = cv;
            sh.cell_spiking[tid] = spike;
            int32_t lft = sh.cell_lft[tid];
            if (spike) {
                const long long clock0 = (long long)(((unsigned long long)sh.cell_clock_hi[tid] << 32) | sh.cell_clock_lo[tid]);
                lft = (int32_t)(clock0 + (long long)s);
                sh.cell_lft[tid] = lft;
            }
            if (writes_cells && a.st_vhist_row) a.st_vhist_row[(size_t)s * a.st_vhist_stride + cell] = cv;
            const long long view_clock = a.view_clock0 + (long long)s + 1;
            sh.cell_presyn[tid] = lft < 0 ? c_v_resting
                                          : (sh.cell_refr[tid] ? exponential_decay_effect<true>(view_clock, lft, c_v_th, c_v_resting, c_k, c_dt)
                                                               : delta_dirac_effect<true>(view_clock, lft, c_v_th, c_v_resting, c_k, c_dt));
        }
        // (2b) several row groups per tile: the groups other than 0 publish their chunk sums, group 0 collects them -- wavefront
        // 1 + j takes remote chunk j -- behind one more barrier of its own
        if (n_groups > 1u) {
            const uint32_t ptag = tag_base + s + 1u;
            unsigned long long *pslot = a.partials + ((size_t)(s & 1u) * n_tiles + tile) * (RUN_RESIDENT_MAX_GROUPS - 1u) * 256u;
            if (!updates) {
                if (UPDATER) {
#pragma unroll 1
                    for (uint32_t c = 0; c < 4u && 4u * group + c < n_chunks; ++c) {
                        const unsigned long long x = ((unsigned long long)ptag << 32) | __float_as_uint(sh.pi[4u * group + c][lane]);
                        __hip_atomic_store(pslot + (group - 1u) * 256u + c * 64u + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                const uint32_t j = wave - 1u;                      // remote chunk: group 1 + j / 4, its chunk j % 4
                if (wave >= 1u && j < 4u * (n_groups - 1u) && 4u + j < n_chunks) {
                    unsigned long long x;
                    uint32_t spins = 0;
                    do {
                        x = __hip_atomic_load(pslot + j * 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((uint32_t)(x >> 32) != ptag && ++spins < a.spin_limit);
                    if ((uint32_t)(x >> 32) != ptag) sh.gave_up = 1u;
                    sh.pi[4u + j][lane] = __uint_as_float((uint32_t)x);
                }
                __syncthreads();
            }
        }
        lap(2);

        // (3) wavefront 0 of group 0: second level of the sum, the neuron update of these 64 columns, their granules
        if (UPDATER && updates) {
            uint32_t spike = 0;
            float v_new = 0.0f;
            float t_new[K_TYPES] = {0.0f, 0.0f, 0.0f};       // CHEM: by TYPE, what the neuron releases after this step (0: not its type)
            if (in_registers) {
                const ResidentRunArgs &b = a;
                float *vhist_row = b.up.vhist_row ? b.up.vhist_row + (size_t)s * b.vhist_stride : nullptr;
                if (col) {
                    float sum = 0.0f;
                    for (uint32_t c = 0; c < n_chunks; ++c) sum += sh.pi[c][lane];
                    const float i_in = sum / n_div;
                    if (MODEL == 0) {            // Izhikevich
                        const float dv = (0.04f * (nv * nv) + 5.0f * nv + 140.0f - n2 + (CHEM && !b.up.electrical ? 0.0f : i_in)) * pr[0];
                        const float dw = (pr[2] * (pr[3] * nv - n2)) * pr[1];
                        if (CHEM) {
                            // receptor kinetics from this step's transmitter input, currents at the old voltage, their sum
                            float total = 0.0f;
#pragma unroll
                            for (int k = 0; k < K_TYPES; ++k) {
                                const uint32_t c_cnt = sh.chem_cnt[k][lane];
                                const bool on = (c_flags >> k & 1u) != 0u, fed = on && c_cnt != 0u;
                                float c_r = sh.chem_state[3 * k][lane];
                                if (fed) {
                                    float st = 0.0f;
                                    for (uint32_t c = 0; c < n_chunks; ++c) st += sh.pt[k][c][lane];
                                    c_r = rc_apply(b.up.rc_kind, c_r, st / (float)c_cnt, sh.chem_par[8 * k][lane], sh.chem_par[8 * k + 1][lane], c_dt);
                                    sh.chem_state[3 * k][lane] = c_r;
                                }
                                if (on) {
                                    const float c_g = sh.chem_par[8 * k + 2][lane], c_e = sh.chem_par[8 * k + 3][lane];
                                    const float c_cur = k == 1 ? ((1.0f / (1.0f + ((expf_glibc(-0.062f * nv) * sh.chem_par[24][lane]) / 3.75f)) * c_g) * c_r) * (nv - c_e)
                                                               : (c_g * c_r) * (nv - c_e);
                                    sh.chem_state[3 * k + 1][lane] = c_cur;
                                    total += c_cur;
                                }
                            }
                            const float neurotransmitter_dv = -(total * pr[0]);
                            v_new = nv + (dv + neurotransmitter_dv);
                            // the neuron's own release: previous spike flag, the voltage before the reset
#pragma unroll
                            for (int k = 0; k < K_TYPES; ++k) {
                                t_new[k] = 0.0f;
                                if (c_flags >> (8 + k) & 1u) {
                                    t_new[k] = nt_apply(b.up.nt_kind, sh.chem_state[3 * k + 2][lane], sh.chem_par[8 * k + 4][lane], sh.chem_par[8 * k + 5][lane],
                                                        sh.chem_par[8 * k + 6][lane], sh.chem_par[8 * k + 7][lane], v_new, last_spike, c_dt);
                                    sh.chem_state[3 * k + 2][lane] = t_new[k];
                                }
                            }
                        } else {
                            v_new = nv + dv;
                        }
                        float w_new = n2 + dw;
                        if (v_new >= pr[6]) {
                            spike = 1;
                            v_new = pr[4];
                            w_new += pr[5];
                        }
                        n2 = w_new;
                    } else if (MODEL == 4) {     // simple leaky integrate-and-fire
                        const float dv = (pr[0] * (nv - pr[1]) + i_in) * pr[2];
                        v_new = nv + dv;
                        if (v_new >= pr[6]) {
                            spike = 1;
                            v_new = pr[5];
                        }
                    } else {                     // leaky (1) / quadratic (3) integrate-and-fire + handle_spiking :87-102
                        const float dv = MODEL == 1 ? ((pr[0] * (nv - pr[1])) + (pr[2] * (i_in / pr[3]))) * pr[4]
                                                    : ((pr[0] * (nv - pr[1]) * (nv - pr[2])) + pr[3] * i_in) * pr[4];
                        v_new = nv + dv;
                        const float v_reset = MODEL == 1 ? pr[5] : pr[1];
                        if (n2 > 0.0f) {
                            v_new = v_reset;
                            n2 -= 1.0f;
                        } else if (v_new >= pr[6]) {
                            spike = 1;
                            v_new = v_reset;
                            n2 = pr[7];
                        }
                    }
                    nv = v_new;
                    last_spike = spike;
                    const uint32_t q = b.up.rows.global_of(ql);
                    if (spike) {
                        b.up.n.last_firing_time[q] = (int32_t)(b.up.clock + s);
                        n_spikes += 1;
                    }
                    if (vhist_row) vhist_row[q] = v_new;
                }
            } else {
                const ResidentRunArgs &b = a;
                float *vhist_row = b.up.vhist_row ? b.up.vhist_row + (size_t)s * b.vhist_stride : nullptr;
                if (col)
                    spike = update_neuron_at<MODEL, LdsSums, false, CHEM>(b.up, ql, LdsSums{sh.pi, CHEM ? sh.pt : nullptr, n_chunks, lane}, b.up.clock + s, vhist_row,
                                                    &v_new, CHEM ? t_new : nullptr);
            }
            if (a.up.spike_row) {
                unsigned long long *spike_row = a.up.spike_row + (size_t)s * a.raster_stride;
                const unsigned long long word = __ballot(spike != 0);
                if (lane == 0) spike_row[(a.up.q0 + ql) >> 6] = word;
            }
            // CHEM: a type the neuron does not release travels as 0 (t_new stays 0 where neuron_nt_update skipped the type; a
            // neuron without the flag may still HOLD a concentration, which nobody reads)
            float t_out[K_TYPES] = {0.0f, 0.0f, 0.0f};       // by live slot
            if (CHEM) {
#pragma unroll
                for (uint32_t j = 0; j < K_TYPES; ++j)
                    if (j < n_live) {
                        const uint32_t type = a.live_type[j];
                        const float tt = type == 0u ? t_new[0] : (type == 1u ? t_new[1] : t_new[2]);
                        t_out[j] = (my_nt_mask >> type & 1u) ? tt : 0.0f;
                    }
            }
            if (STDP) {
                // the plastic columns of this tile that spiked now: next step's column updates (read behind its first barrier)
                const unsigned long long cm = __ballot(spike != 0u && col_plastic);
                if (lane == 0) { sh.stdp_colmask[0] = (uint32_t)cm; sh.stdp_colmask[1] = (uint32_t)(cm >> 32); }
                spk_mine = spike;
            }
            if (alone) {
                v_mine = col ? v_new : 0.0f;
                if (col) sh.v[ql] = v_new;                    // read behind the next step's first barrier
                if (CHEM) {
#pragma unroll
                    for (uint32_t j = 0; j < K_TYPES; ++j) {
                        t_mine[j] = col ? t_out[j] : 0.0f;
                        if (col && j < n_live) sh.t[j][ql] = t_out[j];
                    }
                }
            } else if (col && s + 1 < steps && !(a.fault_step == s + 1u && blockIdx.x == 0u)) {
                unsigned long long *g = a.granules + (size_t)((s + 1) & 1u) * RUN_GRANULE_PLANES * RUN_RESIDENT_MAX_NEURONS + in.q0 + ql;
                const uint32_t tag_word = (tag_base + s + 1) | ((STDP && spike) ? 0x80000000u : 0u);
                const unsigned long long x = ((unsigned long long)tag_word << 32) | __float_as_uint(v_new);
                __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (CHEM) {
#pragma unroll
                    for (uint32_t j = 0; j < K_TYPES; ++j)
                        if (j < n_live) {
                            const unsigned long long y = ((unsigned long long)(tag_base + s + 1) << 32) | __float_as_uint(t_out[j]);
                            __hip_atomic_store(g + (size_t)(1 + j) * RUN_RESIDENT_MAX_NEURONS, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                }
            }
            lap(3);
        }
    }

    if (STDP && completed && rows_live) {
        // the weights as the run leaves them (the updates of its last step's spikes are the host's: the plain kernels), absent
        // edges as the NaN they are in W; every lane writes, padding columns included (they hold what was loaded)
        v4f *out = reinterpret_cast<v4f *>(a.w_out) + (size_t)(row0 >> 2) * in.ld + ql;
#pragma unroll
        for (uint32_t g = 0; g < 16; ++g) {
            if (row0 + 4 * g >= n_tot) continue;
            float e[4];
            if (UPDATER) { const v4f x = sh.w0[g][lane]; e[0] = x.x; e[1] = x.y; e[2] = x.z; e[3] = x.w; }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const float val = UPDATER ? e[k] : w[UPDATER ? 0 : 4 * g + k];
                const bool edge = ((4 * g + k < 32u ? ex_lo >> (4 * g + k) : ex_hi >> (4 * g + k - 32u)) & 1u) != 0u;
                e[k] = edge ? val : quiet_nan();
            }
            out[(size_t)g * in.ld] = v4f{e[0], e[1], e[2], e[3]};
        }
    }

    if (writes_cells) {                          // the cells' state this workgroup's LDS held
        const CellArrays &c = a.cells;
        if (a.st_kind == 1) c.seed[cell] = sh.cell_word[tid];
        else c.step[cell] = __uint_as_float(sh.cell_word[tid]);
        c.last_firing_time[cell] = sh.cell_lft[tid];
        c.presyn_value[cell] = sh.cell_presyn[tid];
        c.current_voltage[cell] = sh.cell_v[tid];
        c.is_spiking[cell] = sh.cell_spiking[tid];
    }
    if (UPDATER) {
        if (in_registers && updates && col) {    // the state the registers held
            const NeuronArrays &n = a.up.n;
            const uint32_t q = a.up.rows.global_of(ql);
            a.up.xout[n.xl.at(q, PLANE_V)] = nv;
            reinterpret_cast<uint32_t *>(a.up.xout)[n.xl.at(q, PLANE_SPIKE)] = last_spike;
            if (MODEL == 0) n.w_value[q] = n2;
            if (MODEL == 1 || MODEL == 3) n.refractory_count[q] = n2;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    const size_t i = (size_t)k * n.n_pad + q;
                    if (c_flags >> k & 1u) { n.rc_r[i] = sh.chem_state[3 * k][lane]; n.rc_current[i] = sh.chem_state[3 * k + 1][lane]; }
                    if (c_flags >> (8 + k) & 1u) a.up.xout[n.xl.at(q, PLANE_T0 + k)] = sh.chem_state[3 * k + 2][lane];
                }
            }
            if (a.up.spike_counts && n_spikes) a.up.spike_counts[q] += n_spikes;
        }
        if (a.timing && lane == 0)
            for (int k = 0; k < 4; ++k) a.timing[blockIdx.x * 4 + k] = spent[k];
    }
}

// Can `gridDim.x` workgroups of k_run_resident's shape (1024 threads, its LDS) run at the same time on this device right now?
// Every workgroup checks in and waits (bounded, a few tens of microseconds) for all the others.  The host asks once per handle
// and grid size before it relies on the one-launch run; a "no" (CUs masked off, the device shared) keeps one launch per step.
__global__ __launch_bounds__(1024) void k_run_resident_probe(uint32_t *counter, uint32_t *not_resident)
{
    __shared__ __attribute__((aligned(16))) ResidentRunShared footprint;
    if (threadIdx.x == 0) {
        footprint.ok[0] = 1u;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 16)) {}
        if (spins >= (1u << 16)) __hip_atomic_store(not_resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (footprint.ok[0] == 0u) *counter = 0u;   // never: keeps the LDS footprint alive
}

// LEND: networks of one chunk at most (<= 256 rows) with gap junctions AND transmitters -- see `shadow` in run_resident_steps
template <int MODEL, bool REGISTERS, bool CELLS, bool CHEM = false, bool STDP = false, bool LEND = false>
__global__ __launch_bounds__(1024) void k_run_resident(const ResidentRunArgs args)
{
    static_assert(!LEND || (CHEM && !CELLS && !STDP), "lent wavefronts: chemical + electrical synapses, neurons only");
    static_assert(!CHEM || !REGISTERS || MODEL == 0, "receptors in registers: Izhikevich only");
    static_assert(!STDP || (!CELLS && !CHEM), "weight updates inside the run: electrical synapses, neurons only");
    __shared__ __attribute__((aligned(16))) ResidentRunShared sh;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) sh.gave_up = 0u;                       // ordered before its first reader by the first step's barrier
    if (wave == 0) run_resident_steps<MODEL, true, REGISTERS, CELLS, CHEM, STDP, LEND>(args, sh, wave);
    else run_resident_steps<MODEL, false, REGISTERS, CELLS, CHEM, STDP, LEND>(args, sh, wave);
    // a poll that gave up ended the loop early everywhere in the workgroup
    if (threadIdx.x == 0) {
        bool failed = false;
        for (int k = 0; k < 16; ++k) failed = failed || sh.ok[k] == 0u;
        failed = failed || sh.gave_up != 0u;
        if (failed) __hip_atomic_store(args.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace snn

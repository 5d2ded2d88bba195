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
// consecutive presynaptic rows x 1024 consecutive postsynaptic columns.  W is stored in quad-row order
// (snn_layout.hpp): a lane owns VEC columns 256 apart and reads 4 consecutive ROWS of one of them with ONE
// global_load_dwordx4; a wavefront's load is 1 KiB contiguous, the four wavefronts' 4 KiB, and the VEC
// loads of a row group cover 16 KiB contiguous.
// blockIdx.y = row chunk, blockIdx.x (fastest in dispatch order) = column tile rotated by the chunk index: the
// workgroups in flight at any moment sweep whole matrix rows (long contiguous HBM bursts) while every XCD -- which
// receives every 8th workgroup -- touches all column groups instead of a fixed eighth of them (measured: -17 % time
// at 128x128, -2..3 % at 256x256).  The presynaptic
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
    RowMap rows;            // sparse handles: local row -> global neuron (rows.q0 = q0 for contiguous shards)
    uint32_t n_neurons;
    uint32_t n_tot;
    const float *xbuf;
    XLayout xl;
    const float *gap_conductance;
    const UniformTable *uni;          // NeuronParam slots (NP_GAP)
    // spike-train cells
    const float *st_value;
    const int32_t *st_last_firing_time;
    // sparse handles: {st_value as bits, 1 when the cell has never fired} per cell, the copy of this step (the cells may be
    // advancing into the other copy inside the same launch, k_step_csr); null: st_value / st_last_firing_time are read
    const uint2 *st_view;
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
    // chunk subset of this launch: chunk = chunk_first + blockIdx.y, skipping [hole_begin, hole_begin+hole_count)
    // (multi-GPU: the chunks of LOCAL presynaptic rows run while the all-gather of the remote state is in flight)
    uint32_t chunk_first, hole_begin, hole_count;
    // transmitter types some neuron or cell of the handle releases (launch-uniform): slot s of a kernel specialised
    // on NT live types handles type live_type[s]; the partial planes of the other types stay zero
    uint32_t live_type[K_TYPES];
    // STDP of the previous step riding on this pass (k_inputs_dense<..., STDP = true>, see StdpDeferred below)
    float *W_rw;                      // = W, writable
    const uint32_t *stdp_count;       // neurons that spiked in the previous step and whose lattice is plastic; 0: plain pass
    const uint32_t *stdp_flag;        // [n_pad] 1 for those neurons
    const float *stdp_dcol;           // [n_lattices][dcol_stride] delta of edge (p -> a post of lattice l that just spiked)
    const float *stdp_drow;           // [ld] delta of edge (a pre that just spiked -> local post r)
    const uint32_t *stdp_rowbits;     // STDP == 2: one bit per presynaptic row (neuron) that spiked in the previous step, [n_chunks][8] words
    const uint32_t *lattice_slot;     // [n_pad]
    uint32_t dcol_stride, n_lattices;
};

// Deferred STDP (dense handles): what STDP::update_weight (plasticity/mod.rs:45-66) adds to an edge in the deferred
// form depends on ONE end of the edge only once the other end is a neuron that spiked in the step
// being closed at clock t: an incoming edge p -> j of a spiking j gets delta(last_firing_time[p], t) -- a function of
// the row p and of j's lattice parameters; an outgoing edge j -> r gets delta(t, last_firing_time[r]) -- a function
// of the column r.  k_stdp_prepare evaluates both vectors at the end of step t (before the spike trains advance);
// the weight itself is rewritten by the NEXT step's input pass, which streams the whole matrix anyway: a lane that
// owns a flagged column adds dcol[row], a flagged row adds drow[column], column first then row as the standalone
// kernels do, and only changed words are stored.  The read half of the scattered column update disappears.
constexpr int STDP_MAX_LATTICES = 4;

// kind word per staged presynaptic row: bits 0..1 = 0 neuron | 1 spike train that never fired |
// 2 spike train that fired; bits 8..10 = carries neurotransmitter type k
constexpr uint32_t KIND_NEURON = 0, KIND_ST_SILENT = 1, KIND_ST_FIRED = 2;

typedef float v4f __attribute__((ext_vector_type(4)));

// Two shapes of the same kernel (identical arithmetic, chosen by the size of W):
//  STREAM  (W larger than the caches): 256 threads, 4 adjacent columns per lane, one
//          global_load_dwordx4 ... nt per wave-row -- W is read exactly once per step, so it is streamed
//          past the caches.  Measured on MI355X at 256x256 (same box, A/B): nt 6.18 TB/s vs default cache
//          policy 5.75 TB/s.  Rows in batches of 8 loads; occupancy covers the HBM latency.
//  !STREAM (W stays resident in the L2s / Infinity Cache between steps, i.e. small lattices): one
//          wavefront per workgroup, ONE column per lane, default cache policy, 32 rows per round trip.
//          A 32x32 lattice then spreads over 64 CUs instead of 4 -- each CU's vector-L1 bandwidth
//          (64 B/clk) is what bounds a cache-resident pass, not HBM.
//  STREAM == 2: the streaming shape with TWO columns per lane (512-B wave-rows): twice the wavefronts for
//          matrices too small to fill the chip with the 4-column shape (e.g. 128x128: 4096 -> 8192 waves).
template <int STREAM> struct InputsShape {
    static constexpr int VEC = STREAM == 1 ? 4 : (STREAM == 2 ? 2 : 1);      // columns per lane, THREADS apart
    static constexpr int THREADS = STREAM ? 256 : 64;
    static constexpr int TILE = VEC * THREADS;            // columns per workgroup
    static constexpr uint32_t ROW_BATCH = STREAM ? 8 : 32;                   // rows in flight per buffer ...
    static constexpr uint32_t GROUP_BATCH = ROW_BATCH / 4;                   // ... = row groups (units per column)
};

__device__ __forceinline__ float acc_if_edge(float acc, float term, float w)
{
    // `w == w` is false exactly for the NaN sentinel of an absent edge
    return (w == w) ? acc + term * w : acc;
}

// NT: number of live transmitter types the launch is specialised on (1..3; only read when CHEM).  A network whose
// cells all release one type (BASELINE configs[2]: AMPA) then carries VEC accumulators for it instead of 3 * VEC, no
// per-row tests of the other types, and the two-register-buffer sweep of the electrical pass.
// STDP: 0 the plain pass; 1 the previous step's whole STDP update rides on it (columns and rows, "defer_stdp" 1); 2 its ROW half
// only ("defer_stdp" 3: the outgoing edges of the neurons that spiked; the incoming edges were scattered by k_stdp_columns when
// the step closed).  The row half costs the pass no fetch and only full-line stores: a flagged row group is rewritten by every
// lane of the wavefront, 1 KiB contiguous per store instruction.
// (second launch bound of the row-half variant: the plain electrical pass runs three wavefronts per SIMD -- 138 registers -- and
// the variant must keep that occupancy: unbounded it took 174 and the pass 5.16 instead of 4.11 ms at C4)
// (the pass as a function: k_inputs_dense below, and k_inputs_dense_close -- snn_kernels_dense_step.hpp -- whose last workgroup of
// a column tile goes on to update the tile's neurons)
// AGENT_STORES: the partials leave as agent-scope relaxed atomic stores (write-through past this XCD's L2), for a reader in the
// SAME launch on another XCD (k_inputs_dense_close); the plain kernel's partials are read by the next launch and stay ordinary stores
template <bool ELEC, bool CHEM, int STREAM = 1, int NT = K_TYPES, int STDP = 0, bool AGENT_STORES = false>
__device__ __forceinline__ void inputs_dense_pass(const InputsArgs &a)
{
    using S = InputsShape<STREAM>;
    constexpr int VEC = S::VEC;
    constexpr int TS = CHEM ? NT : 1;                        // transmitter slots of this instantiation

    // 16-byte aligned: the 4 rows of a row group are read back with one ds_read_b128 per array
    __shared__ __attribute__((aligned(16))) float s_val[CHUNK];
    __shared__ __attribute__((aligned(16))) uint32_t s_kind[CHUNK];
    __shared__ __attribute__((aligned(16))) float s_t[TS][CHUNK];
    __shared__ __attribute__((aligned(16))) float s_dcol[STDP == 1 ? STDP_MAX_LATTICES : 1][STDP == 1 ? CHUNK : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_rowflag[STDP == 1 ? CHUNK : 4];
    // the previous step's weight updates ride on this pass only if some plastic neuron spiked in it (launch-uniform)
    const bool stdp_live = STDP && *a.stdp_count != 0u;

    uint32_t chunk = a.chunk_first + blockIdx.y;
    if (chunk >= a.hole_begin) chunk += a.hole_count;
    const uint32_t p0 = chunk * CHUNK;
    const uint32_t rows = min((uint32_t)CHUNK, a.n_tot - p0);
    const uint32_t tid = threadIdx.x;

    // Column tile of this workgroup, rotated by the chunk index: workgroups are dealt round-robin over the 8
    // XCDs, so with a power-of-two tile count an unrotated mapping would pin every XCD (and its L2 / fabric
    // ports) to the same 1/8 of the columns for the whole pass.
    constexpr uint32_t GB = S::GROUP_BATCH;
    const uint32_t tile = (blockIdx.x + blockIdx.y) % gridDim.x;
    const uint32_t ql = tile * S::TILE + tid;                // the lane's column j is ql + j * THREADS
    const size_t ld = a.ld;
    const uint32_t groups = (rows + 3u) >> 2;                // row groups of the chunk (rows are padded to 4 with NaN)
    // Addresses = a wave-uniform base (row group, tile, column slot j) + the lane's fixed 16-byte index: no per-lane
    // 64-bit arithmetic in the loop.  A column of the last tile past the shard's width reads into the next row group
    // (or, for the last group, into the slack the matrix is allocated with: WMATRIX_SLACK); the value is never used.
    const v4f *ubase = reinterpret_cast<const v4f *>(a.W) + (size_t)(p0 >> 2) * ld + (size_t)tile * S::TILE;
    bool colv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) colv[j] = ql + (uint32_t)j * S::THREADS < a.n_loc;
    auto load_units = [&](uint32_t grp, v4f (&w)[VEC]) {
        const v4f *gp = ubase + (size_t)grp * ld;
#pragma unroll
        for (int j = 0; j < VEC; ++j) w[j] = __builtin_nontemporal_load(gp + j * S::THREADS + tid);
    };

    // The first batch of row groups is requested BEFORE the presynaptic values are staged: all workgroups of a
    // launch start together, and without this the whole chip would leave HBM idle for the staging round trip.
    v4f pre[GB][VEC];
    const bool have_pre = groups >= GB && ql < a.n_loc;
    if (have_pre) {
#pragma unroll
        for (uint32_t u = 0; u < GB; ++u) load_units(u, pre[u]);
    }

    // ---- stage the chunk's presynaptic values in LDS (coalesced reads, one pass per array) ----
    uint32_t kinds_and = 0x703u, kinds_or = 0u;              // over the rows this thread stages
    for (uint32_t i = tid; i < rows; i += S::THREADS) {
        const uint32_t p = p0 + i;
        float val;
        uint32_t kind;
        if (p < a.n_neurons) {
            val = a.xbuf[a.xl.at(p, PLANE_V)];
            kind = KIND_NEURON;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < TS; ++k) {
                    const uint32_t ty = a.live_type[k];
                    const uint32_t f = a.nt_flags[(size_t)ty * a.n_pad + p];
                    kind |= f ? (0x100u << k) : 0u;
                    s_t[k][i] = a.xbuf[a.xl.at(p, PLANE_T0 + ty)];
                }
            }
        } else {
            const uint32_t s = p - a.n_neurons;
            val = a.st_value[s];
            kind = (a.st_last_firing_time[s] < 0) ? KIND_ST_SILENT : KIND_ST_FIRED;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < TS; ++k) {
                    const uint32_t ty = a.live_type[k];
                    const uint32_t f = a.st_nt_flags[(size_t)ty * a.c_pad + s];
                    kind |= f ? (0x100u << k) : 0u;
                    s_t[k][i] = a.st_nt_t[(size_t)ty * a.c_pad + s];
                }
            }
        }
        if (STDP == 1 && stdp_live) {
            s_rowflag[i] = (p < a.n_neurons) ? a.stdp_flag[p] : 0u;
#pragma unroll
            for (int l = 0; l < STDP_MAX_LATTICES; ++l)
                if ((uint32_t)l < a.n_lattices) s_dcol[l][i] = a.stdp_dcol[(size_t)l * a.dcol_stride + p];
        }
        s_val[i] = val;
        s_kind[i] = kind;
        kinds_and &= kind | ~0x703u;
        kinds_or |= kind;
    }
    // rows of the chunk's last group past n_tot are padding (their weights are the absent-edge NaN): harmless values
    for (uint32_t i = rows + tid; i < groups * 4; i += S::THREADS) {
        s_val[i] = 0.0f;
        s_kind[i] = KIND_NEURON;
        if (STDP == 1) s_rowflag[i] = 0u;
    }
    // Homogeneous chunks (every row a neuron; each transmitter type carried by all rows or by none -- any lattice
    // populated from one base neuron) take a row body without per-row kind tests: workgroup-uniform votes, which
    // also are the barrier that publishes the staged values.
    bool uniform_chunk = false, type_on[TS];
    if (CHEM) {
        int ok = __syncthreads_and((kinds_or & 3u) == 0u);
#pragma unroll
        for (int k = 0; k < TS; ++k) {
            const int all_k = __syncthreads_and((kinds_and >> (8 + k)) & 1u);
            const int any_k = __syncthreads_or((kinds_or >> (8 + k)) & 1u);
            type_on[k] = all_k != 0;
            ok = ok && (all_k || !any_k);
        }
        // a launch specialised on ONE live type tests nothing per row: its uniform chunks are those that carry the type
        uniform_chunk = ok != 0 && (NT > 1 || type_on[0]);
    } else {
        type_on[0] = false;
        __syncthreads();
    }

    if (ql >= a.n_loc) return;                               // padding columns (n_loc .. ld) carry no neuron

    // ---- this lane's postsynaptic voltage / conductance, kept in registers ----
    float vq[VEC], gq[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t q = ql + (uint32_t)j * S::THREADS;
        if (ELEC && q < a.n_loc) {
            vq[j] = a.xbuf[a.xl.at(a.q0 + q, PLANE_V)];
            gq[j] = a.gap_conductance[a.q0 + q];
        } else {
            vq[j] = 0.0f;
            gq[j] = 0.0f;
        }
    }

    // deferred STDP: this lane's columns that spiked in the previous step, their lattice and the row-update deltas
    bool cf[VEC];
    uint32_t clat[VEC];
    float dr[VEC];
    bool wave_cols = false;
    uint32_t wave_lat = 0;        // lattice of the wave's flagged columns when they share one (the usual case), else ~0
    // STDP == 2: the chunk's 256 row bits in 8 scalars (wave-uniform loads), the row-update delta of this lane's columns
    uint32_t rowbits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool rows_live = false;
    if (STDP == 2 && stdp_live) {
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rowbits[i] = a.stdp_rowbits[(size_t)chunk * 8 + i];
            any |= rowbits[i];
        }
        rows_live = any != 0u;                                 // chunk-uniform: most chunks have no spiking row at all
        if (rows_live) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const uint32_t q = ql + (uint32_t)j * S::THREADS;
                dr[j] = q < a.n_loc ? a.stdp_drow[q] : 0.0f;
            }
        }
    }
    if (STDP == 1 && stdp_live) {
        uint32_t lat_mask = 0;    // lattices among this lane's flagged columns
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t q = ql + (uint32_t)j * S::THREADS;
            const bool in = q < a.n_loc;
            cf[j] = in && a.stdp_flag[a.q0 + q] != 0u;
            clat[j] = in ? a.lattice_slot[a.q0 + q] : 0u;
            dr[j] = in ? a.stdp_drow[q] : 0.0f;
            lat_mask |= cf[j] ? (1u << clat[j]) : 0u;
        }
        uint32_t wave_mask = 0;
#pragma unroll
        for (int l = 0; l < STDP_MAX_LATTICES; ++l) wave_mask |= __any((lat_mask >> l) & 1u) ? (1u << l) : 0u;
        wave_cols = wave_mask != 0u;
        wave_lat = (wave_mask & (wave_mask - 1u)) ? 0xFFFFFFFFu : (uint32_t)__builtin_ctz(wave_mask | 0x80000000u);
    }
    // One row group of the deferred update: per row the column delta first, then the row delta (the order of the
    // standalone kernels); absent edges stay absent.  Branch-free per element: the delta of a row is a wave-uniform LDS
    // broadcast (one per lattice present), selected per lane.  Write-back in whole 128-byte lines: a lone small store
    // makes the memory side read the rest of the line before it can write it, so when any lane of a 128 B-aligned group
    // of 8 lanes (8 columns x 4 rows) changed a word, every lane of the group stores its 16-byte unit.
    auto stdp_rows_group = [&](uint32_t grp, v4f (&w)[VEC]) {
        const uint32_t bits = (rowbits[grp >> 3] >> ((grp & 7u) * 4u)) & 0xFu;      // rows 4 grp .. 4 grp + 3 of the chunk
        if (bits == 0u) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((bits >> k) & 1u)) continue;                 // wave-uniform
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float w0 = w[j][k];
                w[j][k] = (w0 == w0) ? w0 + dr[j] : w0;        // absent edges stay absent
            }
        }
        // the whole row group of this wavefront's columns goes back: full 128-byte lines, 1 KiB contiguous per instruction
        // (through the address the load used: W_rw is W)
        v4f *gp = const_cast<v4f *>(ubase) + (size_t)grp * ld;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (colv[j]) gp[j * S::THREADS + tid] = w[j];
    };
    auto stdp_group = [&](uint32_t grp, v4f (&w)[VEC]) {
        bool rs[4], any_rs = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rs[k] = __builtin_amdgcn_readfirstlane(s_rowflag[grp * 4 + k]) != 0u;
            any_rs = any_rs || rs[k];
        }
        if (!wave_cols && !any_rs) return;
        bool changed[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) changed[j] = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t r = grp * 4 + k;
            float dl[STDP_MAX_LATTICES];
            if (wave_lat != 0xFFFFFFFFu) {
                dl[0] = s_dcol[wave_lat][r];
            } else {
#pragma unroll
                for (int l = 0; l < STDP_MAX_LATTICES; ++l) dl[l] = ((uint32_t)l < a.n_lattices) ? s_dcol[l][r] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float d = dl[0];
                if (wave_lat == 0xFFFFFFFFu) {
#pragma unroll
                    for (int l = 1; l < STDP_MAX_LATTICES; ++l) d = (clat[j] == (uint32_t)l) ? dl[l] : d;
                }
                const float w0 = w[j][k];
                float wn = cf[j] ? w0 + d : w0;
                wn = rs[k] ? wn + dr[j] : wn;
                const bool ch = (w0 == w0) && __float_as_uint(wn) != __float_as_uint(w0);
                w[j][k] = ch ? wn : w0;
                changed[j] = changed[j] || ch;
            }
        }
        const uint32_t lane = tid & 63u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const unsigned long long votes = __ballot(changed[j] && colv[j]);
            if (votes && colv[j] && ((votes >> (lane & ~7u)) & 0xFFull))
                reinterpret_cast<v4f *>(a.W_rw)[(size_t)((p0 >> 2) + grp) * ld + ql + (uint32_t)j * S::THREADS] = w[j];
        }
    };

    float acc[VEC];
    float tacc[TS][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < TS; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) tacc[k][j] = 0.0f;

    // Row groups are consumed in batches of GROUP_BATCH (= ROW_BATCH rows): all loads of a batch are issued before the
    // first use, so every wave keeps ROW_BATCH rows of reads per column in flight regardless of the branches in the
    // body.  `body(row, w[VEC])` sees one presynaptic row at a time, as before the quad-row layout.
    auto sweep = [&](auto body) {
        auto run_group = [&](uint32_t grp, v4f (&wg)[VEC]) {
            if (STDP == 1 && stdp_live) stdp_group(grp, wg);
            if (STDP == 2 && rows_live) stdp_rows_group(grp, wg);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) w[j] = wg[j][k];
                body(grp * 4 + k, w);
            }
        };
        uint32_t g = 0;
        if constexpr ((STREAM == 1 && (!CHEM || NT == 1)) || STREAM == 2 || STREAM == 0) {
            // Two register buffers, so the next batch is already in flight while the current one is consumed: +1 % over
            // a single buffer for the electrical-only 4-column pass at 256x256 (its chemical variants with 2-3 live
            // types keep one buffer -- they need the registers for their extra accumulators), -5 % time at 96x96 for
            // the 2-column shape, -24 % at 64x64 for the one-column cache-resident shape (2 x 32 rows).
            constexpr uint32_t B = GB;
            if (groups >= 2 * B) {
                v4f (&wa)[B][VEC] = pre;          // groups 0 .. B-1, requested before the staging phase
                v4f wb[B][VEC];
                for (; g + 3 * B <= groups; g += 2 * B) {
#pragma unroll
                    for (uint32_t u = 0; u < B; ++u) load_units(g + B + u, wb[u]);
#pragma unroll
                    for (uint32_t u = 0; u < B; ++u) run_group(g + u, wa[u]);
#pragma unroll
                    for (uint32_t u = 0; u < B; ++u) load_units(g + 2 * B + u, wa[u]);
#pragma unroll
                    for (uint32_t u = 0; u < B; ++u) run_group(g + B + u, wb[u]);
                }
                // wa holds groups g .. g+B-1
#pragma unroll
                for (uint32_t u = 0; u < B; ++u) run_group(g + u, wa[u]);
                g += B;
            }
        }
        if (g == 0 && have_pre) {
#pragma unroll
            for (uint32_t u = 0; u < GB; ++u) run_group(u, pre[u]);
            g = GB;
        }
        for (; g + GB <= groups; g += GB) {
            v4f wb[GB][VEC];
#pragma unroll
            for (uint32_t u = 0; u < GB; ++u) load_units(g + u, wb[u]);
#pragma unroll
            for (uint32_t u = 0; u < GB; ++u) run_group(g + u, wb[u]);
        }
        for (; g < groups; ++g) {
            v4f w[VEC];
            load_units(g, w);
            run_group(g, w);
        }
    };

    const bool plain = !CHEM && (p0 + rows <= a.n_neurons);   // workgroup-uniform
    if (plain) {
        // all presynaptic rows are neurons, electrical only: the C1/C2 inner loop
        sweep([&](uint32_t r, const float (&w)[VEC]) {
            const float vp = s_val[r];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * (vp - vq[j]), w[j]);
        });
    } else if (uniform_chunk) {
        sweep([&](uint32_t r, const float (&w)[VEC]) {
            if (ELEC) {
                const float vp = s_val[r];
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * (vp - vq[j]), w[j]);
            }
#pragma unroll
            for (int k = 0; k < TS; ++k) {
                if (NT == 1 || type_on[k]) {
                    const float t = s_t[k][r];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) tacc[k][j] = acc_if_edge(tacc[k][j], t, w[j]);
                }
            }
        });
    } else {
        sweep([&](uint32_t r, const float (&w)[VEC]) {
            const uint32_t kind = __builtin_amdgcn_readfirstlane(s_kind[r]);
            if (ELEC) {
                const float vp = s_val[r];
                const uint32_t src = kind & 3u;
                if (src == KIND_NEURON) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * (vp - vq[j]), w[j]);
                } else if (src == KIND_ST_SILENT) {
                    // never fired: v_resting without the conductance factor (neuron/mod.rs:126-128)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], vp, w[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * vp, w[j]);
                }
            }
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < TS; ++k) {
                    if (kind & (0x100u << k)) {
                        const float t = s_t[k][r];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) tacc[k][j] = acc_if_edge(tacc[k][j], t, w[j]);
                    }
                }
            }
        });
    }

    auto put = [](float *p, float x) {
        if (AGENT_STORES) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = x;
    };
    if (ELEC) {
        float *dst = a.part_i + (size_t)chunk * a.ld + ql;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (colv[j]) put(dst + (uint32_t)j * S::THREADS, acc[j]);
    }
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < TS; ++k) {
            float *dst = a.part_t + ((size_t)a.live_type[k] * a.n_chunks + chunk) * a.ld + ql;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (colv[j]) put(dst + (uint32_t)j * S::THREADS, tacc[k][j]);
        }
    }
}

template <bool ELEC, bool CHEM, int STREAM = 1, int NT = K_TYPES, int STDP = 0>
__global__ __launch_bounds__(InputsShape<STREAM>::THREADS, (STDP == 2 && STREAM == 1 && !CHEM) ? 3 : 1) void k_inputs_dense(const InputsArgs a)
{
    inputs_dense_pass<ELEC, CHEM, STREAM, NT, STDP>(a);
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
        const float w = a.W[widx(p, q, a.ld)];
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

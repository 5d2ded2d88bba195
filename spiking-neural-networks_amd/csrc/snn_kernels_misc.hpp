// Everything around the two hot kernels: graph import/export (NaN-sentinel fusion of the
// reference's weights f32[N^2] + connections u32[N^2], graph/mod.rs:310-333, 729-770), the
// device-side synthetic graph, spike-train cells (neuron/spike_train/mod.rs:411-435, 1016-1031;
// SpikeTrainLattice::iterate neuron/mod.rs:1377-1393), wave-ballot spike compaction and the STDP
// column/row updates (neuron/plasticity/mod.rs:45-66 driven as neuron/mod.rs:2308-2417, 2573-2576).
#pragma once
#include "snn_layout.hpp"
#include "snn_math.hpp"
#include "snn_custom_model.hpp"

namespace snn {

__device__ __forceinline__ float quiet_nan() { return __int_as_float(0x7FC00000); }

// ---- generic fills -------------------------------------------------------------------------
__global__ void k_fill_f32(float *p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// Snapshot / restore of a list of small arrays in one launch (the one-launch run keeps the state it started from until
// it is known to have finished): entry = blockIdx.y, `restore` copies dst -> src.
struct CopyEntry {
    uint32_t *src, *dst;
    uint32_t words, pad;
};
__global__ __launch_bounds__(256) void k_copy_table(const CopyEntry *table, int restore)
{
    const CopyEntry e = table[blockIdx.y];
    const uint32_t *from = restore ? e.dst : e.src;
    uint32_t *to = restore ? e.src : e.dst;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < e.words; i += gridDim.x * 256u) to[i] = from[i];
}
// Self-check of the stepper (option "verify", tests only): the table's arrays copied to a SECOND buffer laid out like the
// snapshot (dst - base + alt), and compared with it word for word after the same steps have been taken again from the same
// snapshot.  Words that are NaN as binary32 on both sides count as equal (their payload is not part of the contract); entries
// with pad != 0 hold what legitimately depends on the order of atomics (the compacted spike list) and are skipped.
// report: [0] differing words, [1] entry + 1 of the lowest differing entry seen, [2] its word, [3] / [4] the two values.
__global__ __launch_bounds__(256) void k_copy_table_alt(const CopyEntry *table, const uint32_t *base, uint32_t *alt, int restore)
{
    const CopyEntry e = table[blockIdx.y];
    uint32_t *side = alt + (e.dst - base);
    const uint32_t *from = restore ? side : e.src;
    uint32_t *to = restore ? e.src : side;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < e.words; i += gridDim.x * 256u) to[i] = from[i];
}
// entries (1-based) a comparison leaves out: scratch the step forms use differently, caches whose validity flags differ
struct SkipSet { uint32_t n, entry[8]; };
__global__ __launch_bounds__(256) void k_compare_table_alt(const CopyEntry *table, const uint32_t *base, const uint32_t *alt, uint32_t *report,
                                                           SkipSet skip)
{
    const CopyEntry e = table[blockIdx.y];
    if (e.pad) return;
    for (uint32_t k = 0; k < skip.n; ++k)
        if (skip.entry[k] == blockIdx.y + 1u) return;
    const uint32_t *was = alt + (e.dst - base);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < e.words; i += gridDim.x * 256u) {
        const uint32_t x = was[i], y = e.src[i];
        const bool both_nan = (x & 0x7FFFFFFFu) > 0x7F800000u && (y & 0x7FFFFFFFu) > 0x7F800000u;
        if (x == y || both_nan) continue;
        if (atomicAdd(&report[0], 1u) == 0u) {
            report[1] = blockIdx.y + 1u; report[2] = i; report[3] = x; report[4] = y;
        }
    }
}
// the same comparison of two plain arrays (the matrices of a run with weight updates); `entry` names the array in the report
__global__ __launch_bounds__(256) void k_compare_words(const uint32_t *was, const uint32_t *now, size_t words, uint32_t entry, uint32_t *report)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u) {
        const uint32_t x = was[i], y = now[i];
        const bool both_nan = (x & 0x7FFFFFFFu) > 0x7F800000u && (y & 0x7FFFFFFFu) > 0x7F800000u;
        if (x == y || both_nan) continue;
        if (atomicAdd(&report[0], 1u) == 0u) {
            report[1] = entry; report[2] = (uint32_t)i; report[3] = x; report[4] = y;
        }
    }
}
// test hook of "verify": one bit of one word flipped between the two passes' outcome and the comparison
__global__ void k_flip_bit(uint32_t *p, size_t word) { p[word] ^= 1u; }
__global__ void k_iota_u32(uint32_t *p, size_t n, uint32_t first)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = first + (uint32_t)i;
}

// ---- graph import / export ----------------------------------------------------------------
// src rows are `src_ld` wide and hold GLOBAL postsynaptic columns; the handle keeps columns
// [q0, q0+n_loc).  Rows [row0, row0+rows) of W are written; padding columns get the sentinel.
// A connected edge whose weight is NaN cannot be stored (NaN IS the absent-edge sentinel; the reference's Some(NaN),
// graph/mod.rs:204-213, would be an edge that poisons its postsynaptic neuron): bad[0] counts them, bad[1] / bad[2] name one.
__global__ void k_graph_import(float *W, uint32_t ld, uint32_t n_loc, uint32_t q0, uint32_t row0, uint32_t rows,
                               const float *src_w, const uint32_t *src_c, size_t src_ld, uint32_t *bad)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (q >= ld || r >= rows) return;
    float out = quiet_nan();
    if (q < n_loc) {
        const size_t i = (size_t)r * src_ld + q0 + q;
        if (src_c[i] != 0) {
            out = src_w[i];
            if (out != out && atomicAdd(&bad[0], 1u) == 0u) { bad[1] = row0 + r; bad[2] = q0 + q; }
        }
    }
    W[widx(row0 + r, q, ld)] = out;
}

// inverse: absent edges export as weight 0 / connection 0 (graph/mod.rs:310-320)
__global__ void k_graph_export(const float *W, uint32_t ld, uint32_t n_loc, uint32_t q0, uint32_t row0, uint32_t rows,
                               float *dst_w, uint32_t *dst_c, size_t dst_ld)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (q >= n_loc || r >= rows) return;
    const float w = W[widx(row0 + r, q, ld)];
    const size_t i = (size_t)r * dst_ld + q0 + q;
    const bool edge = (w == w);
    dst_w[i] = edge ? w : 0.0f;
    dst_c[i] = edge ? 1u : 0u;
}

// rows [row0, row0 + rows) of a quad-row matrix <-> a row-major staging block [rows][n_loc] (trace transfers)
__global__ void k_rows_staging(float *M, uint32_t ld, uint32_t n_loc, uint32_t row0, uint32_t rows, float *staging, int to_matrix)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (q >= n_loc || r >= rows) return;
    if (to_matrix) M[widx(row0 + r, q, ld)] = staging[(size_t)r * n_loc + q];
    else staging[(size_t)r * n_loc + q] = M[widx(row0 + r, q, ld)];
}

// one step's snapshot of a lattice's internal weights (AdjacencyMatrix::update_history, graph/mod.rs:278-280): [count][count],
// absent edges as 0
__global__ void k_weight_snapshot(const float *W, uint32_t ld, uint32_t first, uint32_t count, float *dst)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (c >= count) return;
    const float w = W[widx(first + r, first + c, ld)];
    dst[(size_t)r * count + c] = (w == w) ? w : 0.0f;
}

__global__ void k_graph_synthetic(float *W, uint32_t ld, uint32_t n_loc, uint32_t q0, uint32_t n_neurons,
                                  uint32_t n_tot, uint64_t seed, float lo, float hi, int with_diagonal)
{
    // one thread = one unit (4 consecutive rows of a column): full 16-byte stores, 1 KiB contiguous per wavefront
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ld) return;
    const uint32_t groups = (n_tot + 3u) >> 2;
    for (uint32_t g = blockIdx.y; g < groups; g += gridDim.y) {
        float out[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = g * 4 + k;
            out[k] = quiet_nan();
            if (q < n_loc && p < n_tot) {
                const uint32_t gq = q0 + q;
                if (with_diagonal || p != gq) out[k] = uniform_from_hash(seed, (uint64_t)p * n_neurons + gq, lo, hi);
            }
        }
        float4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
        reinterpret_cast<float4 *>(W)[(size_t)g * ld + q] = v;
    }
}

// ---- spike-train cells ---------------------------------------------------------------------
struct SpikeTrainArgs {
    CellArrays c;
    uint32_t n_cells;
    int st_kind;            // 1 Poisson (xorshift32), 2 Rate, 3 Preset, 4 BCM Poisson
    int nt_kind;
    int iterate;            // 0: only refresh presyn_value for `view_clock`
    const long long *lattice_clock;   // [n_st_lattices] clocks at the start of this run call
    long long step_offset;            // steps done since then
    long long view_clock;             // network clock of the NEXT input calculation
    float *vhist_row;                 // [c_pad] or null
    int has_nt;                       // some cell releases a neurotransmitter (else the flag planes are not read)
    // Sparse shard handles iterate only the cells their own rows read (the others are never looked at on this rank):
    // thread i handles cell cell_list[i], i < n_listed.  null: every cell.
    const uint32_t *cell_list;
    uint32_t n_listed;
    // sparse handles: where the rows of the NEXT input calculation read {presyn_value, never fired} (InputsArgs::st_view)
    uint2 *view_out;
};

// one spike-train cell: thread i of the cell job
__device__ __forceinline__ void spike_train_cell(const SpikeTrainArgs &a, const uint32_t i)
{
    if (i >= (a.cell_list ? a.n_listed : a.n_cells)) return;
    const uint32_t s = a.cell_list ? a.cell_list[i] : i;
    const CellArrays &c = a.c;
    int32_t lft = c.last_firing_time[s];          // requested before the iteration's own loads (one round trip less)
    if (a.iterate) {
        uint32_t spike;
        float custom_v = 0.0f;
        if (a.st_kind == 1 || a.st_kind == 4) {
            const uint32_t new_seed = xorshift32(c.seed[s]);
            c.seed[s] = new_seed;
            const float random_number = (float)new_seed / 4294967296.0f;   // (float) seed / 0xFFFFFFFF
            spike = random_number < uload(c.uni, CP_CHANCE, c.chance_of_firing, s);
        } else if (a.st_kind == CUSTOM_SPIKE_TRAIN) {
            // generated spike train (nb_macro lib.rs:4884-4891): its on_iteration writes the voltage and the flag
            float x[custom_st::NSTORE];
#pragma unroll
            for (int k = 0; k < custom_st::NVARS; ++k) x[k] = c.custom[k][s];
            float vc = c.current_voltage[s];
            bool sp = c.is_spiking[s] != 0;
            custom_st::on_iteration(vc, sp, x, c.dt[s], c.v_resting[s], c.v_th[s]);
#pragma unroll
            for (int k = 0; k < custom_st::NVARS; ++k) c.custom[k][s] = x[k];
            spike = sp ? 1u : 0u;
            custom_v = vc;
        } else if (a.st_kind == 3) {
            // PresetSpikeTrain::iterate, spike_train/mod.rs:803-827 (`step` holds internal_clock).  A cell
            // without firing times never fires (the reference would index an empty Vec).
            float clock = c.step[s] + c.dt[s];
            const uint32_t f0 = c.preset_ptr[s], len = c.preset_ptr[s + 1] - f0;
            uint32_t counter = c.counter[s];
            spike = len != 0 && clock > c.preset_times[f0 + counter];
            if (spike) {
                clock = 0.0f;
                counter += 1;
                if (counter == len) counter = 0;
                c.counter[s] = counter;
            }
            c.step[s] = clock;
        } else {
            float step = c.step[s] + c.dt[s];
            spike = (c.rate[s] != 0.0f) && (step >= c.rate[s]);
            if (spike) step = 0.0f;
            c.step[s] = step;
        }
        const float cell_v_th = uload(c.uni, CP_V_TH, c.v_th, s), cell_v_resting = uload(c.uni, CP_V_RESTING, c.v_resting, s);
        const float v = (a.st_kind == CUSTOM_SPIKE_TRAIN) ? custom_v : (spike ? cell_v_th : cell_v_resting);
        if (a.st_kind == 4) {
            // BCMPoissonNeuron::iterate (spike_train/mod.rs:931-954): activity = voltage change, replaced by the
            // firing rate when a window closes
            float cur = v - c.current_voltage[s], avg = c.bcm_avg[s], clock = c.bcm_clock[s];
            uint32_t num = c.bcm_num_spikes[s] + (spike ? 1u : 0u);
            bcm_window_update(clock, c.bcm_window[s], c.dt[s], num, c.bcm_period[s], cur, avg, true);
            c.bcm_cur[s] = cur; c.bcm_avg[s] = avg; c.bcm_clock[s] = clock; c.bcm_num_spikes[s] = num;
        }
        c.current_voltage[s] = v;
        c.is_spiking[s] = spike;
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) {
            const size_t i = (size_t)k * c.c_pad + s;
            if (!a.has_nt || !c.nt_flags[i]) continue;
            // spike trains release on their CURRENT spike flag (spike_train/mod.rs:363-365)
            if (SNN_HAVE_CUSTOM_NT && a.nt_kind == CUSTOM_KINETICS) {
                float x[custom_nt::NSTORE];
#pragma unroll
                for (int j = 0; j < custom_nt::NVARS; ++j) x[j] = c.nt_custom[j][i];
                float t = c.nt_t[i];
                custom_nt::apply(t, x, v, spike != 0, c.dt[s]);
#pragma unroll
                for (int j = 0; j < custom_nt::NVARS; ++j) c.nt_custom[j][i] = x[j];
                c.nt_t[i] = t;
                continue;
            }
            c.nt_t[i] = nt_apply(a.nt_kind, c.nt_t[i], c.nt_t_max[i], c.nt_clearance[i], c.nt_v_p[i], c.nt_k_p[i],
                                 v, spike, c.dt[s]);
        }
        if (spike) {
            lft = (int32_t)(a.lattice_clock[c.lattice_slot[s]] + a.step_offset);
            c.last_firing_time[s] = lft;
        }
        if (a.vhist_row) a.vhist_row[s] = v;
    }
    // presynaptic value of the next input calculation (spike_train_gap_junction, neuron/mod.rs:119-137):
    // never fired -> v_resting (used WITHOUT the conductance factor), else the refractoriness effect
    const uint32_t refr = uload(c.uni, CP_REFR, c.refractoriness, s);
    const float p_v_th = uload(c.uni, CP_V_TH, c.v_th, s), p_v_resting = uload(c.uni, CP_V_RESTING, c.v_resting, s);
    const float p_dt = uload(c.uni, CP_DT, c.dt, s), p_k = uload(c.uni, CP_K, c.k, s);
    float value;
    if (lft < 0) {
        value = p_v_resting;
    } else if (SNN_HAVE_CUSTOM_REFRACTORINESS && refr == CUSTOM_REFRACTORINESS) {
        // generated get_effect (nb_macro lib.rs:5736-5750): time_difference = (timestep - last_firing_time) as f32
        float xr[custom_refr::NSTORE] = {};
#pragma unroll
        for (int k = 0; k < custom_refr::NVARS; ++k) xr[k] = c.refr_custom[k][s];
        value = custom_refr::effect((float)(a.view_clock - (long long)lft), p_v_th, p_v_resting, p_dt, p_k, xr);
    } else {
        value = refr ? exponential_decay_effect(a.view_clock, lft, p_v_th, p_v_resting, p_k, p_dt)
                     : delta_dirac_effect(a.view_clock, lft, p_v_th, p_v_resting, p_k, p_dt);
    }
    if (a.view_out) a.view_out[s] = make_uint2(__float_as_uint(value), lft < 0 ? 1u : 0u);   // sparse handles read only the view
    else c.presyn_value[s] = value;
}

__global__ __launch_bounds__(256) void k_spike_trains(const SpikeTrainArgs a)
{
    spike_train_cell(a, blockIdx.x * 256 + threadIdx.x);
}

// ---- plasticity ------------------------------------------------------------------------------
struct StdpArgs {
    float *W;
    uint32_t ld, n_loc, q0, n_neurons, n_tot;
    RowMap rows;                          // sparse handles: local row <-> global neuron
    const float *xbuf;
    XLayout xl;
    const int32_t *last_firing_time;      // neurons (all shards, replicated)
    const int32_t *st_last_firing_time;   // cells
    const uint32_t *lattice_slot;         // [n_pad] neuron -> lattice slot
    const float *stdp;                    // [n_lattices][PL_STRIDE], see plasticity_weight
    // BCMActivity of neurons / cells (read only by lattices whose rule is BCM)
    const float *act, *avg, *st_act;
    const uint32_t *do_plasticity;        // [n_lattices]
    uint32_t *spike_list;                 // compacted gated spiking neurons (global indices)
    uint32_t *spike_count;
    // deferred STDP (dense handles, see snn_kernels_inputs.hpp): flag per neuron + the two delta vectors
    uint32_t *flag;                       // [n_pad] or null: 1 for every listed neuron, 0 otherwise
    float *dcol;                          // [n_lattices][dcol_stride]
    float *drow;                          // [ld]
    uint32_t dcol_stride, n_lattices;
    long long clock;                      // the step being closed (= last_firing_time of the listed neurons)
    // connections of a reward-modulated network (snn_set_connection_kind): [source][post lattice], source = the lattice slot of a
    // presynaptic neuron or n_lattices + the spike-train lattice slot of a cell; a non-zero kind belongs to k_reward_cross, not
    // to the plain rule.  null: none.
    const uint8_t *conn_kind;
    const uint32_t *st_lattice_slot;      // [c_pad] cell -> spike-train lattice slot
};
__device__ __forceinline__ bool plain_connection(const StdpArgs &a, uint32_t p, uint32_t post_slot)
{
    if (!a.conn_kind) return true;
    const uint32_t source = p < a.n_neurons ? a.lattice_slot[p] : a.n_lattices + a.st_lattice_slot[p - a.n_neurons];
    return a.conn_kind[(size_t)source * a.n_lattices + post_slot] == 0;
}

// Wave-ballot + popcount prefix compaction of the neurons that spiked in this step and whose
// lattice has do_plasticity set (the reference's positions_to_update, neuron/mod.rs:2544-2563).
__global__ __launch_bounds__(256) void k_spike_compact(const StdpArgs a)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (q < a.n_neurons) {
        const uint32_t spk = reinterpret_cast<const uint32_t *>(a.xbuf)[a.xl.at(q, PLANE_SPIKE)];
        hit = spk && a.do_plasticity[a.lattice_slot[q]];
        if (a.flag) a.flag[q] = hit ? 1u : 0u;
    }
    const unsigned long long mask = __ballot(hit);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.spike_count, (uint32_t)__popcll(mask));
    base = __shfl(base, 0, 64);
    if (hit) {
        const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        a.spike_list[base + prefix] = q;
    }
}

// Deferred STDP, evaluated at the end of the step that produced the spike list (before the spike trains advance):
//   dcol[l][p] = what an incoming edge p -> j gains when j (lattice l) spiked now:  stdp_delta(lft[p], clock; l)
//   drow[r]    = what an outgoing edge j -> r gains when j spiked now:              stdp_delta(clock, lft[q0 + r]; lattice of r)
// -- exactly the deltas k_stdp_columns / k_stdp_rows add (t_post resp. t_pre of a listed neuron is the clock).
__global__ __launch_bounds__(256) void k_stdp_prepare(const StdpArgs a)
{
    if (*a.spike_count == 0u) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const int32_t now = (int32_t)a.clock;
    if (i < a.n_tot) {
        const int32_t tp = (i < a.n_neurons) ? a.last_firing_time[i] : a.st_last_firing_time[i - a.n_neurons];
        for (uint32_t l = 0; l < a.n_lattices; ++l) {
            const float *prm = a.stdp + PL_STRIDE * l;
            a.dcol[(size_t)l * a.dcol_stride + i] = stdp_delta(tp, now, prm[0], prm[1], prm[2], prm[3], prm[4]);
        }
    }
    if (i < a.n_loc) {
        const uint32_t gr = a.q0 + i;
        const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[gr];
        a.drow[i] = stdp_delta(now, a.last_firing_time[gr], prm[0], prm[1], prm[2], prm[3], prm[4]);
    }
}

// "defer_stdp" 3: the ROW half of the update rides on the next input pass (k_inputs_dense<..., STDP = 2>).  At the end of the step
// that produced the spike list: drow as above, and one bit per listed neuron in rowbits ([n_chunks][8] words, zeroed before).
__global__ __launch_bounds__(256) void k_stdp_prepare_rows(const StdpArgs a, uint32_t *rowbits)
{
    const uint32_t count = *a.spike_count;
    if (count == 0u) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const int32_t now = (int32_t)a.clock;
    if (i < a.n_loc) {
        const uint32_t gr = a.q0 + i;
        const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[gr];
        a.drow[i] = stdp_delta(now, a.last_firing_time[gr], prm[0], prm[1], prm[2], prm[3], prm[4]);
    }
    if (i < count) {
        const uint32_t j = a.spike_list[i];
        atomicOr(&rowbits[j >> 5], 1u << (j & 31u));
    }
}

// How the STDP scatter kernels touch the matrix.  Few spikes (up to n_tot / 256): non-temporal accesses, so that the touched
// lines are not left dirty in L2 / Infinity Cache, where their write-back slowed the next streaming input pass by 0.2 ms at
// C4 with 83 spikes per step.  Many spikes: plain accesses -- the non-temporal read-modify-write is slower per line (0.12
// against 0.07 ms at 83 spikes, 1.8 against 1.1 ms at 826) and then costs more than it saves.
__device__ __forceinline__ bool stdp_streams(uint32_t count, uint32_t n_tot) { return (size_t)count * 256u <= n_tot; }
// (the non-temporal form is written as instructions: as builtins the two forms of a load or store are merged by the compiler
// and the hint is dropped)
__device__ __forceinline__ float stdp_load(const float *p, bool stream)
{
    float w;
    if (stream) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    else w = *p;
    return w;
}
__device__ __forceinline__ void stdp_store(float *p, float v, bool stream)
{
    if (stream) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else *p = v;
}

// The deferred update as standalone passes (a host access to the weights, the end of a run): the same two scatters as
// k_stdp_columns / k_stdp_rows with the prepared deltas.
__global__ __launch_bounds__(256) void k_stdp_apply_columns(const StdpArgs a)
{
    const uint32_t count = *a.spike_count;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_tot) return;
    const bool stream = stdp_streams(count, a.n_tot);
    for (uint32_t s = blockIdx.y; s < count; s += gridDim.y) {
        const uint32_t j = a.spike_list[s];
        if (j < a.q0 || j >= a.q0 + a.n_loc) continue;
        float *wp = a.W + widx(p, j - a.q0, a.ld);
        const float w = stdp_load(wp, stream);
        if (w == w) stdp_store(wp, w + a.dcol[(size_t)a.lattice_slot[j] * a.dcol_stride + p], stream);
    }
}
__global__ __launch_bounds__(256) void k_stdp_apply_rows(const StdpArgs a)
{
    const uint32_t count = *a.spike_count;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_loc) return;
    const float d = a.drow[r];
    const bool stream = stdp_streams(count, a.n_tot);
    for (uint32_t s = blockIdx.y; s < count; s += gridDim.y) {
        const uint32_t j = a.spike_list[s];
        float *wp = a.W + widx(j, r, a.ld);
        const float w = stdp_load(wp, stream);
        if (w == w) stdp_store(wp, w + d, stream);
    }
}

// incoming edges of every listed neuron that is local: column j, one thread per presynaptic row
__global__ __launch_bounds__(256) void k_stdp_columns(const StdpArgs a)
{
    const uint32_t count = *a.spike_count;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_tot) return;
    const int32_t tp = (p < a.n_neurons) ? a.last_firing_time[p] : a.st_last_firing_time[p - a.n_neurons];
    const bool stream = stdp_streams(count, a.n_tot);
    for (uint32_t s = blockIdx.y; s < count; s += gridDim.y) {
        const uint32_t j = a.spike_list[s];
        if (j < a.q0 || j >= a.q0 + a.n_loc) continue;
        float *wp = a.W + widx(p, j - a.q0, a.ld);
        const float w = stdp_load(wp, stream);
        if (w == w && plain_connection(a, p, a.lattice_slot[j])) {
            const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[j];
            const bool bcm = prm[5] != 0.0f;
            const float pre = bcm ? ((p < a.n_neurons) ? a.act[p] : a.st_act[p - a.n_neurons]) : 0.0f;
            stdp_store(wp, plasticity_weight(prm, w, tp, a.last_firing_time[j], pre, bcm ? a.act[j] : 0.0f, bcm ? a.avg[j] : 0.0f), stream);
        }
    }
}

// The same column scatter with the wavefront turned by 90 degrees (the form many spikes per step take, round 5): a lane owns the
// 16-byte unit (4 presynaptic rows) of ONE listed column, the 64 lanes of a wavefront hold 64 DIFFERENT listed columns of the
// SAME row group and walk the row groups of their slab together.  What k_stdp_columns asks of the memory system per
// instruction is 16 units that lie ld * 16 bytes apart (1.3 MB at C4: one page and one DRAM row each); here every
// instruction stays inside one row group's ld * 16 bytes, the presynaptic firing times are wave-uniform (scalar loads) and
// QUADS_IN_FLIGHT units per lane are requested before the first is used.  Same function of the same operands per
// synapse: bit-identical to k_stdp_columns.
typedef float stdp_v4f __attribute__((ext_vector_type(4)));
constexpr uint32_t STDP_QUADS_IN_FLIGHT = 4;
__global__ __launch_bounds__(256, 6) void k_stdp_columns_quads(const StdpArgs a)
{
    const uint32_t count = *a.spike_count;
    const uint32_t groups = (a.n_tot + 3u) >> 2;
    const uint32_t per_slab = (groups + gridDim.y - 1) / gridDim.y;
    const uint32_t g0 = blockIdx.y * per_slab, g1 = min(groups, g0 + per_slab);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t s0 = (blockIdx.x * 4u + wave) * 64u; s0 < count; s0 += gridDim.x * 256u) {
        const uint32_t s = s0 + lane;
        uint32_t j = s < count ? a.spike_list[s] : 0xFFFFFFFFu;
        const bool on = j != 0xFFFFFFFFu && j >= a.q0 && j < a.q0 + a.n_loc;
        if (!on) j = a.q0;                                   // an idle lane walks column 0 and stores nothing
        const uint32_t slot = a.lattice_slot[j];
        const float *prm = a.stdp + PL_STRIDE * slot;
        const bool bcm = prm[5] != 0.0f;
        const int32_t tj = a.last_firing_time[j];
        const float post_act = bcm ? a.act[j] : 0.0f, post_avg = bcm ? a.avg[j] : 0.0f;
        stdp_v4f *col = reinterpret_cast<stdp_v4f *>(a.W) + (j - a.q0);
        for (uint32_t g = g0; g < g1; g += STDP_QUADS_IN_FLIGHT) {
            stdp_v4f w[STDP_QUADS_IN_FLIGHT];
#pragma unroll
            for (uint32_t u = 0; u < STDP_QUADS_IN_FLIGHT; ++u)
                if (g + u < g1) w[u] = col[(size_t)(g + u) * a.ld];
            // (the arithmetic is not unrolled: sixteen inlined exponentials kept 197 registers alive -- two wavefronts per SIMD)
#pragma unroll 1
            for (uint32_t u = 0; u < STDP_QUADS_IN_FLIGHT; ++u) {
                if (g + u >= g1) break;
                const stdp_v4f w0 = w[u];
                bool changed = false;
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    const uint32_t p = (g + u) * 4u + (uint32_t)k;                  // wave-uniform
                    if (p >= a.n_tot) break;
                    const int32_t tp = (p < a.n_neurons) ? a.last_firing_time[p] : a.st_last_firing_time[p - a.n_neurons];
                    const float pre = bcm ? ((p < a.n_neurons) ? a.act[p] : a.st_act[p - a.n_neurons]) : 0.0f;
                    const float x = w0[k];
                    if (x == x && plain_connection(a, p, slot)) {
                        const float y = plasticity_weight(prm, x, tp, tj, pre, post_act, post_avg);
                        changed = changed || __float_as_uint(y) != __float_as_uint(x);
                        w[u][k] = y;
                    }
                }
                if (on && changed) col[(size_t)(g + u) * a.ld] = w[u];
            }
        }
    }
}

// outgoing edges of every listed neuron: row j, one thread per local postsynaptic column
__global__ __launch_bounds__(256) void k_stdp_rows(const StdpArgs a)
{
    const uint32_t count = *a.spike_count;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_loc) return;
    const uint32_t gr = a.q0 + r;
    const int32_t tr = a.last_firing_time[gr];
    const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[gr];
    const bool bcm = prm[5] != 0.0f;
    const float post_act = bcm ? a.act[gr] : 0.0f, post_avg = bcm ? a.avg[gr] : 0.0f;
    const bool stream = stdp_streams(count, a.n_tot);
    for (uint32_t s = blockIdx.y; s < count; s += gridDim.y) {
        const uint32_t j = a.spike_list[s];
        float *wp = a.W + widx(j, r, a.ld);
        const float w = stdp_load(wp, stream);
        if (w == w && plain_connection(a, j, a.lattice_slot[gr]))
            stdp_store(wp, plasticity_weight(prm, w, a.last_firing_time[j], tr, bcm ? a.act[j] : 0.0f, post_act, post_avg), stream);
    }
}

// Small dense networks (<= 1024 rows, the sizes of the one-launch step): compaction, column scatter and row scatter in ONE launch
// instead of four (counter fill 3.8 + k_spike_compact 3.6 + k_stdp_columns 2.6 + k_stdp_rows 2.4 us at 32 x 32, against 9.1 us for
// the neuron step itself: profiles/r04/small_stdp_kernel_stats.csv).  Every workgroup compacts the spike flags by itself (<= 1024
// loads, ballot + LDS prefix: the list comes out in ascending neuron order) and takes the listed neurons blockIdx.x, + gridDim.x,
// ...: thread p updates the incoming edge p -> j, thread r the outgoing edge j -> r.  STDP only (the host keeps BCM lattices on the
// separate kernels): a weight both of whose ends spiked in the step gets stdp_delta(t, t) = 0 from its column visit and from its
// row visit, possibly by two workgroups at once -- both store the same bits (w + 0.0f), whichever order they run in; every other
// weight is touched by one visit only.  Workgroup 0 also leaves the list and its length where the separate kernels would.
__global__ __launch_bounds__(1024) void k_stdp_small(const StdpArgs a)
{
    __shared__ uint32_t s_list[1024];
    __shared__ uint32_t s_wave[16];
    const uint32_t q = threadIdx.x, lane = q & 63u, wave = q >> 6;
    bool hit = false;
    if (q < a.n_neurons) {
        const uint32_t spk = reinterpret_cast<const uint32_t *>(a.xbuf)[a.xl.at(q, PLANE_SPIKE)];
        hit = spk && a.do_plasticity[a.lattice_slot[q]];
    }
    const unsigned long long mask = __ballot(hit);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0, count = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
        base += w < wave ? s_wave[w] : 0u;
        count += s_wave[w];
    }
    if (hit) s_list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = q;
    __syncthreads();
    if (blockIdx.x == 0) {
        if (q < count) a.spike_list[q] = s_list[q];
        if (q == 0) *a.spike_count = count;
    }
    const uint32_t p = q;                                      // the thread's presynaptic row ...
    const uint32_t r = q;                                      // ... and its local postsynaptic column
    const int32_t tp = p < a.n_tot ? ((p < a.n_neurons) ? a.last_firing_time[p] : a.st_last_firing_time[p - a.n_neurons]) : -1;
    const uint32_t gr = a.q0 + r;
    const int32_t tr = r < a.n_loc ? a.last_firing_time[gr] : -1;
    const float *prm_r = a.stdp + PL_STRIDE * (r < a.n_loc ? a.lattice_slot[gr] : 0u);
    for (uint32_t s = blockIdx.x; s < count; s += gridDim.x) {
        const uint32_t j = s_list[s];
        const int32_t tj = a.last_firing_time[j];
        if (p < a.n_tot && j >= a.q0 && j < a.q0 + a.n_loc) {                    // incoming edge p -> j (k_stdp_columns)
            float *wp = a.W + widx(p, j - a.q0, a.ld);
            const float w = *wp;
            if (w == w && plain_connection(a, p, a.lattice_slot[j])) {
                const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[j];
                *wp = plasticity_weight(prm, w, tp, tj, 0.0f, 0.0f, 0.0f);
            }
        }
        if (r < a.n_loc) {                                                       // outgoing edge j -> r (k_stdp_rows)
            float *wp = a.W + widx(j, r, a.ld);
            const float w = *wp;
            if (w == w && plain_connection(a, j, a.lattice_slot[gr])) *wp = plasticity_weight(prm_r, w, tj, tr, 0.0f, 0.0f, 0.0f);
        }
    }
}

// ---- reduced per-lattice histories ---------------------------------------------------------------
// AverageVoltageHistory (neuron/mod.rs:305-322) and EEGHistory (:233-284) on the device: one float per lattice
// and step instead of the T x N voltage history.  One workgroup per lattice; every thread sums one 256-neuron
// chunk sequentially, thread 0 adds the chunk partials in ascending order -- the canonical chunked order the
// oracle uses (the reference sums strictly sequentially).
struct SummaryArgs {
    const float *xbuf;
    XLayout xl;
    const uint32_t *first, *count;     // [n_lattices]
    float *avg_row, *eeg_row;          // this step's rows [n_lattices] (either may be null)
    float reference_voltage, distance, conductivity;
};

__global__ __launch_bounds__(256) void k_lattice_summary(const SummaryArgs a)
{
    __shared__ float s_sum[256], s_eeg[256];
    const uint32_t l = blockIdx.x;
    const uint32_t first = a.first[l], count = a.count[l];
    const uint32_t n_chunks = (count + CHUNK - 1) / CHUNK;
    float tot = 0.0f, tot_e = 0.0f;
    for (uint32_t g = 0; g < n_chunks; g += 256) {
        const uint32_t c = g + threadIdx.x;
        float part = 0.0f, part_e = 0.0f;
        if (c < n_chunks) {
            const uint32_t i1 = min(count, (c + 1) * CHUNK);
            for (uint32_t i = c * CHUNK; i < i1; ++i) {
                const float v = a.xbuf[a.xl.at(first + i, PLANE_V)];
                part += v;
                part_e += v - a.reference_voltage;
            }
        }
        s_sum[threadIdx.x] = part;
        s_eeg[threadIdx.x] = part_e;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t m = min(256u, n_chunks - g);
            for (uint32_t t = 0; t < m; ++t) { tot += s_sum[t]; tot_e += s_eeg[t]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float pi = 3.14159274101257324f;     // std::f32::consts::PI
        if (a.avg_row) a.avg_row[l] = tot / (float)count;
        if (a.eeg_row) a.eeg_row[l] = (1.0f / (4.0f * pi * a.conductivity * a.distance)) * tot_e;
    }
}

// HBM ceilings of THIS device, measured with the access shape of k_inputs_dense (16 B per lane, nt):
// read-only stream (what the synaptic-input pass can reach at best) and read+write copy.
typedef float probe_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_read(const probe_v4f *src, size_t n4, float *sink)
{
    probe_v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const probe_v4f v = __builtin_nontemporal_load(src + i);
        acc += v;
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) *sink = s;      // never true for the zero-filled buffer: keeps the loads alive
}
__global__ __launch_bounds__(256) void k_probe_copy(const probe_v4f *src, probe_v4f *dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// two matrices of one layout read and written back in place, index-aligned: the four streams of k_inputs_rstdp (weights
// and traces, read + write) without changing a bit -- times a candidate placement of the trace matrix next to W
// (source and destination arrive as separate arguments: a store of the value just loaded from the same pointer would be
// removed by the compiler, and the loads with it)
__global__ __launch_bounds__(256) void k_probe_rw_pair(const probe_v4f *a, probe_v4f *a_out, const probe_v4f *b, probe_v4f *b_out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const probe_v4f va = __builtin_nontemporal_load(a + i), vb = __builtin_nontemporal_load(b + i);
        __builtin_nontemporal_store(va, a_out + i);
        __builtin_nontemporal_store(vb, b_out + i);
    }
}

// Synthetic drive (benchmarks and load tests only, off by default): before the step at `clock`, every neuron q with
// hash32(seed, clock * n + q) < threshold has its membrane voltage set to `voltage` -- a device-side generator of a
// chosen spike rate, so that plasticity can be measured under load without the host touching the state each step.
__global__ void k_synthetic_drive(float *xbuf, XLayout xl, uint32_t n, uint64_t seed, long long clock, uint32_t threshold,
                                  float voltage)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n && hash32(seed, (uint64_t)clock * n + q) < threshold) xbuf[xl.at(q, PLANE_V)] = voltage;
}

// Uniform-parameter scan (UniformTable, snn_layout.hpp): slot <- {1, bits of arr[0]}, cleared by any element that differs
__global__ void k_uniform_begin(UniformTable *t, int slot, const uint32_t *arr, uint32_t n)
{
    t->flag[slot] = n ? 1u : 0u;
    t->bits[slot] = n ? arr[0] : 0u;
}
__global__ __launch_bounds__(256) void k_uniform_scan(UniformTable *t, int slot, const uint32_t *arr, uint32_t n)
{
    const uint32_t first = arr[0];
    bool differs = false;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) differs = differs || arr[i] != first;
    if (differs) t->flag[slot] = 0u;
}

// device-function probe for the parity tests of the scalar formulas
__global__ void k_probe_math(int which, const float *in, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    out[i] = which == 0 ? expf_glibc(x) : (which == 1 ? pow3f_glibc(x) : pow4f_glibc(x));
}

// the same over a range of float bit patterns generated on the device: x = asfloat(first + i * stride); which = 3: powf(x, y)
__global__ void k_probe_math_bits(int which, uint32_t first, uint32_t stride, float y, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(first + (uint32_t)i * stride);
    if (which >= 4) {
        // the form the Hodgkin-Huxley step uses: the branch-free main path, the full function where that one says "special"
        bool special = false;
        float r = which == 4 ? expf_glibc_main(x, special) : which == 5 ? powf_glibc_main(x, 3.0f, special) : powf_glibc_main(x, 4.0f, special);
        if (special) r = which == 4 ? expf_glibc(x) : which == 5 ? pow3f_glibc(x) : pow4f_glibc(x);
        out[i] = r;
        return;
    }
    out[i] = which == 0 ? expf_glibc(x) : which == 1 ? pow3f_glibc(x) : which == 2 ? pow4f_glibc(x) : powf_glibc(x, y);
}

} // namespace snn

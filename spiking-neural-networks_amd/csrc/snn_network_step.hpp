// Kernel launches and the step loop of run_lattice / run_lattices (neuron/gpu_lattices/mod.rs:791-896, 2284-2583):
// input pass, neuron update (two kernels or the one-launch small-lattice step), plasticity, reward modulation,
// histories, HBM placement of the synapse matrix, run bookkeeping and dense graph transfers.
// Included by snn_network.hip only (one translation unit).
#pragma once
#include "snn_network_state.hpp"

namespace {

// Arguments of the cells' job (k_spike_trains, or the cell blocks of k_step_csr / k_step_close); returns the number of
// threads it needs.  iterate = 1 flips the sparse handles' view: the launch writes the copy the NEXT input calculation reads.
uint32_t spike_train_args(snn_network *net, SpikeTrainArgs &a, int iterate, long long step_offset, long long view_clock)
{
    a = SpikeTrainArgs{};
    if (net->nc == 0) return 0;
    a.c = net->ca; a.n_cells = net->nc; a.st_kind = net->st_kind; a.nt_kind = net->nt_kind;
    a.iterate = iterate; a.lattice_clock = net->st_clock_dev; a.step_offset = step_offset;
    a.has_nt = net->any_nt_cells ? 1 : 0;
    a.view_clock = view_clock;
    a.vhist_row = (iterate && record_now(net) && net->want_vhist && net->st_vhist) ? net->st_vhist + (size_t)net->hist_steps * net->c_pad : nullptr;
    a.cell_list = net->cell_list_dev; a.n_listed = net->n_cells_listed;
    const uint32_t work = net->cell_list_dev ? net->n_cells_listed : net->nc;
    if (work && net->cell_view[0]) {
        if (iterate) net->cell_view_cur ^= 1;
        a.view_out = net->cell_view[net->cell_view_cur];
    }
    return work;
}

int launch_spike_trains(snn_network *net, int iterate, long long step_offset, long long view_clock)
{
    SpikeTrainArgs a;
    const uint32_t work = spike_train_args(net, a, iterate, step_offset, view_clock);
    if (work == 0) return SNN_OK;
    hipLaunchKernelGGL(k_spike_trains, dim3((work + 255) / 256), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

enum InputsPart { INPUTS_ALL = 0, INPUTS_LOCAL = 1, INPUTS_REMOTE = 2 };
int flush_rstdp(snn_network *net);
int flush_stdp(snn_network *net);
bool fused_step_possible(const snn_network *net);

// the dense synapse matrix is streamed (not cache resident): the shapes the fused weight updates exist for
inline bool matrix_streamed(const snn_network *net)
{
    return !net->csr && (size_t)net->n_tot * net->ld * 4 > ((size_t)64 << 20);
}

// STDP of step t applied by the input pass of step t + 1 (k_inputs_dense<..., STDP>): dense streamed matrices, plain
// STDP everywhere (the BCM rule reads the weight itself), nothing else rewriting or snapshotting W in step_end
bool stdp_deferral_applies(const snn_network *net)
{
    if (SNN_HAVE_CUSTOM_MODEL) return false;      // a library carrying generated code keeps the standalone kernels (shorter compile)
    if (!net->defer_stdp || !net->any_plasticity || net->any_modulation || net->any_whist || !matrix_streamed(net) || net->any_conn_kind ||
        net->lattices.size() > (size_t)STDP_MAX_LATTICES || net->n_loc == 0)
        return false;
    for (size_t l = 0; l < net->lattices.size(); ++l)
        if (net->stdp_host[l * PL_STRIDE + 5] != 0.0f) return false;
    return true;
}

StdpArgs stdp_args(snn_network *net)
{
    StdpArgs a{};
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.rows = net->rowmap; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = net->xbuf; a.xl = net->xl;
    a.last_firing_time = net->na.last_firing_time; a.st_last_firing_time = net->ca.last_firing_time;
    a.lattice_slot = net->lattice_slot; a.stdp = net->stdp_dev; a.do_plasticity = net->plast_dev;
    a.act = net->na.bcm_cur; a.avg = net->na.bcm_avg; a.st_act = net->ca.bcm_cur;
    a.spike_list = net->spike_list; a.spike_count = net->spike_count;
    a.flag = nullptr; a.dcol = net->stdp_dcol; a.drow = net->stdp_drow;
    a.dcol_stride = net->dcol_stride; a.n_lattices = (uint32_t)net->lattices.size();
    a.clock = net->clock;
    a.conn_kind = net->any_conn_kind ? net->conn_kind_dev : nullptr;
    a.st_lattice_slot = net->ca.lattice_slot;
    return a;
}
// snn_network_exchange.hpp
int ensure_exchange_plan(snn_network *net);
int launch_exchange_unpack(snn_network *net);

// chunks whose presynaptic rows all belong to this shard's own neurons
void local_chunks(const snn_network *net, uint32_t *begin, uint32_t *count)
{
    const uint32_t cb = (net->q0 + CHUNK - 1) / CHUNK, ce = net->q1 / CHUNK;
    *begin = cb;
    *count = ce > cb ? ce - cb : 0;
}

int launch_inputs(snn_network *net, InputsPart part = INPUTS_ALL)
{
    if (net->n_loc == 0 || net->n_tot == 0) return SNN_OK;
    uint32_t lc_begin = 0, lc_count = 0;
    local_chunks(net, &lc_begin, &lc_count);
    uint32_t grid_chunks = net->n_chunks;
    InputsArgs a{};
    a.chunk_first = 0; a.hole_begin = net->n_chunks; a.hole_count = 0;
    if (part == INPUTS_LOCAL) { a.chunk_first = lc_begin; grid_chunks = lc_count; }
    if (part == INPUTS_REMOTE) { a.hole_begin = lc_begin; a.hole_count = lc_count; grid_chunks = net->n_chunks - lc_count; }
    if (grid_chunks == 0) return SNN_OK;
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.rows = net->rowmap; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = net->xbuf; a.xl = net->xl; a.gap_conductance = net->na.gap_conductance; a.uni = net->uni_neuron;
    a.st_value = net->ca.presyn_value; a.st_last_firing_time = net->ca.last_firing_time;
    a.st_view = net->cell_view[net->cell_view_cur];
    a.st_nt_t = net->ca.nt_t; a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
    a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_chunks = net->n_chunks;
    for (int k = 0; k < K_TYPES; ++k) a.live_type[k] = net->live_type[k];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (net->profile) {
        if (net->ev_used == net->ev_pool.size()) {
            hipEvent_t x, y;
            HIP_TRY(hipEventCreate(&x), SNN_ERR_QUEUE);
            HIP_TRY(hipEventCreate(&y), SNN_ERR_QUEUE);
            net->ev_pool.emplace_back(x, y);
        }
        e0 = net->ev_pool[net->ev_used].first;
        e1 = net->ev_pool[net->ev_used].second;
        net->ev_counts.resize(net->ev_pool.size(), 1);
        // LOCAL + REMOTE = one pass over W (counted on the REMOTE half; a lone shard has no REMOTE half)
        net->ev_counts[net->ev_used] = (part == INPUTS_LOCAL && lc_count < net->n_chunks) ? 0 : 1;
        ++net->ev_used;
        HIP_TRY(hipEventRecord(e0, net->stream), SNN_ERR_QUEUE);
    }
    if (net->csr) {
        if (net->csr_ptr) {
            CsrInputsArgs ca{};
            ca.g = csr_graph(net);
            ca.in = a;
            dim3 g((((net->n_loc + 63) / 64) * 64 + 255) / 256);
            if (net->electrical && net->chemical) hipLaunchKernelGGL((k_inputs_csr<true, true>), g, dim3(256), 0, net->stream, ca);
            else if (net->electrical) hipLaunchKernelGGL((k_inputs_csr<true, false>), g, dim3(256), 0, net->stream, ca);
            else hipLaunchKernelGGL((k_inputs_csr<false, true>), g, dim3(256), 0, net->stream, ca);
        } else {   // no graph set: no edges
            HIP_TRY(hipMemsetAsync(net->part_i, 0, (size_t)net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
            HIP_TRY(hipMemsetAsync(net->part_t, 0, (size_t)K_TYPES * net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
        }
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    // shape of the pass: cache-resident matrices take the latency-oriented one-wave shape; streamed matrices the
    // 4-columns-per-lane shape, or the 2-column shape while that would leave the chip under-filled
    const bool resident = (size_t)net->n_tot * net->ld * 4 <= ((size_t)64 << 20);
    const uint64_t waves4 = (uint64_t)((net->n_loc + 255) / 256) * grid_chunks;
    // same-box A/B of the two streamed shapes (profiles/ab_input_shape.py, profiles/r03/ab_input_shape_by_size.txt): the
    // 2-column shape wins by 0.8 - 2.4 % from 96x96 to 240x240 (waves4 up to 50 625), the 4-column shape by 0.5 % at 256x256
    int shape = resident ? 0 : (waves4 < 57600 ? 2 : 1);
    if (net->force_shape > 0 && !resident) shape = net->force_shape;     // SNN_AMD_INPUT_SHAPE=1|2 (experiments)
    if (net->rstdp_pending && part != INPUTS_ALL) TRY(flush_rstdp(net));
    const bool stdp_fused = net->stdp_pending && !net->rstdp_pending && part == INPUTS_ALL && shape != 0;
    if (net->stdp_pending && !stdp_fused) TRY(flush_stdp(net));
#if !SNN_HAVE_CUSTOM_MODEL
    if (stdp_fused) {
        // the STDP update of the previous step rides on this pass over W
        net->stdp_pending = false;
        const bool rows_only = net->stdp_pending_rows_only;
        net->stdp_pending_rows_only = false;
        a.W_rw = net->W; a.stdp_count = net->spike_count; a.stdp_flag = net->stdp_flag;
        a.stdp_dcol = net->stdp_dcol; a.stdp_drow = net->stdp_drow; a.lattice_slot = net->lattice_slot;
        a.dcol_stride = net->dcol_stride; a.n_lattices = (uint32_t)net->lattices.size();
        a.stdp_rowbits = net->stdp_rowbits;
#define SNN_LAUNCH_SSHAPE(E, C, SH)                                                                       \
    do {                                                                                                 \
        const dim3 g_((net->n_loc + InputsShape<SH>::TILE - 1) / InputsShape<SH>::TILE, grid_chunks);    \
        if (rows_only) hipLaunchKernelGGL((k_inputs_dense<E, C, SH, K_TYPES, 2>), g_, dim3(InputsShape<SH>::THREADS), 0, net->stream, a); \
        else hipLaunchKernelGGL((k_inputs_dense<E, C, SH, K_TYPES, 1>), g_, dim3(InputsShape<SH>::THREADS), 0, net->stream, a); \
    } while (0)
#define SNN_LAUNCH_SINPUTS(E, C)                                                                         \
    do {                                                                                                 \
        if (shape == 1) SNN_LAUNCH_SSHAPE(E, C, 1);                                                      \
        else SNN_LAUNCH_SSHAPE(E, C, 2);                                                                 \
    } while (0)
        for (int k = 0; k < K_TYPES; ++k) a.live_type[k] = (uint32_t)k;     // the generic three-slot chemical variant
        if (net->electrical && net->chemical) SNN_LAUNCH_SINPUTS(true, true);
        else if (net->electrical) SNN_LAUNCH_SINPUTS(true, false);
        else SNN_LAUNCH_SINPUTS(false, true);
#undef SNN_LAUNCH_SINPUTS
#undef SNN_LAUNCH_SSHAPE
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
        return SNN_OK;
    }
#endif
    if (net->rstdp_pending) {
        // the reward-modulated weight update of the previous step rides on this pass over W
        net->rstdp_pending = false;
        RstdpInputsArgs ra{};
        ra.in = a; ra.W = net->W; ra.C = net->trace;
        ra.last_firing_time = net->na.last_firing_time; ra.lattice_slot = net->lattice_slot;
        ra.rm = net->rm_dev; ra.rm_on = net->rm_on_dev;
        ra.dop = net->reward_since_defer ? RM_DOPAMINE_BEFORE : RM_DOPAMINE;
#define SNN_LAUNCH_RSHAPE(E, C, SH)                                                                       \
    hipLaunchKernelGGL((k_inputs_rstdp<E, C, SH>),                                                       \
                       dim3((net->n_loc + InputsShape<SH>::TILE - 1) / InputsShape<SH>::TILE, grid_chunks), \
                       dim3(InputsShape<SH>::THREADS), 0, net->stream, ra)
#define SNN_LAUNCH_RINPUTS(E, C)                                                                         \
    do {                                                                                                 \
        if (shape == 1) SNN_LAUNCH_RSHAPE(E, C, 1);                                                      \
        else if (shape == 2) SNN_LAUNCH_RSHAPE(E, C, 2);                                                 \
        else SNN_LAUNCH_RSHAPE(E, C, 0);                                                                 \
    } while (0)
        if (net->electrical && net->chemical) SNN_LAUNCH_RINPUTS(true, true);
        else if (net->electrical) SNN_LAUNCH_RINPUTS(true, false);
        else SNN_LAUNCH_RINPUTS(false, true);
#undef SNN_LAUNCH_RINPUTS
#undef SNN_LAUNCH_RSHAPE
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
        return SNN_OK;
    }
#define SNN_LAUNCH_SHAPE(E, C, SH, NT)                                                                    \
    hipLaunchKernelGGL((k_inputs_dense<E, C, SH, NT>),                                                   \
                       dim3((net->n_loc + InputsShape<SH>::TILE - 1) / InputsShape<SH>::TILE, grid_chunks), \
                       dim3(InputsShape<SH>::THREADS), 0, net->stream, a)
#define SNN_LAUNCH_INPUTS(E, C, NT)                                                                      \
    do {                                                                                                 \
        if (shape == 1) SNN_LAUNCH_SHAPE(E, C, 1, NT);                                                   \
        else if (shape == 2) SNN_LAUNCH_SHAPE(E, C, 2, NT);                                              \
        else SNN_LAUNCH_SHAPE(E, C, 0, NT);                                                              \
    } while (0)
#if !SNN_HAVE_CUSTOM_MODEL
#define SNN_LAUNCH_CHEM(E)                                                                               \
    do {                                                                                                 \
        if (net->n_live == 1) SNN_LAUNCH_INPUTS(E, true, 1);                                             \
        else if (net->n_live == 2) SNN_LAUNCH_INPUTS(E, true, 2);                                        \
        else SNN_LAUNCH_INPUTS(E, true, 3);                                                              \
    } while (0)
#else       // a library carrying generated code: the generic three-slot variant only (shorter compile)
#define SNN_LAUNCH_CHEM(E)                                                                               \
    do {                                                                                                 \
        for (int k = 0; k < K_TYPES; ++k) a.live_type[k] = (uint32_t)k;                                  \
        SNN_LAUNCH_INPUTS(E, true, 3);                                                                   \
    } while (0)
#endif
    if (net->electrical && net->chemical) SNN_LAUNCH_CHEM(true);
    else if (net->electrical) SNN_LAUNCH_INPUTS(true, false, 3);
    else SNN_LAUNCH_CHEM(false);
#undef SNN_LAUNCH_CHEM
#undef SNN_LAUNCH_INPUTS
#undef SNN_LAUNCH_SHAPE
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    return SNN_OK;
}

// -DSNN_LAB_BUILD: a library for kernel experiments (profiles/experiments/README.md) that instantiates the model-templated kernels
// for Izhikevich and Hodgkin-Huxley only -- a quarter of the compile time; never the library the tests or the bench load by default
#ifdef SNN_LAB_BUILD
#define SNN_FOR_MODEL(MACRO)                                                                                         \
    switch (net->model) {                                                                                            \
    case 2: MACRO(2); break;                                                                                         \
    default: MACRO(0); break;                                                                                        \
    }
#else
#define SNN_FOR_MODEL(MACRO)                                                                                         \
    switch (net->model) {                                                                                            \
    case 1: MACRO(1); break;                                                                                         \
    case 2: MACRO(2); break;                                                                                         \
    case 3: MACRO(3); break;                                                                                         \
    case 4: MACRO(4); break;                                                                                         \
    case 5: MACRO(5); break;                                                                                         \
    case 6: MACRO(6); break;                                                                                         \
    case 7: MACRO(7); break;                                                                                         \
    default: MACRO(0); break;                                                                                        \
    }
#endif

int launch_update(snn_network *net)
{
    if (net->n_loc == 0) return SNN_OK;
    UpdateArgs a{};
    a.n = net->na;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_in = net->n_in; a.tcount = net->tcount;
    a.ld = net->ld; a.n_chunks = net->n_tot ? net->n_chunks : 0; a.q0 = net->q0; a.n_loc = net->n_loc; a.rows = net->rowmap;
    a.clock = net->clock;
    a.electrical = net->electrical; a.chemical = net->chemical; a.nt_kind = net->nt_kind; a.rc_kind = net->rc_kind;
    a.vhist_row = (record_now(net) && net->want_vhist && net->vhist) ? net->vhist + (size_t)net->hist_steps * net->n_pad : nullptr;
    a.spike_row = (record_now(net) && net->want_raster && net->raster) ? net->raster + (size_t)net->hist_steps * (net->n_pad / 64) : nullptr;
    a.spike_counts = net->want_counts ? net->spike_counts : nullptr;
    a.xout = net->xbuf; a.xout2 = nullptr;
    a.has_nt = net->any_nt_neurons ? 1 : 0;
    a.live_mask = net->live_mask_applied;       // transmitter types some neuron or cell releases (ensure_counts; all ones: unknown)
    a.bcm = net->model == SNN_MODEL_BCM_IZHIKEVICH;
    a.model_is_custom = net->model == SNN_MODEL_CUSTOM;
    // dense shard handles: the own slot of the all-gather buffer is written by this launch (no pack launch)
    net->update_packed = false;
    if (net->sharded && !net->csr && net->update_packs && !net->x_dirty && net->x_mode == SNN_EXCHANGE_ALLGATHER &&
        net->wire && net->x_block_words && net->n_loc <= net->shard_stride) {
        a.wire_out = net->wire + (size_t)net->shard_index * net->x_block_words;
        a.wire_count = net->shard_stride; a.wire_planes = net->x_planes;
        for (uint32_t s = 0; s < net->x_planes; ++s) a.wire_plane_id[s] = net->x_plane_id[s];
        net->update_packed = true;
    }
    net->shadow_valid = false;            // the exchange buffer moves on without the shadows
    const uint32_t ub = 256u;          // (one wavefront per workgroup for small launches was measured: no gain)
    dim3 grid((net->ld + ub - 1) / ub);
    // built-in models on dense handles with chemical synapses: the partials of every plane requested together
    const bool all_planes = net->update_all_planes && net->chemical && !net->csr && net->model != SNN_MODEL_CUSTOM;
    // ... and, by default, WIDE: four wavefronts share a column's partials and three of them warm the cache for the fourth
    const bool wide = all_planes && net->update_all_planes >= 2 && a.n_chunks >= 4 && (a.n_chunks + 3) / 4 <= 32;
    TouchList touch{};
    if (wide) {
        grid = dim3(net->ld / 64);
        auto add = [&](const void *p) {
            if (!p || touch.n_neuron >= 56) return;
            for (uint32_t i = 0; i < touch.n_neuron; ++i) if (touch.by_neuron[i] == p) return;
            touch.by_neuron[touch.n_neuron++] = static_cast<const uint32_t *>(p);
        };
        for (const auto &kv : net->neuron_attrs) {
            const Attr &at = kv.second;
            if (at.store == S_PLAIN) add(at.base);
            else if (at.store == S_PLAIN_K)
                for (int k = 0; k < K_TYPES; ++k) if (a.live_mask >> k & 1u) add(static_cast<const uint32_t *>(at.base) + (size_t)k * at.pad);
        }
        for (int pl : {PLANE_V, PLANE_SPIKE}) add(net->xbuf + net->xl.at(0, pl));
        for (int k = 0; k < K_TYPES; ++k) if (a.live_mask >> k & 1u) add(net->xbuf + net->xl.at(0, PLANE_T0 + k));
        touch.by_column[touch.n_column++] = net->n_in;
        for (int k = 0; k < K_TYPES; ++k) if (a.live_mask >> k & 1u) touch.by_column[touch.n_column++] = net->tcount + (size_t)k * net->ld;
        if (net->update_all_planes == 3) touch.n_neuron = touch.n_column = 0;      // (A/B: the wide form without the cache warming)
    }
#define SNN_LAUNCH_UPDATE(M) do { \
        if (wide) hipLaunchKernelGGL((k_update_wide<M>), grid, dim3(256), 0, net->stream, a, touch); \
        else if (all_planes) hipLaunchKernelGGL((k_update<M, true>), grid, dim3(ub), 0, net->stream, a); \
        else hipLaunchKernelGGL((k_update<M, false>), grid, dim3(ub), 0, net->stream, a); } while (0)
    switch (net->model) {
    case SNN_MODEL_HODGKIN_HUXLEY: SNN_LAUNCH_UPDATE(2); break;
#ifndef SNN_LAB_BUILD
    case SNN_MODEL_LIF: SNN_LAUNCH_UPDATE(1); break;
    case SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE: SNN_LAUNCH_UPDATE(3); break;
    case SNN_MODEL_SIMPLE_LIF: SNN_LAUNCH_UPDATE(4); break;
    case SNN_MODEL_ADAPTIVE_LIF: SNN_LAUNCH_UPDATE(5); break;
    case SNN_MODEL_ADAPTIVE_EXP_LIF: SNN_LAUNCH_UPDATE(6); break;
    case SNN_MODEL_LEAKY_IZHIKEVICH: SNN_LAUNCH_UPDATE(7); break;
#endif
#if SNN_HAVE_CUSTOM_NEURON
    case SNN_MODEL_CUSTOM: hipLaunchKernelGGL((k_update<CUSTOM_MODEL, false>), grid, dim3(ub), 0, net->stream, a); break;
#endif
    default: SNN_LAUNCH_UPDATE(0); break;
    }
#undef SNN_LAUNCH_UPDATE
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

int launch_plasticity_kernels(snn_network *net);

// spike compaction + weight updates of the step, bracketed by HIP events while profiling is on
int launch_plasticity(snn_network *net)
{
    if (!net->any_plasticity || net->nn == 0) return SNN_OK;
    if (!net->profile) return launch_plasticity_kernels(net);
    if (net->ev_used_pl == net->ev_pool_pl.size()) {
        hipEvent_t x, y;
        HIP_TRY(hipEventCreate(&x), SNN_ERR_QUEUE);
        HIP_TRY(hipEventCreate(&y), SNN_ERR_QUEUE);
        net->ev_pool_pl.emplace_back(x, y);
    }
    const auto ev = net->ev_pool_pl[net->ev_used_pl++];
    HIP_TRY(hipEventRecord(ev.first, net->stream), SNN_ERR_QUEUE);
    TRY(launch_plasticity_kernels(net));
    HIP_TRY(hipEventRecord(ev.second, net->stream), SNN_ERR_QUEUE);
    return SNN_OK;
}

// small dense unsharded networks under STDP (no BCM lattice among the plastic ones): compaction and both scatters in one launch
bool stdp_small_applies(const snn_network *net)
{
    if (!net->stdp_small || net->csr || net->sharded || net->n_tot > 1024u || net->n_loc != net->nn || net->nn == 0) return false;
    // ANY lattice under the BCM rule rules the form out, plastic or not (as in stdp_deferral_applies): k_stdp_small visits the
    // outgoing edge j -> r under the rule of r's lattice with zero activities, which is not what k_stdp_rows computes for a BCM
    // lattice whose own do_plasticity is off but which receives edges from a plastic STDP lattice
    for (size_t l = 0; l < net->lattices.size(); ++l)
        if (net->stdp_host[l * PL_STRIDE + 5] != 0.0f) return false;
    return true;
}

int launch_plasticity_kernels(snn_network *net)
{
    StdpArgs a = stdp_args(net);
    if (stdp_small_applies(net)) {
        hipLaunchKernelGGL(k_stdp_small, dim3(16), dim3(1024), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    const bool defer = stdp_deferral_applies(net);
    if (defer) a.flag = net->stdp_flag;
    HIP_TRY(hipMemsetAsync(net->spike_count, 0, 4, net->stream), SNN_ERR_BUFFER_WRITE);
    hipLaunchKernelGGL(k_spike_compact, dim3((net->nn + 255) / 256), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (net->n_loc == 0) return SNN_OK;
    if (defer && net->defer_stdp == 3) {
        // the incoming edges now (the column scatter), the outgoing edges with the next input pass: the row half costs that pass
        // no fetch and only full-line stores, where k_stdp_rows reads and writes back every line of the listed rows
        hipLaunchKernelGGL(k_stdp_columns, dim3((net->n_tot + 255) / 256, 64), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(hipMemsetAsync(net->stdp_rowbits, 0, (size_t)net->n_chunks * 32, net->stream), SNN_ERR_BUFFER_WRITE);
        hipLaunchKernelGGL(k_stdp_prepare_rows, dim3((std::max(net->nn, net->n_loc) + 255) / 256), dim3(256), 0, net->stream, a, net->stdp_rowbits);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        net->stdp_pending = true;
        net->stdp_pending_rows_only = true;
        return SNN_OK;
    }
    if (defer) {
        // only the two delta vectors now; the weights are rewritten by the next input pass (or by flush_stdp)
        hipLaunchKernelGGL(k_stdp_prepare, dim3((std::max(net->n_tot, net->n_loc) + 255) / 256), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        net->stdp_pending = true;
        if (net->defer_stdp == 2) TRY(flush_stdp(net));      // the prepared deltas applied right away by the scatter passes
        return SNN_OK;
    }
    if (net->csr) {
        if (!net->csr_ptr) return SNN_OK;
        CsrStdpArgs ca{};
        ca.g = csr_graph(net);
        ca.s = a;
        net->img_stale = net->img_stale_direct = true;              // (weights change: the step image's records are behind)
        hipLaunchKernelGGL(k_stdp_csr_in, dim3(1024), dim3(64), 0, net->stream, ca);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        hipLaunchKernelGGL(k_stdp_csr_out, dim3(1024), dim3(64), 0, net->stream, ca);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    const unsigned sy = 64;   // spiking neurons processed concurrently; the rest grid-strides
    if (net->stdp_columns_form == 1) {
        // quad form: 256 listed columns per workgroup (64 per wavefront) x slabs of row groups
        const uint32_t groups = (net->n_tot + 3u) / 4u;
        const unsigned slabs = std::max(1u, std::min(1024u, groups / 16u));
        hipLaunchKernelGGL(k_stdp_columns_quads, dim3(4, slabs), dim3(256), 0, net->stream, a);
    } else {
        hipLaunchKernelGGL(k_stdp_columns, dim3((net->n_tot + 255) / 256, sy), dim3(256), 0, net->stream, a);
    }
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    hipLaunchKernelGGL(k_stdp_rows, dim3((net->n_loc + 255) / 256, sy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// sizes served by the one-launch small-lattice step (snn_kernels_resident.hpp)
bool fused_step_possible(const snn_network *net)
{
    // (a library carrying generated code has the one-launch step for ITS neuron model only: shorter compile)
    const bool compiled = !SNN_HAVE_CUSTOM_MODEL || (SNN_HAVE_CUSTOM_NEURON && net->model == SNN_MODEL_CUSTOM);
    return compiled && net->fused_step && !net->csr && !net->sharded && !net->drive_threshold && net->n_loc && net->n_tot &&
           net->n_chunks <= RESIDENT_MAX_CHUNKS && (size_t)net->n_tot * net->ld * 4 <= ((size_t)64 << 20);
}

// RewardModulatedLattice::update_weights_from_neurons for every modulated lattice (deferred form), as a standalone
// pass over the weights and traces.  `dop`: which dopamine slot of the modulator table applies.
int launch_rstdp_pass(snn_network *net, int dop)
{
    if (net->csr) {
        if (!net->csr_ptr) return SNN_OK;
        CsrRewardArgs a{};
        a.g = csr_graph(net); a.c = net->trace; a.q0 = net->q0; a.rows = net->rowmap; a.n_neurons = net->nn;
        a.last_firing_time = net->na.last_firing_time; a.lattice_slot = net->lattice_slot;
        a.rm = net->rm_dev; a.rm_on = net->rm_on_dev; a.dop = dop;
        net->img_stale = net->img_stale_direct = true;              // (weights change: the step image's records are behind)
        hipLaunchKernelGGL(k_rstdp_csr, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    RewardArgs a{};
    a.W = net->W; a.C = net->trace; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.n_neurons = net->nn;
    a.last_firing_time = net->na.last_firing_time; a.lattice_slot = net->lattice_slot;
    a.rm = net->rm_dev; a.rm_on = net->rm_on_dev; a.dop = dop;
    const unsigned gx = (net->n_loc + 255) / 256;
    const unsigned gy = std::max(1u, std::min<unsigned>((net->nn + 3) / 4, std::max(1u, 8192u / gx)));     // ~8192 workgroups in flight
    hipLaunchKernelGGL(k_rstdp_dense, dim3(gx, gy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// the connections between the lattices of a reward-modulated network (k_reward_cross), right after the lattices' own edges
static RewardCrossArgs reward_cross_args(snn_network *net)
{
    RewardCrossArgs a{};
    a.s = stdp_args(net);
    a.s.conn_kind = net->conn_kind_dev;
    a.C = net->trace; a.P = net->pending; a.K = net->edge_counter;
    a.rm = net->rm_dev; a.rm_on = net->rm_on_dev; a.bad = net->cross_bad;
    return a;
}

int launch_reward_cross(snn_network *net)
{
    if (!net->any_conn_kind || net->nn == 0 || !net->trace || !net->pending) return SNN_OK;
    if (net->csr) {
        if (!net->csr_ptr || net->n_loc == 0) return SNN_OK;
        CsrRewardCrossArgs c{};
        c.g = csr_graph(net);
        c.r = reward_cross_args(net);
        net->img_stale = net->img_stale_direct = true;              // (weights change: the step image's records are behind)
        hipLaunchKernelGGL(k_reward_cross_csr, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, c);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    const unsigned gx = (net->nn + 255) / 256;
    const unsigned gy = std::max(1u, std::min<unsigned>(net->n_tot, std::max(1u, 16384u / gx)));
    hipLaunchKernelGGL(k_reward_cross, dim3(gx, gy), dim3(256), 0, net->stream, reward_cross_args(net));
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// Refuses, before the first step, connection kinds that lie where the reference's visits unwrap None (k_reward_cross_check);
// repeated after anything that may change the answer (kinds, graph, plasticity, modulators).
int check_reward_cross(snn_network *net)
{
    if (!net->any_conn_kind || net->cross_checked || net->nn == 0 || !net->pending) return SNN_OK;
    HIP_TRY(hipMemsetAsync(net->cross_bad, 0, 4, net->stream), SNN_ERR_BUFFER_WRITE);
    const unsigned gx = (net->nn + 255) / 256;
    const unsigned gy = std::max(1u, std::min<unsigned>(net->n_tot, std::max(1u, 16384u / gx)));
    if (net->csr) {
        if (net->csr_ptr && net->n_loc) {
            CsrRewardCrossArgs c{};
            c.g = csr_graph(net);
            c.r = reward_cross_args(net);
            hipLaunchKernelGGL(k_reward_cross_check_csr, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, c);
        }
    } else {
        hipLaunchKernelGGL(k_reward_cross_check, dim3(gx, gy), dim3(256), 0, net->stream, reward_cross_args(net));
    }
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, net->cross_bad, 4, hipMemcpyDeviceToHost, net->stream), SNN_ERR_BUFFER_READ);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    static const char *why[] = {"", "a connection of a visited lattice has no reverse connection of the same kind: the reference's outgoing half "
        "looks the reverse connection up and unwraps it (neuron/mod.rs:4768-4771, 4929-4932)",
        "reward-modulated weights where no side has a modulator, or from a spike train into a plastic plain lattice (neuron/mod.rs:4743, 4789)",
        "plain weights between a plastic plain lattice and a reward-modulated one (neuron/mod.rs:4729-4733, 4778)",
        "a BCM lattice on a connection of a reward-modulated network (the reference's network has one plasticity rule)"};
    if (bad) return fail(SNN_ERR_BAD_STATE, why[bad < 5 ? bad : 1]);
    net->cross_checked = true;
    return SNN_OK;
}

// End of a step: dense handles only NOTE that the update is due -- the next step's input pass applies it while it
// streams W anyway (launch_inputs -> k_inputs_rstdp); sparse handles and the one-launch small-lattice step run the
// standalone pass right away.
int launch_reward_modulation(snn_network *net)
{
    if (!net->any_modulation || net->nn == 0 || net->n_loc == 0 || !net->trace) return SNN_OK;
    if (!net->csr && net->defer_rstdp && !net->any_whist && !fused_step_possible(net)) {
        net->rstdp_pending = true;
        net->reward_since_defer = false;
        return SNN_OK;
    }
    return launch_rstdp_pass(net, RM_DOPAMINE);
}

// The deferred STDP update as standalone passes (a host access to the weights, the end of a run, an input pass that
// cannot carry it).
int flush_stdp(snn_network *net)
{
    if (!net->stdp_pending) return SNN_OK;
    net->stdp_pending = false;
    if (net->n_loc == 0 || net->nn == 0) { net->stdp_pending_rows_only = false; return SNN_OK; }
    StdpArgs a = stdp_args(net);
    const unsigned sy = 64;
    if (net->stdp_pending_rows_only) {          // "defer_stdp" 3: the columns were scattered when the step closed
        net->stdp_pending_rows_only = false;
    } else {
        hipLaunchKernelGGL(k_stdp_apply_columns, dim3((net->n_tot + 255) / 256, sy), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    hipLaunchKernelGGL(k_stdp_apply_rows, dim3((net->n_loc + 255) / 256, sy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// Apply a deferred update now (a host access to weights, traces or firing times is about to happen).
int flush_rstdp(snn_network *net)
{
    if (!net->rstdp_pending) return SNN_OK;
    net->rstdp_pending = false;
    return launch_rstdp_pass(net, net->reward_since_defer ? RM_DOPAMINE_BEFORE : RM_DOPAMINE);
}

// The synapse matrix is the one allocation whose HBM placement matters: on MI355X two 17 GB allocations of one
// process can differ by 5-6 % in the sustained rate of the input pass (stable per allocation, different from
// process to process).  For matrices >= 1 GiB a second candidate is allocated while the first is held, the real
// kernel is timed on both (one warm + one timed pass each, once per handle) and the faster allocation is kept.
int time_input_pass(snn_network *net, float *ms)
{
    *ms = 0.0f;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(SNN_ERR_QUEUE, "hipEventCreate failed");
    // a pass over a matrix of a few GiB is short: several timed passes per candidate, or the 1 % decision threshold is noise
    const size_t bytes = (size_t)net->n_tot * net->ld * 4;
    const int passes = bytes < ((size_t)4 << 30) ? 8 : 1;
    int rc = launch_inputs(net);
    if (rc == SNN_OK && hipEventRecord(e0, net->stream) != hipSuccess) rc = fail(SNN_ERR_QUEUE, "hipEventRecord failed");
    for (int i = 0; i < passes && rc == SNN_OK; ++i) rc = launch_inputs(net);
    if (rc == SNN_OK && (hipEventRecord(e1, net->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                         hipEventElapsedTime(ms, e0, e1) != hipSuccess))
        rc = fail(SNN_ERR_WAIT, "placement timing failed");
    *ms /= (float)passes;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int choose_matrix_placement(snn_network *net)
{
    const size_t count = net->csr ? 0 : wcount(net->n_tot, net->ld);
    const size_t bytes = count * sizeof(float);
    if (bytes >= ((size_t)1 << 30) && net->n_loc) {
        const int prof = net->profile;
        net->profile = 0;
        float best_ms = 0.0f;
        int rc = time_input_pass(net, &best_ms);
        // up to four more candidates; every loser stays allocated until the end so that each new candidate is
        // forced into a different HBM region (a freed block would simply be handed out again)
        hvec<void *> losers;
        for (int cand = 0; cand < 4 && rc == SNN_OK; ++cand) {
            size_t free_b = 0, total_b = 0;
            void *b = nullptr;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (bytes >> 2) ||
                alloc_streamed(&b, bytes) != hipSuccess)
                break;
            float *a = net->W;
            float ms_b = 0.0f;
            net->W = static_cast<float *>(b);
            rc = time_input_pass(net, &ms_b);
            if (getenv("SNN_DEBUG_PLACEMENT"))
                fprintf(stderr, "[snn] matrix placement: held %p %.3f ms, candidate %p %.3f ms\n", (void *)a, best_ms, b, ms_b);
            if (rc == SNN_OK && ms_b < best_ms * 0.99f) {       // the candidate wins
                for (auto &p : net->allocs) if (p == a) p = b;
                net->alloc_bytes.erase(a);
                net->alloc_bytes[b] = bytes;
                losers.push_back(a);
                best_ms = ms_b;
            } else {
                net->W = a;
                losers.push_back(b);
            }
        }
        for (void *p : losers) (void)hipFree(p);
        net->profile = prof;
        if (rc != SNN_OK) return rc;
    }
    if (count) {   // no edges until a graph is set: every entry is the absent-edge sentinel
        hipLaunchKernelGGL(k_fill_u32, dim3(4096), dim3(256), 0, net->stream,
                           reinterpret_cast<uint32_t *>(net->W), count, 0x7FC00000u);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    }
    return SNN_OK;
}

// Small dense lattices on an unsharded handle: inputs + update in ONE launch (snn_kernels_resident.hpp).
bool fused_step_applies(const snn_network *net)
{
    return fused_step_possible(net) && !net->local_inputs_done;
}

// Arguments of a one-launch step: S(t) is read from the current shadow of the exchange buffer, S(t+1) goes to the
// exchange buffer and to the other shadow (flipped by the caller after the launch).
// in_place (the many-steps launch): everything reads and writes the exchange buffer itself.
int fused_step_args(snn_network *net, InputsArgs &a, UpdateArgs &u, bool in_place = false)
{
    const size_t xelems = (size_t)NUM_PLANES * net->xl.stride;
    if (in_place) {
        net->shadow_valid = false;
    } else if (!net->shadow[0]) {
        TRY(dev_alloc_t(net, &net->shadow[0], xelems));
        TRY(dev_alloc_t(net, &net->shadow[1], xelems));
        net->shadow_valid = false;
    }
    if (!in_place && !net->shadow_valid) {
        // both shadows: entries the step never rewrites (absent transmitter types, padding) must agree everywhere
        for (int i = 0; i < 2; ++i)
            HIP_TRY(hipMemcpyAsync(net->shadow[i], net->xbuf, xelems * 4, hipMemcpyDeviceToDevice, net->stream),
                    SNN_ERR_BUFFER_WRITE);
        net->shadow_valid = true;
        net->stat_shadow_refreshes += 1;
    }
    float *cur = in_place ? net->xbuf : net->shadow[net->shadow_cur];
    float *next = in_place ? nullptr : net->shadow[net->shadow_cur ^ 1];
    a = InputsArgs{};
    a.chunk_first = 0; a.hole_begin = net->n_chunks; a.hole_count = 0;
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.rows = net->rowmap; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = cur; a.xl = net->xl; a.gap_conductance = net->na.gap_conductance; a.uni = net->uni_neuron;
    a.st_value = net->ca.presyn_value; a.st_last_firing_time = net->ca.last_firing_time;
    a.st_view = net->cell_view[net->cell_view_cur];
    a.st_nt_t = net->ca.nt_t; a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
    a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_chunks = net->n_chunks;
    for (int k = 0; k < K_TYPES; ++k) a.live_type[k] = (uint32_t)k;
    u = UpdateArgs{};
    u.n = net->na;
    u.n.xbuf = cur;
    u.part_i = net->part_i; u.part_t = net->part_t; u.n_in = net->n_in; u.tcount = net->tcount;
    u.ld = net->ld; u.n_chunks = net->n_chunks; u.q0 = net->q0; u.n_loc = net->n_loc; u.rows = net->rowmap;
    u.clock = net->clock;
    u.electrical = net->electrical; u.chemical = net->chemical; u.nt_kind = net->nt_kind; u.rc_kind = net->rc_kind;
    u.vhist_row = (record_now(net) && net->want_vhist && net->vhist) ? net->vhist + (size_t)net->hist_steps * net->n_pad : nullptr;
    u.spike_row = (record_now(net) && net->want_raster && net->raster) ? net->raster + (size_t)net->hist_steps * (net->n_pad / 64) : nullptr;
    u.spike_counts = net->want_counts ? net->spike_counts : nullptr;
    u.xout = net->xbuf; u.xout2 = next;
    u.has_nt = net->any_nt_neurons ? 1 : 0;
    u.live_mask = net->live_mask_applied;
    u.bcm = net->model == SNN_MODEL_BCM_IZHIKEVICH;
    u.model_is_custom = net->model == SNN_MODEL_CUSTOM;
    return SNN_OK;
}

// HIP events around the launch that streams the graph (snn_profile_*)
int profile_open(snn_network *net, hipEvent_t *e1)
{
    *e1 = nullptr;
    if (!net->profile) return SNN_OK;
    if (net->ev_used == net->ev_pool.size()) {
        hipEvent_t x, y;
        HIP_TRY(hipEventCreate(&x), SNN_ERR_QUEUE);
        HIP_TRY(hipEventCreate(&y), SNN_ERR_QUEUE);
        net->ev_pool.emplace_back(x, y);
    }
    hipEvent_t e0 = net->ev_pool[net->ev_used].first;
    *e1 = net->ev_pool[net->ev_used].second;
    net->ev_counts.resize(net->ev_pool.size(), 1);
    net->ev_counts[net->ev_used] = 1;
    ++net->ev_used;
    HIP_TRY(hipEventRecord(e0, net->stream), SNN_ERR_QUEUE);
    return SNN_OK;
}

int launch_step_resident(snn_network *net)
{
    ResidentArgs r{};
    TRY(fused_step_args(net, r.in, r.up));
    hipEvent_t e1 = nullptr;
    TRY(profile_open(net, &e1));
    // a chunk's rows over four wavefronts (k_step_resident_q) where the workgroup stays within 512 threads: at most two chunks
    // (a network of at most 64 rows has one quarter's worth of them: nothing to spread, and the turns cost 1 - 8 us)
    const bool quarters = net->resident_quarters && net->n_chunks <= 2 && net->n_tot > 64;
    const dim3 grid((net->n_loc + 63) / 64), block(64 * net->n_chunks * (quarters ? 4 : 1));
#define SNN_RESIDENT(M)                                                                                              \
    do {                                                                                                             \
        if (quarters && net->electrical && net->chemical) hipLaunchKernelGGL((k_step_resident_q<M, true, true>), grid, block, 0, net->stream, r); \
        else if (quarters && net->electrical) hipLaunchKernelGGL((k_step_resident_q<M, true, false>), grid, block, 0, net->stream, r);            \
        else if (quarters) hipLaunchKernelGGL((k_step_resident_q<M, false, true>), grid, block, 0, net->stream, r);                              \
        else if (net->electrical && net->chemical) hipLaunchKernelGGL((k_step_resident<M, true, true>), grid, block, 0, net->stream, r);  \
        else if (net->electrical) hipLaunchKernelGGL((k_step_resident<M, true, false>), grid, block, 0, net->stream, r);             \
        else hipLaunchKernelGGL((k_step_resident<M, false, true>), grid, block, 0, net->stream, r);                                  \
    } while (0)
#if !SNN_HAVE_CUSTOM_MODEL        // a library carrying a generated model has the one-launch step for that model only (shorter compile)
    SNN_FOR_MODEL(SNN_RESIDENT)
#elif SNN_HAVE_CUSTOM_NEURON
    SNN_RESIDENT(CUSTOM_MODEL);
#else
    (void)grid, (void)block;
#endif
#undef SNN_RESIDENT
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (e1) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    net->shadow_cur ^= 1;
    return SNN_OK;
}

// Streamed dense matrices on an unsharded handle: the input pass closes the step itself (k_inputs_dense_close) -- no pending weight
// update to carry, no drive kernel in front, gap junctions on (the chemical-only pass keeps its two kernels).
bool dense_close_applies(const snn_network *net)
{
    return !SNN_HAVE_CUSTOM_MODEL && net->dense_close && !net->csr && !net->sharded && !net->drive_threshold && net->n_loc && net->n_tot &&
           net->n_loc == net->nn && net->electrical && matrix_streamed(net) && !net->local_inputs_done && !net->stdp_pending &&
           !net->rstdp_pending && net->force_shape == 0 && net->n_chunks <= net->dense_close_max_chunks;
}

int launch_dense_close(snn_network *net)
{
    DenseStepArgs r{};
    TRY(fused_step_args(net, r.in, r.up));
    const uint64_t waves4 = (uint64_t)((net->n_loc + 255) / 256) * net->n_chunks;
    const int shape = waves4 < 57600 ? 2 : 1;                                  // as launch_inputs chooses
    const uint32_t tiles = shape == 1 ? (net->n_loc + InputsShape<1>::TILE - 1) / InputsShape<1>::TILE
                                      : (net->n_loc + InputsShape<2>::TILE - 1) / InputsShape<2>::TILE;
    if (net->tile_done_len < tiles) {
        net->tile_done = nullptr;                                              // (a handle's width is fixed after finalize: allocated once)
        TRY(dev_alloc_t(net, &net->tile_done, (size_t)tiles));
        HIP_TRY(memset_sync(net, net->tile_done, 0, (size_t)tiles * 4), SNN_ERR_BUFFER_WRITE);
        net->tile_done_len = tiles;
        TRY(fused_step_args(net, r.in, r.up));                                 // (nothing moved; the arguments are cheap to rebuild)
    }
    r.tile_done = net->tile_done;
    // one live transmitter type: the pass specialised on it (the planes of the other types hold zeros, which the update adds)
    const bool one_type = net->chemical && net->n_live == 1;
    if (one_type) for (int k = 0; k < K_TYPES; ++k) r.in.live_type[k] = net->live_type[k];
    hipEvent_t e1 = nullptr;
    TRY(profile_open(net, &e1));
    const dim3 grid(tiles, net->n_chunks), block(256);
#define SNN_DENSE_CLOSE_SHAPE(M, SH)                                                                                                  \
    do {                                                                                                                              \
        if (!net->chemical) hipLaunchKernelGGL((k_inputs_dense_close<M, true, false, SH, 3>), grid, block, 0, net->stream, r);        \
        else if (one_type) hipLaunchKernelGGL((k_inputs_dense_close<M, true, true, SH, 1>), grid, block, 0, net->stream, r);          \
        else hipLaunchKernelGGL((k_inputs_dense_close<M, true, true, SH, 3>), grid, block, 0, net->stream, r);                        \
    } while (0)
#define SNN_DENSE_CLOSE(M)                                                                                                            \
    do {                                                                                                                              \
        if (shape == 1) SNN_DENSE_CLOSE_SHAPE(M, 1);                                                                                  \
        else SNN_DENSE_CLOSE_SHAPE(M, 2);                                                                                             \
    } while (0)
#if !SNN_HAVE_CUSTOM_MODEL
    SNN_FOR_MODEL(SNN_DENSE_CLOSE)
#else
    (void)grid, (void)block;
#endif
#undef SNN_DENSE_CLOSE
#undef SNN_DENSE_CLOSE_SHAPE
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (e1) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    net->shadow_cur ^= 1;
    return SNN_OK;
}

// Small lattices of neurons: ALL steps of a run call in one launch (k_run_resident) when nothing has to happen between two
// steps on the host's side of the stream -- no weight updates, no per-step reductions, every step recorded (or none), cells
// (if any) Poisson or Rate without transmitters.  Electrical synapses: up to 4096 rows; with chemical synapses (built-in
// kinetics): one row group, up to 1024 rows.
bool run_resident_shape(const snn_network *net);
bool run_resident_stdp_ok(const snn_network *net);
int launch_plasticity(snn_network *net);
bool run_resident_applies(const snn_network *net)
{
    // the run's outcome is read after a host synchronisation (snn_run): not on a caller's stream
    return run_resident_shape(net) && !net->external_stream;
}
// STDP inside the run: electrical synapses, neurons only, one row group, at most four lattices, the STDP rule (no BCM), no
// connection kinds of a reward-modulated network, the scatter form of the weight update (defer_stdp 0)
bool run_resident_stdp_ok(const snn_network *net)
{
    if (!net->persistent_stdp || net->chemical || net->nc || net->n_tot > RUN_RESIDENT_GROUP_ROWS || net->any_conn_kind || net->defer_stdp ||
        net->lattices.size() > RUN_STDP_MAX_LATTICES)
        return false;
    for (size_t l = 0; l < net->lattices.size(); ++l)
        if (net->plast_host[l] && net->stdp_host[l * PL_STRIDE + 5] != 0.0f) return false;
    return true;
}
bool run_resident_shape(const snn_network *net)
{
    const bool chem_ok = !net->chemical || (net->n_tot <= RUN_RESIDENT_GROUP_ROWS && net->persistent_chem &&
                                            net->nt_kind != SNN_NT_CUSTOM && net->rc_kind != SNN_RC_CUSTOM);
    return !SNN_HAVE_CUSTOM_MODEL && net->fused_step && !net->csr && !net->sharded && !net->drive_threshold && net->n_loc &&
           net->persistent_run && net->n_loc == net->nn && net->nn + net->nc == net->n_tot &&
           (net->nc == 0 || ((net->st_kind == SNN_ST_POISSON || net->st_kind == SNN_ST_RATE) && !net->any_nt_cells &&
                             !net->cell_list_dev && !SNN_HAVE_CUSTOM_REFRACTORINESS)) &&
           net->n_tot <= RUN_RESIDENT_MAX_NEURONS && (net->electrical || net->chemical) && chem_ok &&
           (!net->any_plasticity || run_resident_stdp_ok(net)) &&
           !net->any_modulation && !net->any_whist && !net->want_avg && !net->want_eeg && net->hist_every == 1 &&
           net->model != SNN_MODEL_BCM_IZHIKEVICH && !net->local_inputs_done;
}

// chunk sums travelling between the row groups of a tile: [2][tiles][groups - 1][4 chunks][64] granules
constexpr size_t RUN_GRANULE_WORDS = (size_t)2 * RUN_GRANULE_PLANES * RUN_RESIDENT_MAX_NEURONS;
constexpr size_t RUN_PARTIAL_WORDS = (size_t)2 * RUN_RESIDENT_MAX_TILES * (RUN_RESIDENT_MAX_GROUPS - 1) * 256;

// What a one-launch run may overwrite is every SMALL array of the handle: per-neuron and per-cell state, the exchange
// buffer and its shadows, spike totals, the device clocks (the matrix, the partial planes and the histories are either
// read-only here, scratch, or rewritten by a repeated step).  "Small" = no larger than the exchange buffer / a typed
// [3][pad] block.  (Re)built when the handle has allocated since.
int run_snapshot(snn_network *net, bool restore)
{
    if (!restore && (!net->snap_table || net->snap_allocs_seen != net->allocs.size())) {
        const size_t limit = 4 * std::max<size_t>({(size_t)NUM_PLANES * net->xl.stride, (size_t)K_TYPES * net->n_pad,
                                                   (size_t)K_TYPES * net->c_pad, 256});
        if (net->alloc_bytes.count(net->xbuf) == 0 || net->alloc_bytes[net->xbuf] > limit)
            return fail(SNN_ERR_BAD_STATE, "exchange buffer missing from the snapshot set");
        hvec<CopyEntry> table;
        size_t words = 0;
        uint32_t max_words = 0;
        for (const auto &kv : net->alloc_bytes) {
            if (kv.second > limit || kv.first == net->snap_table || kv.first == net->snap_buf ||
                kv.first == net->run_granules || kv.first == net->run_partials || kv.first == net->run_timing || kv.first == net->verify_buf)
                continue;
            const uint32_t w = (uint32_t)(kv.second / 4);
            // pad = 1: contents depend on the order of atomics (the compacted spike list) -- copied, but not compared by "verify"
            table.push_back(CopyEntry{static_cast<uint32_t *>(kv.first), nullptr, w, kv.first == (void *)net->spike_list ? 1u : 0u});
            words += w;
            max_words = std::max(max_words, w);
        }
        // (the two buffers are replaced, not grown: a handle allocates a handful of times in its life)
        for (void *old : {(void *)net->snap_table, (void *)net->snap_buf})
            if (old) {
                (void)hipFree(old);
                net->alloc_bytes.erase(old);
                net->allocs.erase(std::remove(net->allocs.begin(), net->allocs.end(), old), net->allocs.end());
            }
        net->snap_table = nullptr; net->snap_buf = nullptr;
        TRY(dev_alloc_t(net, &net->snap_buf, words));
        net->snap_words = words;
        TRY(dev_alloc_t(net, &net->snap_table, table.size()));
        size_t off = 0;
        for (auto &e : table) { e.dst = net->snap_buf + off; off += e.words; }
        HIP_TRY(hipMemcpyAsync(net->snap_table, table.data(), table.size() * sizeof(CopyEntry), hipMemcpyHostToDevice, net->stream),
                SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);       // `table` leaves scope
        net->snap_table_host = table;
        net->snap_generation += 1;
        net->snap_entries = (uint32_t)table.size();
        net->snap_max_words = max_words;
        net->snap_allocs_seen = net->allocs.size();
    }
    if (!net->snap_entries) return SNN_OK;
    const uint32_t bx = std::max(1u, std::min(16u, (net->snap_max_words + 1023u) / 1024u));
    hipLaunchKernelGGL(k_copy_table, dim3(bx, net->snap_entries), dim3(256), 0, net->stream, net->snap_table, restore ? 1 : 0);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

int launch_run_resident(snn_network *net, uint64_t iterations, uint64_t steps_before = 0)
{
    if (!net->run_granules) {
        TRY(dev_alloc_t(net, &net->run_granules, RUN_GRANULE_WORDS));
        HIP_TRY(hipMemsetAsync(net->run_granules, 0, RUN_GRANULE_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
        TRY(dev_alloc_t(net, &net->run_partials, RUN_PARTIAL_WORDS));
        HIP_TRY(hipMemsetAsync(net->run_partials, 0, RUN_PARTIAL_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(host_malloc(reinterpret_cast<void **>(&net->run_failed), 8, hipHostMallocMapped), SNN_ERR_BUFFER_CREATE);
        net->run_failed[0] = net->run_failed[1] = 0u;
        net->run_tag = 1;
    }
    // workgroups: column tiles x row groups of 1024 rows (one group up to 1024 neurons)
    const uint32_t row_groups = (net->n_tot + RUN_RESIDENT_GROUP_ROWS - 1) / RUN_RESIDENT_GROUP_ROWS;
    const uint32_t n_groups = (net->n_loc + 63) / 64 * row_groups;
    if (net->run_probed_grid != n_groups) {
        // once per handle and grid size: can that many workgroups of this shape be resident together?  (the granule words
        // double as the probe's counter: they are cleared again before any run uses them)
        uint32_t *counter = reinterpret_cast<uint32_t *>(net->run_granules);
        HIP_TRY(hipMemsetAsync(counter, 0, 8, net->stream), SNN_ERR_BUFFER_WRITE);
        hipLaunchKernelGGL(k_run_resident_probe, dim3(n_groups), dim3(1024), 0, net->stream, counter, net->run_failed + 1);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(hipMemsetAsync(counter, 0, 8, net->stream), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
        if (net->run_failed[1]) {
            net->run_failed[1] = 0u;
            net->persistent_run = 0;                 // one launch per step for this handle from now on
            return SNN_OK;
        }
        net->run_probed_grid = n_groups;
    }
    {
        // ONE chunk per call (at most RUN_RESIDENT_CHUNK_STEPS steps): the caller takes its snapshot per chunk, so a chunk that
        // gives up is rolled back to ITS start -- the weights and counters the earlier chunks committed stay
        const uint32_t steps = (uint32_t)iterations;
        if (net->run_tag > 0x7FFFFFFFu - steps - 2u) {           // tags would wrap (bit 31 carries a spike): start over on clean slots
            HIP_TRY(hipMemsetAsync(net->run_granules, 0, RUN_GRANULE_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
            HIP_TRY(hipMemsetAsync(net->run_partials, 0, RUN_PARTIAL_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
            net->run_tag = 1;
        }
        ResidentRunArgs r{};
        TRY(fused_step_args(net, r.in, r.up, /*in_place=*/true));
        r.steps = steps;
        r.vhist_stride = net->n_pad;
        r.raster_stride = net->n_pad / 64;
        r.granules = net->run_granules;
        r.partials = net->run_partials;
        r.n_groups = row_groups;
        r.tag_base = net->run_tag;
        r.failed = net->run_failed;
        r.spin_limit = net->run_spin_limit;
        // (the test hook counts steps of the RUN CALL: a fault can be placed in a later chunk of it)
        r.fault_step = (net->run_fault_step > steps_before && net->run_fault_step <= steps_before + steps) ? (uint32_t)(net->run_fault_step - steps_before) : 0u;
        r.cells = net->ca;
        r.st_kind = net->st_kind;
        r.lattice_clock = net->st_clock_dev;
        r.step_offset0 = net->run_step_offset;
        r.view_clock0 = net->clock;
        r.st_vhist_row = (recording(net) && net->want_vhist && net->st_vhist) ? net->st_vhist + (size_t)net->hist_steps * net->c_pad : nullptr;
        r.st_vhist_stride = net->c_pad;
        if (!net->run_timing && (net->run_timing_opt || getenv("SNN_AMD_RUN_TIMING"))) TRY(dev_alloc_t(net, &net->run_timing, (size_t)RUN_RESIDENT_MAX_TILES * RUN_RESIDENT_MAX_GROUPS * 4));
        r.timing = net->run_timing;
        // chemical synapses: the transmitter types some NEURON releases travel (cells with transmitters keep the per-step forms)
        if (net->chemical)
            for (uint32_t k = 0; k < K_TYPES; ++k)
                if (net->live_mask_applied != 0xFFFFFFFFu && (net->live_mask_applied >> k & 1u)) r.live_type[r.n_live++] = k;
        const bool stdp = net->any_plasticity;                  // (run_resident_shape let it through: run_resident_stdp_ok)
        if (stdp) {
            if (!net->run_w_out) TRY(dev_alloc_t(net, &net->run_w_out, wcount(net->n_tot, net->ld)));
            r.stdp_table = net->stdp_dev; r.stdp_on = net->plast_dev; r.stdp_lattice = net->lattice_slot;
            r.stdp_lattices = (uint32_t)net->lattices.size();
            r.w_out = net->run_w_out;
        }
        hipLaunchKernelGGL(k_run_resident_seed, dim3((net->nn + 255) / 256), dim3(256), 0, net->stream, net->xbuf, net->xl,
                           net->nn, net->run_granules, r.tag_base, net->na.nt_flags, net->n_pad, r.n_live, r.live_type[0],
                           r.live_type[1], r.live_type[2]);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        // profiling: one event pair around the launch, counted as `steps` passes over the graph
        hipEvent_t e1 = nullptr;
        TRY(profile_open(net, &e1));
        if (e1) net->ev_counts[net->ev_used - 1] = (int)steps;
        const dim3 grid(n_groups), block(1024);
#define SNN_RUN_RESIDENT(M) hipLaunchKernelGGL((k_run_resident<M, false, false>), grid, block, 0, net->stream, r)
#define SNN_RUN_RESIDENT_CELLS(M) hipLaunchKernelGGL((k_run_resident<M, false, true>), grid, block, 0, net->stream, r)
#define SNN_RUN_RESIDENT_CHEM(M) hipLaunchKernelGGL((k_run_resident<M, false, false, true>), grid, block, 0, net->stream, r)
#define SNN_RUN_RESIDENT_CHEM_CELLS(M) hipLaunchKernelGGL((k_run_resident<M, false, true, true>), grid, block, 0, net->stream, r)
#define SNN_RUN_RESIDENT_CHEM_LEND(M) hipLaunchKernelGGL((k_run_resident<M, false, false, true, false, true>), grid, block, 0, net->stream, r)
#define SNN_RUN_RESIDENT_STDP(M) hipLaunchKernelGGL((k_run_resident<M, false, false, false, true>), grid, block, 0, net->stream, r)
#if !SNN_HAVE_CUSTOM_MODEL
        // neuron state in registers for the whole run where the kernel carries the model's update itself
        const bool regs = !r.up.has_nt && !r.up.bcm && !net->chemical;
        if (stdp && regs && net->model == SNN_MODEL_IZHIKEVICH) {                  // weight updates inside the run
            hipLaunchKernelGGL((k_run_resident<0, true, false, false, true>), grid, block, 0, net->stream, r);
        } else if (stdp) {
            SNN_FOR_MODEL(SNN_RUN_RESIDENT_STDP);
        } else if (net->chemical && net->nc) {                                     // chemical synapses: the generic update
            SNN_FOR_MODEL(SNN_RUN_RESIDENT_CHEM_CELLS);
        } else if (net->chemical && net->electrical && net->n_tot <= CHUNK && net->model == SNN_MODEL_IZHIKEVICH) {
            // (one chunk at most, both kinds of synapse: the idle wavefronts take the transmitter chains)
            hipLaunchKernelGGL((k_run_resident<0, true, false, true, false, true>), grid, block, 0, net->stream, r);
        } else if (net->chemical && net->electrical && net->n_tot <= CHUNK) {
            SNN_FOR_MODEL(SNN_RUN_RESIDENT_CHEM_LEND);
        } else if (net->chemical && net->model == SNN_MODEL_IZHIKEVICH) {          // ... Izhikevich: receptors and transmitters resident too
            hipLaunchKernelGGL((k_run_resident<0, true, false, true>), grid, block, 0, net->stream, r);
        } else if (net->chemical) {
            SNN_FOR_MODEL(SNN_RUN_RESIDENT_CHEM);
        } else if (net->nc && regs && net->model == SNN_MODEL_IZHIKEVICH) {        // rows that are spike-train cells
            hipLaunchKernelGGL((k_run_resident<0, true, true>), grid, block, 0, net->stream, r);
        } else if (net->nc && regs && net->model == SNN_MODEL_LIF) {
            hipLaunchKernelGGL((k_run_resident<1, true, true>), grid, block, 0, net->stream, r);
        } else if (net->nc && regs && net->model == SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE) {
            hipLaunchKernelGGL((k_run_resident<3, true, true>), grid, block, 0, net->stream, r);
        } else if (net->nc && regs && net->model == SNN_MODEL_SIMPLE_LIF) {
            hipLaunchKernelGGL((k_run_resident<4, true, true>), grid, block, 0, net->stream, r);
        } else if (net->nc) {
            SNN_FOR_MODEL(SNN_RUN_RESIDENT_CELLS);
        } else if (regs && net->model == SNN_MODEL_IZHIKEVICH) {
            hipLaunchKernelGGL((k_run_resident<0, true, false>), grid, block, 0, net->stream, r);
        } else if (regs && net->model == SNN_MODEL_LIF) {
            hipLaunchKernelGGL((k_run_resident<1, true, false>), grid, block, 0, net->stream, r);
        } else if (regs && net->model == SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE) {
            hipLaunchKernelGGL((k_run_resident<3, true, false>), grid, block, 0, net->stream, r);
        } else if (regs && net->model == SNN_MODEL_SIMPLE_LIF) {
            hipLaunchKernelGGL((k_run_resident<4, true, false>), grid, block, 0, net->stream, r);
        } else {
            SNN_FOR_MODEL(SNN_RUN_RESIDENT);
        }
#else
        (void)grid, (void)block;
#endif
#undef SNN_RUN_RESIDENT
#undef SNN_RUN_RESIDENT_CELLS
#undef SNN_RUN_RESIDENT_CHEM
#undef SNN_RUN_RESIDENT_CHEM_CELLS
#undef SNN_RUN_RESIDENT_STDP
#undef SNN_RUN_RESIDENT_CHEM_LEND
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        if (e1) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
        if (net->run_timing) {          // debugging aid: workgroup 0's phases in shader clocks per step
            hvec<unsigned long long> t((size_t)grid.x * 4);
            HIP_TRY(hipMemcpyAsync(t.data(), net->run_timing, t.size() * 8, hipMemcpyDeviceToHost, net->stream), SNN_ERR_BUFFER_READ);
            HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
            for (int k = 0; k < 4; ++k) net->run_timing_last[k] = t[k];
            net->run_timing_steps = steps;
            for (unsigned b = 0; getenv("SNN_AMD_RUN_TIMING") && b < grid.x; b += (grid.x > 1 ? grid.x - 1 : 1))
                fprintf(stderr, "k_run_resident workgroup %u: poll %.0f, barrier %.0f, turns %.0f, update+publish %.0f clocks/step (%u steps)\n",
                        b, (double)t[b * 4] / steps, (double)t[b * 4 + 1] / steps, (double)t[b * 4 + 2] / steps, (double)t[b * 4 + 3] / steps, steps);
        }
        if (stdp) {
            // The workgroups applied the weight updates of every step but the last and left their weights in run_w_out -- unless
            // the run gave up, in which case W is as it was and the caller rolls the handle back.  On success: W <- run_w_out, then
            // the last step's updates with the plain kernels (they read the spike flags and firing times the run left).
            HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
            if (net->run_failed[0]) return SNN_OK;
            HIP_TRY(hipMemcpyAsync(net->W, net->run_w_out, (size_t)4 * ((net->n_tot + 3) / 4) * net->ld * 4, hipMemcpyDeviceToDevice, net->stream),
                    SNN_ERR_BUFFER_WRITE);
            net->clock += steps - 1;
            TRY(launch_plasticity(net));
            net->clock -= steps - 1;
            net->stat_run_stdp_steps += steps;
        }
        net->run_tag += steps;
        net->stat_run_launches += 1;
        net->stat_run_steps += steps;
        net->clock += steps;
        net->run_step_offset += steps;
        if (recording(net)) { net->hist_steps += steps; net->hist_tick += steps; }
    }
    return SNN_OK;
}

// Sparse handles: row sums + neuron update in one launch (k_step_csr).  Shard handles too: what arrives from the peers
// is written into the shadow the next step reads as well (wire_args / k_step_close).
bool fused_csr_step_applies(const snn_network *net)
{
    return !SNN_HAVE_CUSTOM_MODEL && net->fused_step && net->csr && net->csr_ptr && !net->drive_threshold &&
           net->n_loc && !net->local_inputs_done &&
           (!net->sharded || net->n_shards == 1 || net->x_mode == SNN_EXCHANGE_HALO);
}

// ... and, when nothing has to happen between the unpack and the spike trains (no weight updates, no per-lattice
// reductions), the whole step of a shard handle is: k_step_csr over the border slices, packing as it goes -> [exchange]
// -> k_step_csr over the interior slices -> k_step_close (cells + unpack + clearing the outgoing bitmaps).
bool csr_fast_step(const snn_network *net)
{
    return fused_csr_step_applies(net) && net->sharded && net->x_mode == SNN_EXCHANGE_HALO && !net->any_plasticity &&
           !net->any_modulation && !net->want_avg && !net->want_eeg && !net->any_whist;
}

// The spike-train cells advance inside the step's (last) k_step_csr launch: nothing may read a cell's own arrays between the
// neuron update and the cells' iteration (weight updates do: last_firing_time), and the rows must read the cells through
// the two-copy view only (chemical synapses read the cells' transmitter planes directly).
bool cells_ride_allowed(const snn_network *net)         // (before the first run the two-copy view is not allocated yet)
{
    return net->nc && net->electrical && !net->chemical && !net->any_plasticity && !net->any_modulation &&
           net->cells_in_step && (!net->sharded || net->n_shards == 1 || csr_fast_step(net));
}
bool cells_ride_with_rows(const snn_network *net) { return net->cell_view[0] && cells_ride_allowed(net); }

enum CsrStepPart { CSR_STEP_ALL = 0, CSR_STEP_BORDER = 1, CSR_STEP_INTERIOR = 2 };

int launch_step_close(snn_network *net, bool cells, bool unpack);
WireArgs wire_args(snn_network *net, int which, int set = 0);

int launch_step_csr(snn_network *net, CsrStepPart part = CSR_STEP_ALL, bool pack = false)
{
    CsrStepArgs c{};
    TRY(fused_step_args(net, c.c.in, c.up));
    c.c.g = csr_graph(net);
    uint32_t waves = c.c.g.n_slices;
    if (part == CSR_STEP_BORDER) { c.slice_list = net->csr_border_dev; c.n_listed = waves = net->n_border; }
    if (part == CSR_STEP_INTERIOR) { c.slice_list = net->csr_interior_dev; c.n_listed = waves = net->n_interior; }
    if (net->direct_run) {       // the halo is gathered from the received segments of the previous step
        c.c.g.plan = net->csr_plan_direct;
        c.c.g.halo = net->hx_par ? net->halo_recv_buf : net->halo_recv_buf2;
        c.c.g.halo_base = net->nn + net->nc;
    }
    if (net->peer_run) {         // ... which the peers stored as granules into this handle's set of the previous step's parity
        c.c.g.halo64 = net->p2p_recv[(net->p2p_epoch + 1u) & 1u];
        c.c.g.halo_tag = net->p2p_epoch;
        c.c.g.spin_limit = net->p2p_spin_limit;
        c.c.g.failed = PeerFailure{{net->p2p_failed, net->p2p_done_blocks + 1}};
        // a halo neuron's granules are adjacent, one per plane of the plan, in the plan's order
        c.c.g.halo_slot_v = 0xFFFFFFFFu;
        for (int k = 0; k < K_TYPES; ++k) c.c.g.halo_slot_t[k] = 0xFFFFFFFFu;
        for (uint32_t s = 0; s < net->x_planes; ++s) {
            if (net->x_plane_id[s] == (uint32_t)PLANE_V) c.c.g.halo_slot_v = s;
            else c.c.g.halo_slot_t[net->x_plane_id[s] - PLANE_T0] = s;
        }
        c.c.g.delay = net->peer_delay; c.c.g.delay_seed = net->shard_index * 7919u + 1u;
    }
    if (pack) {
        c.pack.ptr = net->pack_ptr_dev; c.pack.seg_off = net->pack_segoff_dev; c.pack.seg_count = net->pack_count_dev;
        c.pack.index = net->pack_index_dev; c.pack.planes = net->x_planes;
        c.pack.buf = (net->direct_run && net->hx_par) ? net->halo_send_buf2 : net->halo_send_buf;
        for (int s = 0; s < WIRE_MAX_PLANES; ++s) c.pack.plane_id[s] = net->x_plane_id[s];
        if (net->peer_run) {
            c.pack.dst = net->p2p_dst_dev[net->p2p_epoch & 1u];
            c.pack.peer = net->p2p_peer_dev;
            c.pack.flags = net->p2p_flags;
            // the set was last read by the peer's step of epoch - 1 (rows and mirror job): its counter says when that is over
            c.pack.tag_out = net->p2p_epoch + 1u; c.pack.need_done = net->p2p_epoch - 1u;
            c.pack.spin_limit = net->p2p_spin_limit; c.pack.failed = PeerFailure{{net->p2p_failed, net->p2p_done_blocks + 1}};
            c.pack.nt_flags = net->na.nt_flags; c.pack.n_pad = net->n_pad;
            c.pack.delay = net->peer_delay; c.pack.delay_seed = net->shard_index * 7919u + 2u;
        }
    }
    if (net->peer_run) {
        c.peer.signal = net->p2p_signal_dev; c.peer.n_signal = net->p2p_n_signal;
        c.peer.done_value = net->p2p_epoch - 1u;     // this launch running = the step of the epoch before is over
        c.peer.delay = net->peer_delay; c.peer.delay_seed = net->shard_index * 7919u + 3u;
    }
    // the cells ride with the step's last row launch (the rows of BOTH launches read the view the cells do not write)
    const bool last_part = part != CSR_STEP_BORDER || net->n_interior == 0;
    if (last_part && !net->cells_stepped && cells_ride_with_rows(net)) {
        c.tail.cell_blocks = (spike_train_args(net, c.tail.cells, 1, net->run_step_offset, net->clock + 1) + 255) / 256;
        net->cells_stepped = true;
    }
    if (last_part && net->direct_run && !net->tail_done) {
        net->tail_done = true;
        // behind the rows: the previous step's arrivals into the mirror (+ their last_firing_time stamps), and the bitmaps of
        // the outgoing set the NEXT step packs into (its last send completed before this step's first launch)
        if (net->stamp_pending && net->seg_n[1] && net->recv_total) {
            c.tail.recv = wire_args(net, 1, net->hx_par ^ 1);
            c.tail.recv.clock = net->clock - 1;
            c.tail.recv.xbuf2 = nullptr;
            c.tail.recv_total = net->recv_total;
            c.tail.recv_segments = net->seg_n[1];
            c.tail.unpack_blocks = (net->recv_total + 255) / 256;
            if (net->peer_run) {         // the same set the rows of this launch read
                c.tail.recv64 = net->p2p_recv[(net->p2p_epoch + 1u) & 1u]; c.tail.recv_tag = net->p2p_epoch;
                c.tail.spin_limit = net->p2p_spin_limit; c.tail.failed = PeerFailure{{net->p2p_failed, net->p2p_done_blocks + 1}};
            }
        }
        net->stamp_pending = false;
        if (net->send_bitmap_words && !net->peer_run) {
            c.tail.send = wire_args(net, 0, net->hx_par ^ 1);
            c.tail.send_segments = net->seg_n[0];
            c.tail.send_bitmap_words = net->send_bitmap_words;
        }
    }
    const uint32_t tail_blocks = c.tail.blocks();
    c.xcd_bands = net->csr_xcd_bands ? 1u : 0u;
    // the step image (static weights, gap junctions only, every source in this handle's own arrays): records of 16 bytes and the
    // slices' presynaptic windows in LDS
    // ... shard handles too: border and interior launches, the border rows packing as they go; in a direct run (the halo gathered
    // from the received segments) the image is the one built with the exchange plan -- a halo neuron's source is its word of the
    // receive buffer.  Not the peer form (a granule is polled, not copied).
    const bool direct_image = net->direct_run && !net->peer_run;
    const bool image = net->csr_image && !net->peer_run && net->electrical && !net->chemical && !net->any_plasticity && !net->any_modulation &&
                       !net->any_conn_kind && net->model != SNN_MODEL_CUSTOM && (net->nc == 0 || c.c.in.st_view) &&
                       (direct_image ? net->csr_img_hdr_direct != nullptr : (net->csr_img_hdr != nullptr && !net->direct_run));
    if (image) {
        bool &stale = direct_image ? net->img_stale_direct : net->img_stale;
        uint32_t *hdr = direct_image ? net->csr_img_hdr_direct : net->csr_img_hdr;
        uint4 *rec = direct_image ? net->csr_img_rec_direct : net->csr_img_rec;
        if (stale) {
            hipLaunchKernelGGL(k_csr_image, dim3((c.c.g.n_slices * 64 + 255) / 256), dim3(256), 0, net->stream, c.c.g,
                               direct_image ? net->csr_plan_win_direct : net->csr_plan_win, hdr, rec);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
            stale = false;
        }
        c.img.hdr = hdr; c.img.rec = rec;
        if (part != CSR_STEP_INTERIOR) net->stat_steps_sparse_image += 1;      // (a step is one ALL launch, or a BORDER launch and perhaps an INTERIOR one)
    } else if (net->any_plasticity || net->any_modulation || net->any_conn_kind) {
        net->img_stale = net->img_stale_direct = true;          // this step's weight updates leave the records behind
    }
    hipEvent_t e1 = nullptr;
    if (waves || tail_blocks) {
        TRY(profile_open(net, &e1));
        if (e1 && part == CSR_STEP_BORDER && net->n_interior) net->ev_counts[net->ev_used - 1] = 0;   // the interior launch counts the pass
        const dim3 grid((waves + 3) / 4 + tail_blocks), block(256);
#define SNN_CSR_STEP(M)                                                                                              \
    do {                                                                                                             \
        if (image && c.pack.ptr) hipLaunchKernelGGL((k_step_csr_img<M, true>), grid, block, 0, net->stream, c);              \
        else if (image) hipLaunchKernelGGL((k_step_csr_img<M, false>), grid, block, 0, net->stream, c);                        \
        else if (net->peer_run && net->electrical && net->chemical) hipLaunchKernelGGL((k_step_csr<M, true, true, true>), grid, block, 0, net->stream, c); \
        else if (net->peer_run && net->electrical) hipLaunchKernelGGL((k_step_csr<M, true, false, true>), grid, block, 0, net->stream, c); \
        else if (net->peer_run) hipLaunchKernelGGL((k_step_csr<M, false, true, true>), grid, block, 0, net->stream, c);        \
        else if (net->electrical && net->chemical) hipLaunchKernelGGL((k_step_csr<M, true, true>), grid, block, 0, net->stream, c);  \
        else if (net->electrical) hipLaunchKernelGGL((k_step_csr<M, true, false>), grid, block, 0, net->stream, c);             \
        else hipLaunchKernelGGL((k_step_csr<M, false, true>), grid, block, 0, net->stream, c);                                  \
    } while (0)
#if !SNN_HAVE_CUSTOM_MODEL
        SNN_FOR_MODEL(SNN_CSR_STEP)
#else
        (void)grid, (void)block;
#endif
#undef SNN_CSR_STEP
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        if (e1) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    }
    if (part != CSR_STEP_BORDER) net->shadow_cur ^= 1;       // S(t+1) is complete once the last part is enqueued
    return SNN_OK;
}

// the interior half of a sparse shard handle's step (no-op unless step_begin left it pending)
int step_interior(snn_network *net)
{
    if (!net->interior_pending) return SNN_OK;
    net->interior_pending = false;
    return launch_step_csr(net, CSR_STEP_INTERIOR);
}

// first half of a step: inputs from S(t) and the local neurons' update
int step_begin(snn_network *net)
{
    if (net->drive_threshold && net->nn) {
        hipLaunchKernelGGL(k_synthetic_drive, dim3((net->nn + 255) / 256), dim3(256), 0, net->stream, net->xbuf, net->xl,
                           net->nn, net->drive_seed, net->clock, net->drive_threshold, net->drive_voltage);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    if (fused_step_applies(net)) { net->stat_steps_dense_one_launch += 1; return launch_step_resident(net); }
    if (dense_close_applies(net)) { net->stat_steps_dense_close += 1; return launch_dense_close(net); }
    if (csr_fast_step(net) && net->peer_run) {
        // peer form: nothing to overlap with -- border and interior slices in ONE launch, the border rows store into the peers
        net->stat_steps_sparse_one_launch += 1;
        TRY(launch_step_csr(net, CSR_STEP_ALL, /*pack=*/true));
        net->step_packed = true;
        net->interior_pending = false;
        return SNN_OK;
    }
    if (csr_fast_step(net)) {
        net->stat_steps_sparse_split += 1;
        // border slices first, writing the outgoing segments themselves; the interior slices follow once the caller has
        // started the exchange (step_interior: snn_run_sharded, snn_step_begin_local, or at the latest step_end)
        // (direct runs: the set this step packs into was cleared behind the previous step's rows, or at the run's start)
        if (!net->send_bits_clean && !net->direct_run) TRY(launch_step_close(net, /*cells=*/false, /*unpack=*/false));
        if (net->n_border) TRY(launch_step_csr(net, CSR_STEP_BORDER, /*pack=*/true));
        net->send_bits_clean = net->n_border == 0 && !net->direct_run;
        net->step_packed = true;
        net->interior_pending = true;
        return SNN_OK;
    }
    if (fused_csr_step_applies(net)) { net->stat_steps_sparse_one_launch += 1; return launch_step_csr(net); }
    net->stat_steps_two_kernel += 1;
    TRY(launch_inputs(net, net->local_inputs_done ? INPUTS_REMOTE : INPUTS_ALL));
    net->local_inputs_done = false;
    TRY(launch_update(net));
    return SNN_OK;
}

uint32_t plan_plane_mask(const snn_network *net);

// second half: remote last_firing_time, plasticity, histories, clock, spike trains (steps 3-6)
int step_end(snn_network *net)
{
    // of the neurons owned elsewhere only what this step's exchange carries stays current in the mirror
    if (net->sharded && net->n_shards > 1) net->mirror_mask = plan_plane_mask(net);
    TRY(step_interior(net));
    if (net->step_packed) {
        // the fast sparse step (csr_fast_step): unpack, spike trains and the clearing of the outgoing bitmaps in ONE launch
        net->step_packed = false;
        if (net->direct_run) {
            // nothing to unpack before the next rows: they read the received segments themselves
            if (!net->cells_stepped) TRY(launch_step_close(net, /*cells=*/true, /*unpack=*/false));
            net->stamp_pending = true;
            net->stat_direct_steps += 1;
            net->tail_done = false;
            net->hx_par ^= 1;
            if (net->peer_run) { net->p2p_epoch += 1; net->stat_peer_steps += 1; }
        } else {
            TRY(launch_step_close(net, /*cells=*/!net->cells_stepped, /*unpack=*/true));
        }
        net->cells_stepped = false;
        net->clock += 1;
        net->run_step_offset += 1;
        if (record_now(net)) net->hist_steps += 1;
        if (recording(net)) net->hist_tick += 1;
        return SNN_OK;
    }
    TRY(launch_exchange_unpack(net));      // shard handles: the other ranks' state of this step, their last_firing_time
    // weight snapshots: order 2 = before the step's weight updates (LatticeNetwork::iterate, neuron/mod.rs:2450-2461),
    // order 1 = after them (a lone Lattice, neuron/mod.rs:904-910)
    auto snapshots = [&](int order) -> int {
        if (!net->any_whist || !record_now(net)) return SNN_OK;
        for (const auto &l : net->lattices) {
            if (net->want_whist[l.slot] != order || l.count == 0) continue;
            float *dst = net->whist[l.slot] + (size_t)net->hist_steps * l.count * l.count;
            hipLaunchKernelGGL(k_weight_snapshot, dim3((l.count + 255) / 256, l.count), dim3(256), 0, net->stream,
                               net->W, net->ld, l.first, l.count, dst);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        }
        return SNN_OK;
    };
    TRY(snapshots(2));
    TRY(launch_plasticity(net));
    TRY(launch_reward_modulation(net));
    TRY(launch_reward_cross(net));
    TRY(snapshots(1));
    if ((net->want_avg || net->want_eeg) && record_now(net) && !net->lattices.empty()) {
        // after the exchange, so that a sharded handle reduces over every lattice's full population
        const size_t nl = net->lattices.size();
        SummaryArgs sa{};
        sa.xbuf = net->xbuf; sa.xl = net->xl; sa.first = net->lat_first_dev; sa.count = net->lat_count_dev;
        sa.avg_row = net->want_avg ? net->summ_avg + (size_t)net->hist_steps * nl : nullptr;
        sa.eeg_row = net->want_eeg ? net->summ_eeg + (size_t)net->hist_steps * nl : nullptr;
        sa.reference_voltage = net->eeg_ref; sa.distance = net->eeg_dist; sa.conductivity = net->eeg_cond;
        hipLaunchKernelGGL(k_lattice_summary, dim3((unsigned)nl), dim3(256), 0, net->stream, sa);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->clock += 1;
    if (!net->cells_stepped) TRY(launch_spike_trains(net, 1, net->run_step_offset, net->clock));
    net->cells_stepped = false;
    net->run_step_offset += 1;
    if (record_now(net)) net->hist_steps += 1;
    if (recording(net)) net->hist_tick += 1;
    return SNN_OK;
}

int grow_history(snn_network *net, uint64_t extra)
{
    if (!recording(net)) return SNN_OK;
    const uint64_t need = net->hist_steps + (extra + net->hist_every - 1) / net->hist_every + 1;
    if (need <= net->hist_cap && (!net->want_vhist || net->vhist) && (!net->want_raster || net->raster) &&
        (!net->want_avg || net->summ_avg) && (!net->want_eeg || net->summ_eeg)) {
        bool ok = true;
        for (const auto &l : net->lattices) ok = ok && (!net->want_whist[l.slot] || net->whist[l.slot] || l.count == 0);
        if (ok) return SNN_OK;
    }
    const uint64_t cap = std::max<uint64_t>(need, net->hist_cap + net->hist_cap / 2);   // geometric: O(T) copies overall
    auto regrow = [&](void **buf, size_t row_bytes, bool wanted) -> int {
        if (!wanted || row_bytes == 0) return SNN_OK;
        void *nb = nullptr;
        HIP_TRY(snn_malloc(&nb, std::max<size_t>(256, cap * row_bytes)), SNN_ERR_BUFFER_CREATE);
        if (*buf && net->hist_steps)
            HIP_TRY(hipMemcpyAsync(nb, *buf, net->hist_steps * row_bytes, hipMemcpyDeviceToDevice, net->stream),
                    SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
        if (*buf) (void)hipFree(*buf);
        *buf = nb;
        return SNN_OK;
    };
    TRY(regrow(reinterpret_cast<void **>(&net->vhist), (size_t)net->n_pad * 4, net->want_vhist));
    TRY(regrow(reinterpret_cast<void **>(&net->st_vhist), (size_t)net->c_pad * 4, net->want_vhist));
    TRY(regrow(reinterpret_cast<void **>(&net->raster), (size_t)(net->n_pad / 64) * 8, net->want_raster));
    TRY(regrow(reinterpret_cast<void **>(&net->summ_avg), net->lattices.size() * 4, net->want_avg));
    TRY(regrow(reinterpret_cast<void **>(&net->summ_eeg), net->lattices.size() * 4, net->want_eeg));
    for (const auto &l : net->lattices)
        TRY(regrow(reinterpret_cast<void **>(&net->whist[l.slot]), (size_t)l.count * l.count * 4, net->want_whist[l.slot] != 0));
    net->hist_cap = cap;
    net->stat_history_regrows += 1;
    return SNN_OK;
}

// Opens a run (snn_run, or a sequence of externally driven steps): static counts, history capacity, the
// spike-train lattices' clocks on the device and -- only when cell state or the clock changed behind the
// stepper's back -- the spike-train gap-junction values for the current clock.
int begin_run(snn_network *net, uint64_t iterations)
{
    TRY(ensure_counts(net));
    TRY(ensure_uniform_tables(net));
    TRY(ensure_exchange_plan(net));
    if ((net->want_avg || net->want_eeg) && net->sharded && net->n_shards > 1 &&
        (net->x_mode != SNN_EXCHANGE_ALLGATHER || !net->electrical))
        return fail(SNN_ERR_BAD_STATE, "per-lattice voltage reductions on a shard handle need every neuron's voltage: "
                                       "an all-gather exchange with electrical synapses on");
    TRY(grow_history(net, iterations));
    if (net->run_active) return SNN_OK;
    if (net->nc && net->csr && !net->cell_view[0]) {
        for (int i = 0; i < 2; ++i) TRY(dev_alloc_t(net, &net->cell_view[i], (size_t)net->c_pad));
        net->view_dirty = true;
    }
    if (net->nc) {
        // from a page-locked staging copy: the transfer may read its source any time until the stream has drained, and the
        // staging words are next written by the begin_run after this run's end_run (which waits for the stream)
        for (size_t i = 0; i < net->st_clock.size(); ++i) net->st_clock_pinned[i] = net->st_clock[i];
        HIP_TRY(hipMemcpyAsync(net->st_clock_dev, net->st_clock_pinned, net->st_clock.size() * sizeof(long long),
                               hipMemcpyHostToDevice, net->stream), SNN_ERR_BUFFER_WRITE);
        if (net->view_dirty) { net->stat_view_refreshes += 1; TRY(launch_spike_trains(net, 0, 0, net->clock)); }
    }
    net->view_dirty = false;
    TRY(check_reward_cross(net));
    net->run_step_offset = 0;
    net->run_active = true;
    return SNN_OK;
}

// Closes the open run: waits for the stream and folds the steps done into the host-side lattice clocks.
// keep_stdp: the end of a run call leaves a deferred STDP update pending -- it is self-contained (flags and delta vectors
// were evaluated when the step closed) and only the weights themselves depend on it, so the NEXT run call's first input
// pass applies it; every other entry point (getters, setters, graph access) flushes.
int end_run(snn_network *net, bool keep_stdp)
{
    TRY(flush_rstdp(net));
    if (!keep_stdp) TRY(flush_stdp(net));
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    if (net->run_active) {
        for (auto &c : net->st_clock) c += net->run_step_offset;
        net->run_step_offset = 0;
        net->run_active = false;
    }
    return SNN_OK;
}

int collect_profile(snn_network *net)
{
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    for (size_t i = 0; i < net->ev_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, net->ev_pool[i].first, net->ev_pool[i].second), SNN_ERR_WAIT);
        net->prof_ms += ms;
        net->prof_launches += (i < net->ev_counts.size()) ? net->ev_counts[i] : 1;
    }
    net->ev_used = 0;
    for (size_t i = 0; i < net->ev_used_pl; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, net->ev_pool_pl[i].first, net->ev_pool_pl[i].second), SNN_ERR_WAIT);
        net->prof_ms_pl += ms;
        net->prof_launches_pl += 1;
    }
    net->ev_used_pl = 0;
    return SNN_OK;
}

int graph_rows_io(snn_network *net, uint32_t pre_begin, uint32_t pre_count, float *weights, uint32_t *conns,
                  size_t host_ld, bool set)
{
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a CSR graph: use snn_set_graph_csr / snn_get_graph_csr");
    if ((uint64_t)pre_begin + pre_count > net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "row range exceeds n_tot");
    if (pre_count == 0 || net->nn == 0) return SNN_OK;
    if (!weights || !conns) return fail(SNN_ERR_BAD_ARG, "null graph pointer");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    // staged through a bounded device buffer: <= 64 MiB of host rows per hop
    // <= 64 MiB of host rows per hop and <= 32768 rows (grid.y of the import / export kernels)
    const uint32_t hop = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(pre_count, 32768),
                                                                        (64u << 20) / (host_ld * 4)));
    float *dw = nullptr;
    uint32_t *dc = nullptr;
    HIP_TRY(snn_malloc(&dw, (size_t)hop * host_ld * 4), SNN_ERR_BUFFER_CREATE);
    if (snn_malloc(&dc, (size_t)hop * host_ld * 4) != hipSuccess) {
        (void)hipFree(dw);
        return fail(SNN_ERR_BUFFER_CREATE, "staging allocation failed");
    }
    int rc = SNN_OK;
    uint32_t *bad = nullptr;
    if (set && (snn_malloc(&bad, 256) != hipSuccess || hipMemsetAsync(bad, 0, 256, net->stream) != hipSuccess)) {
        (void)hipFree(dw); (void)hipFree(dc);
        if (bad) (void)hipFree(bad);
        return fail(SNN_ERR_BUFFER_CREATE, "staging allocation failed");
    }
    for (uint32_t r = 0; r < pre_count && rc == SNN_OK; r += hop) {
        const uint32_t rows = std::min(hop, pre_count - r);
        const size_t bytes = (size_t)rows * host_ld * 4;
        if (set) {
            if (hipMemcpyAsync(dw, weights + (size_t)r * host_ld, bytes, hipMemcpyHostToDevice, net->stream) != hipSuccess ||
                hipMemcpyAsync(dc, conns + (size_t)r * host_ld, bytes, hipMemcpyHostToDevice, net->stream) != hipSuccess) {
                rc = fail(SNN_ERR_BUFFER_WRITE, "graph upload failed");
                break;
            }
            hipLaunchKernelGGL(k_graph_import, dim3((net->ld + 255) / 256, rows), dim3(256), 0, net->stream,
                               net->W, net->ld, net->n_loc, net->q0, pre_begin + r, rows, dw, dc, host_ld, bad);
        } else {
            // columns outside the shard are left untouched in the caller's buffers
            if (hipMemcpyAsync(dw, weights + (size_t)r * host_ld, bytes, hipMemcpyHostToDevice, net->stream) != hipSuccess ||
                hipMemcpyAsync(dc, conns + (size_t)r * host_ld, bytes, hipMemcpyHostToDevice, net->stream) != hipSuccess) {
                rc = fail(SNN_ERR_BUFFER_WRITE, "graph staging failed");
                break;
            }
            if (net->n_loc)
                hipLaunchKernelGGL(k_graph_export, dim3((net->n_loc + 255) / 256, rows), dim3(256), 0, net->stream,
                                   net->W, net->ld, net->n_loc, net->q0, pre_begin + r, rows, dw, dc, host_ld);
            if (hipMemcpyAsync(weights + (size_t)r * host_ld, dw, bytes, hipMemcpyDeviceToHost, net->stream) != hipSuccess ||
                hipMemcpyAsync(conns + (size_t)r * host_ld, dc, bytes, hipMemcpyDeviceToHost, net->stream) != hipSuccess) {
                rc = fail(SNN_ERR_BUFFER_READ, "graph download failed");
                break;
            }
        }
        if (hipGetLastError() != hipSuccess) { rc = fail(SNN_ERR_QUEUE, "graph kernel launch failed"); break; }
        if (hipStreamSynchronize(net->stream) != hipSuccess) { rc = fail(SNN_ERR_WAIT, "graph transfer wait failed"); break; }
    }
    uint32_t bad_host[3] = {0, 0, 0};
    if (set && rc == SNN_OK && copy_sync(net, bad_host, bad, 12, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SNN_ERR_BUFFER_READ, "graph check download failed");
    (void)hipFree(dw);
    (void)hipFree(dc);
    if (bad) (void)hipFree(bad);
    if (set) net->counts_dirty = true;
    if (rc == SNN_OK && bad_host[0])
        return fail(SNN_ERR_BAD_ARG, std::to_string(bad_host[0]) + " connected edge(s) carry a NaN weight, e.g. (pre " + std::to_string(bad_host[1]) +
                    ", post " + std::to_string(bad_host[2]) + "): NaN is the absent-edge sentinel of the device matrix, such an edge cannot be "
                    "stored (graph/mod.rs:204-213); the rows of this call were imported with those edges absent");
    return rc;
}

} // namespace

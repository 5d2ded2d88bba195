// k_update<MODEL> -- one thread per local neuron: fixed-order combine of the chunk partials,
// averaging, receptor kinetics + currents, the model's Euler step, the neuron's own
// neurotransmitter release, spike test / reset, last_firing_time stamp, spike-raster ballot and the
// optional voltage-history store, all in one launch.  It supersedes the reference's per-model
// iterate kernels plus set_last_firing_time and add_grid_voltage_history
// (neuron/gpu_lattices/mod.rs:141-155, 282-296; integrate_and_fire/mod.rs:379-416, 633-694) and
// provides what the reference GPU path lacks: Izhikevich, LIF and Hodgkin-Huxley steps of the
// backend structs (integrate_and_fire/mod.rs:173-255, 1222-1267; hodgkin_huxley/mod.rs:156-241;
// ion_channels/mod.rs:40-44, 219-316).
//
// Step order per neuron = SURVEY §8(g) step 2; every expression keeps the reference's f32
// operation order (no FMA contraction in this TU).
#pragma once
#include "snn_custom_model.hpp"
#include "snn_layout.hpp"
#include "snn_math.hpp"

namespace snn {

struct UpdateArgs {
    NeuronArrays n;
    const float *part_i;       // [n_chunks][ld]
    const float *part_t;       // [3][n_chunks][ld]
    const uint32_t *n_in;      // [ld]
    const uint32_t *tcount;    // [3][ld]
    uint32_t ld, n_chunks, q0, n_loc;
    RowMap rows;               // local row -> global neuron (rows.q0 = q0 unless the shard owns a set of ranges)
    long long clock;
    int electrical, chemical, nt_kind, rc_kind;
    float *vhist_row;          // this step's row of the voltage history (global neuron index) or null
    unsigned long long *spike_row;   // this step's row of the bit-packed raster or null
    uint32_t *spike_counts;          // per-neuron spike totals (SpikeHistory::aggregate) or null
    // Where the exchanged state (V, spike flag, t) of the step is stored: `xout` = the handle's exchange buffer
    // (= n.xbuf on the two-kernel path, which updates in place); the fused small-lattice step reads S(t) from a
    // shadow copy (n.xbuf) while it writes S(t+1) to the exchange buffer and to the other shadow (`xout2`).
    float *xout, *xout2;
    int has_nt;                 // some neuron of the handle releases a neurotransmitter (else the flag planes are not read)
    uint32_t live_mask;         // bit k: some neuron or cell of the handle releases transmitter type k (all ones: unknown).  A
                                // type nobody releases has count 0 at every receptor: its kinetics inputs are never loaded
    int bcm;                    // BCMIzhikevichNeuron: keep the activity bookkeeping (the step itself is Izhikevich's)
    int model_is_custom;        // the generated neuron model: it uses the library's generated receptor set, if any
    // Dense shard handles (all-gather exchange): k_update writes the handle's own slot of the wire buffer itself -- per plane
    // wire_count words, then the spike bits, the wire format of snn_kernels_exchange.hpp -- so the step needs no pack launch.
    // wire_out = the slot (null: no packing); neuron ql of the shard is entry ql of the slot.
    uint32_t *wire_out;
    uint32_t wire_count, wire_planes, wire_plane_id[1 + K_TYPES];
};

// Second level of the canonical sum: chunk partials added in ascending chunk order from 0.0f.  The loads of
// a batch are issued together (the adds stay strictly sequential), so a thread keeps 16 reads in flight.  (Batches of 32
// with the next batch requested early made C3's update 1.3 us faster and, at 104 registers instead of 63, the update of
// the 1 M neurons of configs[4] on a shard handle 3.5 us slower: not kept.)
__device__ __forceinline__ float combine_partials(const float *p, uint32_t n_chunks, size_t ld)
{
    constexpr uint32_t B = 16;
    float s = 0.0f;
    uint32_t c = 0;
    for (; c + B <= n_chunks; c += B) {
        float v[B];
#pragma unroll
        for (uint32_t u = 0; u < B; ++u) v[u] = p[(size_t)(c + u) * ld];
#pragma unroll
        for (uint32_t u = 0; u < B; ++u) s += v[u];
    }
    for (; c < n_chunks; ++c) s += p[(size_t)c * ld];
    return s;
}

// NeurotransmitterKinetics::apply_t_change: Approximate iterate_and_spike/mod.rs:193-196,
// Destexhe :148-150
// t_capture (or null): [K_TYPES], receives the new concentration of every type the neuron releases (the one-launch run
// publishes it without reading it back)
__device__ __forceinline__ void neuron_nt_update(const UpdateArgs &a, uint32_t q, float voltage,
                                                 uint32_t spiking_prev, float dt, float *t_capture = nullptr)
{
    if (!a.has_nt) return;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        if (!a.n.nt_flags[i]) continue;
        const size_t at = a.n.xl.at(q, PLANE_T0 + k);
        float t;
        if (SNN_HAVE_CUSTOM_NT && a.nt_kind == CUSTOM_KINETICS) {
            // generated apply_t_change (nb_macro lib.rs:6489-6498); a neuron releases on its previous spike flag
            float x[custom_nt::NSTORE];
#pragma unroll
            for (int j = 0; j < custom_nt::NVARS; ++j) x[j] = a.n.nt_custom[j][i];
            t = a.n.xbuf[at];
            custom_nt::apply(t, x, voltage, spiking_prev != 0, dt);
#pragma unroll
            for (int j = 0; j < custom_nt::NVARS; ++j) a.n.nt_custom[j][i] = x[j];
        } else {
            t = nt_apply(a.nt_kind, a.n.xbuf[at], a.n.nt_t_max[i], a.n.nt_clearance[i], a.n.nt_v_p[i],
                         a.n.nt_k_p[i], voltage, spiking_prev, dt);
        }
        a.xout[at] = t;
        if (a.xout2) a.xout2[at] = t;
        if (t_capture) t_capture[k] = t;
    }
}

// Where a column's second-level sums come from: the chunk partials in global memory (two-kernel path) ...
struct GlobalSums {
    const UpdateArgs &a;
    uint32_t ql;
    __device__ __forceinline__ float elec() const { return combine_partials(a.part_i + ql, a.n_chunks, a.ld); }
    __device__ __forceinline__ float chem(int k) const
    {
        return combine_partials(a.part_t + (size_t)k * a.n_chunks * a.ld + ql, a.n_chunks, a.ld);
    }
};

// Ionotropic::update_receptor_kinetics (iterate_and_spike/mod.rs:1186-1205); a type absent from the aggregated input
// (count 0) leaves r untouched
template <class Sums>
__device__ __forceinline__ void receptors_kinetics(const UpdateArgs &a, uint32_t q, uint32_t ql, float dt, const Sums &sums)
{
    if (SNN_HAVE_CUSTOM_RECEPTORS && custom_receptors::MULTI_STATE && a.model_is_custom) {
        // a generated receptor set with several states per type (nb_macro lib.rs:7306-7316, 7391-7404): the states and
        // their kinetics variables are the set's own variables; a type absent from the input leaves its states untouched
        float x[custom_receptors::NSTORE];
#pragma unroll
        for (int j = 0; j < custom_receptors::NVARS; ++j) x[j] = a.n.rx_custom[j][q];
#pragma unroll
        for (int k = 0; k < custom_receptors::NTYPES; ++k) {
            if (!a.n.rc_flags[(size_t)k * a.n.n_pad + q]) continue;
            const uint32_t cnt = a.tcount[(size_t)k * a.ld + ql];
            if (cnt != 0) custom_receptors::update_kinetics(k, sums.chem(k) / (float)cnt, dt, x);
        }
#pragma unroll
        for (int j = 0; j < custom_receptors::NVARS; ++j) a.n.rx_custom[j][q] = x[j];
        return;
    }
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        if (!a.n.rc_flags[i]) continue;
        const uint32_t cnt = a.tcount[(size_t)k * a.ld + ql];
        if (cnt != 0) {
            // second level of the canonical sum, then the per-type average
            const float s = sums.chem(k);
            const float t = s / (float)cnt;
            if (SNN_HAVE_CUSTOM_RC && a.rc_kind == CUSTOM_KINETICS) {
                // generated apply_r_change (nb_macro lib.rs:6778-6786)
                float x[custom_rc::NSTORE];
#pragma unroll
                for (int j = 0; j < custom_rc::NVARS; ++j) x[j] = a.n.rc_custom[j][i];
                float r = a.n.rc_r[i];
                custom_rc::apply(r, x, t, dt);
#pragma unroll
                for (int j = 0; j < custom_rc::NVARS; ++j) a.n.rc_custom[j][i] = x[j];
                a.n.rc_r[i] = r;
            } else {
                a.n.rc_r[i] = rc_apply(a.rc_kind, a.n.rc_r[i], t, a.n.rc_alpha[i], a.n.rc_beta[i], dt);
            }
        }
    }
}

// Ionotropic::set_receptor_currents (iterate_and_spike/mod.rs:1260-1284; currents :1103-1105, 1132-1137, 1164-1166)
__device__ __forceinline__ void receptors_set_currents(const UpdateArgs &a, uint32_t q, float v_old)
{
    if (SNN_HAVE_CUSTOM_RECEPTORS && a.model_is_custom) {
        // generated receptor set (nb_macro lib.rs:7512-7543): the on_iteration of every receptor present, in
        // declaration order, over the set's variables
        float x[custom_receptors::NSTORE];
#pragma unroll
        for (int j = 0; j < custom_receptors::NVARS; ++j) x[j] = a.n.rx_custom[j][q];
#pragma unroll
        for (int k = 0; k < custom_receptors::NTYPES; ++k) {
            const size_t i = (size_t)k * a.n.n_pad + q;
            if (a.n.rc_flags[i]) custom_receptors::iterate(k, v_old, a.n.rc_r[i], x);
        }
#pragma unroll
        for (int j = 0; j < custom_receptors::NVARS; ++j) a.n.rx_custom[j][q] = x[j];
        return;
    }
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        if (!a.n.rc_flags[i]) continue;
        const float r = a.n.rc_r[i];
        if (k == 1) {
            a.n.rc_current[i] = ((1.0f / (1.0f + ((expf_glibc(-0.062f * v_old) * a.n.rc_mg[i]) / 3.75f))
                                  * a.n.rc_g[i]) * r) * (v_old - a.n.rc_e[i]);
        } else {
            a.n.rc_current[i] = (a.n.rc_g[i] * r) * (v_old - a.n.rc_e[i]);
        }
    }
}

template <class Sums>
__device__ __forceinline__ void receptors_update(const UpdateArgs &a, uint32_t q, uint32_t ql,
                                                 float v_old, float dt, const Sums &sums)
{
    receptors_kinetics(a, q, ql, dt, sums);
    receptors_set_currents(a, q, v_old);
}

// Ionotropic::get_receptor_currents (iterate_and_spike/mod.rs:1286-1304)
__device__ __forceinline__ float receptor_currents(const UpdateArgs &a, uint32_t q, float dt, float c_m)
{
    float total = 0.0f;
    if (SNN_HAVE_CUSTOM_RECEPTORS && a.model_is_custom) {      // lib.rs:7546-7566: the `current`s of the receptors present
#pragma unroll
        for (int k = 0; k < custom_receptors::NTYPES; ++k) {
            if (custom_receptors::CURRENT_INDEX[k] < 0) continue;
            if (a.n.rc_flags[(size_t)k * a.n.n_pad + q]) total += a.n.rx_custom[custom_receptors::CURRENT_INDEX[k]][q];
        }
        return total * (dt / c_m);
    }
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        if (a.n.rc_flags[i]) total += a.n.rc_current[i];
    }
    return total * (dt / c_m);
}

// ---- the built-in chemical step with every load ahead of every store ------------------------------------------------
// A neuron's update is one wavefront walking a dozen small arrays; as the reference writes it (kinetics, then currents,
// then the sum of the currents, then the transmitter release, each type behind a flag test) every flagged type costs a chain
// of dependent memory round trips -- flag -> count -> r -> store r -> load r -> store current -> load current -- because a
// store may alias the next load.  Measured on MI355X: 1.9 us per live type in the one-launch run of a 32 x 32 lattice, and
// the better part of k_update<2>'s 12 us at BASELINE configs[2].  Same arithmetic, restated as: load everything the step
// can need (independent loads, one round trip), compute in registers, store at the end of update_neuron_at.
struct ChemStep {
    bool rc_on = false, nt_on = false;     // the fused forms are in use (built-in kinetics)
    float total = 0.0f;                    // (I_AMPA + I_NMDA + I_GABA) of the receptors present, in that order
    float r_new[K_TYPES], cur_new[K_TYPES];
    uint32_t store_r = 0, store_cur = 0;   // bit k: write r_new[k] / cur_new[k] back
    uint32_t nt_flags = 0;                 // bit k: the neuron releases type k
    float nt_t[K_TYPES], nt_t_max[K_TYPES], nt_c[K_TYPES], nt_v_p[K_TYPES], nt_k_p[K_TYPES], nt_new[K_TYPES];
};

// Ionotropic::update_receptor_kinetics + set_receptor_currents + the sum of get_receptor_currents
// (iterate_and_spike/mod.rs:1186-1205, 1260-1304), built-in receptor kinetics
// what the receptor step reads of a neuron: requested at the very start of the neuron's update (round 6), together with everything
// else the update reads -- every load of the update in ONE round trip; the arithmetic follows when the sums are there
struct RcLoaded {
    uint32_t fl[K_TYPES], cnt[K_TYPES];
    float r[K_TYPES], al[K_TYPES], be[K_TYPES], g[K_TYPES], e[K_TYPES], mg;
};
__device__ __forceinline__ void chem_receptors_load(const UpdateArgs &a, uint32_t q, uint32_t ql, RcLoaded &l)
{
    l.mg = a.n.rc_mg[(size_t)1 * a.n.n_pad + q];
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        const bool live = (a.live_mask >> k & 1u) != 0u;                 // launch-uniform
        l.fl[k] = a.n.rc_flags[i];
        l.r[k] = a.n.rc_r[i]; l.g[k] = a.n.rc_g[i]; l.e[k] = a.n.rc_e[i];
        l.cnt[k] = 0u; l.al[k] = 0.0f; l.be[k] = 0.0f;
        if (live) {
            l.cnt[k] = a.tcount[(size_t)k * a.ld + ql];
            l.al[k] = a.n.rc_alpha[i]; l.be[k] = a.n.rc_beta[i];
        }
    }
}

// ... and the arithmetic, once the sums are there
template <class Sums>
__device__ __forceinline__ void chem_receptors(const UpdateArgs &a, const RcLoaded &l, float v_old, float dt, const Sums &sums, ChemStep &c)
{
    float s[K_TYPES];
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) s[k] = (a.live_mask >> k & 1u) ? sums.chem(k) : 0.0f;      // second level of the canonical sum
    c.rc_on = true;
    c.total = 0.0f;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const bool on = l.fl[k] != 0u, fed = on && l.cnt[k] != 0u;       // a type absent from the input leaves r untouched
        const float t = s[k] / (float)(fed ? l.cnt[k] : 1u);             // the per-type average
        const float rk = fed ? rc_apply(a.rc_kind, l.r[k], t, l.al[k], l.be[k], dt) : l.r[k];
        const float cur = k == 1 ? ((1.0f / (1.0f + ((expf_glibc(-0.062f * v_old) * l.mg) / 3.75f)) * l.g[k]) * rk) * (v_old - l.e[k])
                                 : (l.g[k] * rk) * (v_old - l.e[k]);
        c.r_new[k] = rk; c.cur_new[k] = cur;
        c.store_r |= fed ? (1u << k) : 0u;
        c.store_cur |= on ? (1u << k) : 0u;
        if (on) c.total += cur;
    }
}

// the loads of NeurotransmitterKinetics::apply_t_change for the types the neuron releases (built-in kinetics)
__device__ __forceinline__ void chem_nt_load(const UpdateArgs &a, uint32_t q, ChemStep &c)
{
    c.nt_on = true;
    if (!a.has_nt) return;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        c.nt_t[k] = 0.0f; c.nt_t_max[k] = 0.0f; c.nt_c[k] = 0.0f; c.nt_v_p[k] = 0.0f; c.nt_k_p[k] = 1.0f;
        if (!(a.live_mask >> k & 1u)) continue;                          // launch-uniform: nobody releases this type
        const size_t i = (size_t)k * a.n.n_pad + q;
        c.nt_flags |= a.n.nt_flags[i] ? (1u << k) : 0u;
        c.nt_t[k] = a.n.xbuf[a.n.xl.at(q, PLANE_T0 + k)];
        c.nt_t_max[k] = a.n.nt_t_max[i]; c.nt_c[k] = a.n.nt_clearance[i]; c.nt_v_p[k] = a.n.nt_v_p[i]; c.nt_k_p[k] = a.n.nt_k_p[i];
    }
}

// ... its arithmetic (at the place the reference calls apply_t_changes: after the voltage update, before the spike test) ...
__device__ __forceinline__ void chem_nt_apply(const UpdateArgs &a, ChemStep &c, float voltage, uint32_t spiking_prev, float dt)
{
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k)
        c.nt_new[k] = (c.nt_flags >> k & 1u) ? nt_apply(a.nt_kind, c.nt_t[k], c.nt_t_max[k], c.nt_c[k], c.nt_v_p[k], c.nt_k_p[k], voltage, spiking_prev, dt)
                                             : 0.0f;
}

// ... and every store of the chemical step, at the end of the neuron's update
__device__ __forceinline__ void chem_store(const UpdateArgs &a, uint32_t q, const ChemStep &c, float *t_capture)
{
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        const size_t i = (size_t)k * a.n.n_pad + q;
        if (c.rc_on && (c.store_r >> k & 1u)) a.n.rc_r[i] = c.r_new[k];
        if (c.rc_on && (c.store_cur >> k & 1u)) a.n.rc_current[i] = c.cur_new[k];
        if (c.nt_on && (c.nt_flags >> k & 1u)) {
            const size_t at = a.n.xl.at(q, PLANE_T0 + k);
            a.xout[at] = c.nt_new[k];
            if (a.xout2) a.xout2[at] = c.nt_new[k];
            if (t_capture) t_capture[k] = c.nt_new[k];
        }
    }
}

// What a generated on_electrochemical_iteration (nb_macro lib.rs:2280-2316) may call on its neuron
template <class Sums>
struct ChemicalStep {
    const UpdateArgs &a;
    uint32_t q, ql, spiking_prev;
    float dt;
    const Sums &sums;
    float *t_capture;
    __device__ __forceinline__ void update_receptor_kinetics() { receptors_kinetics(a, q, ql, dt, sums); }
    __device__ __forceinline__ void set_receptor_currents(float voltage) { receptors_set_currents(a, q, voltage); }
    __device__ __forceinline__ float get_receptor_currents(float step, float c_m) { return receptor_currents(a, q, step, c_m); }
    __device__ __forceinline__ void apply_t_changes(float voltage) { neuron_nt_update(a, q, voltage, spiking_prev, dt, t_capture); }
};

__device__ __forceinline__ float gate_update(float state, float alpha, float beta, float dt)
{
    const float alpha_state = alpha * (1.0f - state);
    const float beta_state = beta * state;
    return state + dt * (alpha_state - beta_state);
}

// One neuron's step (local column ql) at network clock `clock`, its voltage also recorded in `vhist_row` (or null);
// returns its spike flag (and, where asked for, the voltage it stored).
// HOIST: Hodgkin-Huxley's exponentials and powers through the branch-free main paths (more registers live at once: the one-launch
// run, whose registers are the weights', keeps the plain calls)
// CHEM_OK = false: the caller is a kernel instantiated for gap junctions only -- the receptor step is compiled out (its preloaded
// operands would otherwise cost the sparse electrical step its eighth wavefront per SIMD)
template <int MODEL, class Sums, bool HOIST = true, bool CHEM_OK = true>
__device__ __forceinline__ uint32_t update_neuron_at(const UpdateArgs &a, uint32_t ql, const Sums &sums, long long clock,
                                                     float *vhist_row, float *v_stored = nullptr, float *t_capture = nullptr)
{
    const int a_chemical = CHEM_OK ? a.chemical : 0;
    uint32_t spike = 0;
    {
        const uint32_t q = a.rows.global_of(ql);
        const size_t v_at = a.n.xl.at(q, PLANE_V), s_at = a.n.xl.at(q, PLANE_SPIKE);
        const float v = a.n.xbuf[v_at];
        const float dt = uload(a.n.uni, NP_DT, a.n.dt, q);
        const float c_m = uload(a.n.uni, NP_C_M, a.n.c_m, q);
        const uint32_t spiking_prev = reinterpret_cast<const uint32_t *>(a.n.xbuf)[s_at];
        // ---- every load of the update, up front (round 6).  The receptor step, the transmitter release and the model's own state
        // used to be loaded where they were used -- behind the sums, behind each other's arithmetic and its branches: three to
        // five dependent round trips per neuron.  Nothing below stores before it has been read here. ----
        constexpr bool own_chemical_step = MODEL == CUSTOM_MODEL && custom::HAS_ELECTROCHEMICAL;
        // built-in kinetics (every library without generated code): the chemical step loads first and stores last (ChemStep)
        const bool fused_rc = MODEL != CUSTOM_MODEL && !(SNN_HAVE_CUSTOM_RECEPTORS && a.model_is_custom) &&
                              !(SNN_HAVE_CUSTOM_RC && a.rc_kind == CUSTOM_KINETICS);
        const bool fused_nt = MODEL != CUSTOM_MODEL && !(SNN_HAVE_CUSTOM_NT && a.nt_kind == CUSTOM_KINETICS);
        ChemStep cs;
        RcLoaded rcl;
        const bool rc_fused_now = a_chemical && !own_chemical_step && fused_rc;
        if (rc_fused_now) chem_receptors_load(a, q, ql, rcl);
        if (fused_nt) chem_nt_load(a, q, cs);
        const uint32_t cnt_in = a.electrical ? a.n_in[ql] : 0u;
        float izh_w = 0.0f;
        if (MODEL == 0) izh_w = a.n.w_value[q];
        float hh_m = 0.0f, hh_h = 0.0f, hh_n = 0.0f, hh_g_na = 0.0f, hh_e_na = 0.0f, hh_g_k = 0.0f, hh_e_k = 0.0f, hh_g_kl = 0.0f, hh_e_kl = 0.0f,
              hh_v_th = 0.0f;
        uint32_t hh_was_increasing = 0u;
        if (MODEL == 2) {
            hh_m = a.n.m_state[q]; hh_h = a.n.h_state[q]; hh_n = a.n.n_state[q];
            hh_g_na = a.n.g_na[q]; hh_e_na = a.n.e_na[q]; hh_g_k = a.n.g_k[q]; hh_e_k = a.n.e_k[q];
            hh_g_kl = a.n.g_k_leak[q]; hh_e_kl = a.n.e_k_leak[q];
            hh_v_th = a.n.v_th[q]; hh_was_increasing = a.n.was_increasing[q];
        }
        if (MODEL == 0 && a.bcm) {
            // BCMIzhikevichNeuron::iterate_and_spike, integrate_and_fire/mod.rs:1458-1469 (electrical) / :1484-1495
            float cur = a.n.bcm_cur[q], avg = a.n.bcm_avg[q], clock = a.n.bcm_clock[q];
            const uint32_t num = a.n.bcm_num_spikes[q] + (spiking_prev ? 1u : 0u);
            bcm_window_update(clock, a.n.bcm_window[q], dt, num, a.n.bcm_period[q], cur, avg, !a_chemical);
            a.n.bcm_cur[q] = cur; a.n.bcm_avg[q] = avg; a.n.bcm_clock[q] = clock; a.n.bcm_num_spikes[q] = num;
        }

        // input current: chunk partials in ascending order, then the averager (neuron/mod.rs:722-729)
        float i_in = 0.0f;
        if (a.electrical) {
            const float s = sums.elec();
            i_in = s / (cnt_in == 0 ? 1.0f : (float)cnt_in);
        }

        if (a_chemical && !own_chemical_step) {
            if (fused_rc) chem_receptors(a, rcl, v, dt, sums, cs);
            else receptors_update(a, q, ql, v, dt, sums);
        }
        // (the reference's get_receptor_currents / apply_t_changes at their places in the model's step)
        auto receptor_currents = [&](const UpdateArgs &aa, uint32_t qq, float step, float cm) {
            return cs.rc_on ? cs.total * (step / cm) : snn::receptor_currents(aa, qq, step, cm);
        };
        auto neuron_nt_update = [&](const UpdateArgs &aa, uint32_t qq, float voltage, uint32_t prev, float step, float *capture) {
            if (cs.nt_on) chem_nt_apply(aa, cs, voltage, prev, step);
            else snn::neuron_nt_update(aa, qq, voltage, prev, step, capture);
        };

        float v_new;
        if (MODEL == 0) {            // Izhikevich
            const float w = izh_w;
            const float dv = (0.04f * (v * v) + 5.0f * v + 140.0f - w + i_in) * (dt / c_m);
            const float dw = (uload(a.n.uni, NP_A, a.n.a, q) * (uload(a.n.uni, NP_B, a.n.b, q) * v - w)) *
                             (dt / uload(a.n.uni, NP_TAU_M, a.n.tau_m, q));
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            float w_new = w + dw;
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            if (v_new >= uload(a.n.uni, NP_V_TH, a.n.v_th, q)) {
                spike = 1;
                v_new = uload(a.n.uni, NP_C, a.n.c, q);
                w_new += uload(a.n.uni, NP_D, a.n.d, q);
            }
            a.n.w_value[q] = w_new;
        } else if (MODEL == 1) {     // leaky integrate-and-fire
            const float dv = ((a.n.leak_constant[q] * (v - a.n.e_l[q])) +
                              (a.n.integration_constant[q] * (i_in / a.n.g_l[q]))) * (dt / a.n.tau_m[q]);
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            float rc = a.n.refractory_count[q];
            if (rc > 0.0f) {
                v_new = a.n.v_reset[q];
                rc -= 1.0f;
            } else if (v_new >= a.n.v_th[q]) {
                spike = 1;
                v_new = a.n.v_reset[q];
                rc = a.n.tref[q] / dt;
            }
            a.n.refractory_count[q] = rc;
        } else if (MODEL == 3) {     // quadratic integrate-and-fire (integrate_and_fire/mod.rs:324-365)
            const float dv = ((a.n.qif_alpha[q] * (v - a.n.v_reset[q]) * (v - a.n.qif_v_c[q])) +
                              a.n.integration_constant[q] * i_in) * (dt / a.n.tau_m[q]);
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            float rc = a.n.refractory_count[q];
            if (rc > 0.0f) {
                v_new = a.n.v_reset[q];
                rc -= 1.0f;
            } else if (v_new >= a.n.v_th[q]) {
                spike = 1;
                v_new = a.n.v_reset[q];
                rc = a.n.tref[q] / dt;
            }
            a.n.refractory_count[q] = rc;
        } else if (MODEL == 4) {     // simple leaky integrate-and-fire (integrate_and_fire/mod.rs:1577-1630)
            const float dv = (a.n.slif_g[q] * (v - a.n.slif_e[q]) + i_in) * dt;
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            if (v_new >= a.n.v_th[q]) {
                spike = 1;
                v_new = a.n.v_reset[q];
            }
        } else if (MODEL == 5 || MODEL == 6) {
            // adaptive (5) and adaptive exponential (6) leaky integrate-and-fire, integrate_and_fire/mod.rs:1001-1049,
            // 1132-1155; adaptive_get_dw_change / adaptive_handle_spiking :1001-1031
            const float w = a.n.w_value[q], e_l = a.n.e_l[q], g_l = a.n.g_l[q];
            float acc = a.n.leak_constant[q] * (v - e_l);
            if (MODEL == 6) {
                const float sf = a.n.slope_factor[q];
                acc = acc + (sf * expf_glibc((v - a.n.v_th[q]) / sf));
            }
            const float dv = (acc + (a.n.integration_constant[q] * (i_in / g_l)) - (w / g_l)) * (dt / c_m);
            const float dw = (a.n.adp_alpha[q] * (v - e_l) - w) * (dt / a.n.tau_m[q]);
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            float w_new = w + dw;
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            float rc = a.n.refractory_count[q];
            if (rc > 0.0f) {
                v_new = a.n.v_reset[q];
                rc -= 1.0f;
            } else if (v_new >= a.n.v_th[q]) {
                spike = 1;
                v_new = a.n.v_reset[q];
                w_new += a.n.adp_beta[q];
                rc = a.n.tref[q] / dt;
            }
            a.n.refractory_count[q] = rc;
            a.n.w_value[q] = w_new;
        } else if (MODEL == 7) {     // leaky Izhikevich (integrate_and_fire/mod.rs:1336-1356), powf(2.0) taken as v * v
            const float w = a.n.w_value[q];
            const float dv = (0.04f * (v * v) + 5.0f * v + 140.0f - w * (v - a.n.e_l[q]) + i_in) * (dt / c_m);
            const float dw = (a.n.a[q] * (a.n.b[q] * v - w)) * (dt / a.n.tau_m[q]);
            if (a_chemical) {
                const float neurotransmitter_dv = -receptor_currents(a, q, dt, c_m);
                v_new = v + (dv + neurotransmitter_dv);
            } else {
                v_new = v + dv;
            }
            float w_new = w + dw;
            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);
            if (v_new >= a.n.v_th[q]) {
                spike = 1;
                v_new = a.n.c[q];
                w_new += a.n.d[q];
            }
            a.n.w_value[q] = w_new;
        } else if (MODEL == CUSTOM_MODEL) {
            // generated model, nb_macro semantics (build_test/nb_macro/src/lib.rs:2259-2345; hand expansion
            // tests/lif_reference.rs): on_iteration; with neurotransmission `v -= receptor currents` (the receptor kinetics
            // and currents were taken at the old voltage above) and the transmitter release; spike detection, on_spike
            float x[custom::NSTORE];
#pragma unroll
            for (int k = 0; k < custom::NVARS; ++k) x[k] = a.n.custom[k][q];
            const float g_gap = a.n.gap_conductance[q];
            float vc = v;
            if (own_chemical_step && a_chemical) {
                ChemicalStep<Sums> chem{a, q, ql, spiking_prev, dt, sums, t_capture};
                custom::on_electrochemical_iteration(vc, x, i_in, dt, c_m, g_gap, chem);
            } else {
                custom::on_iteration(vc, x, i_in, dt, c_m, g_gap);
                if (a_chemical) {          // the electrical form is on_iteration + spike handling alone (lib.rs:2266-2272)
                    vc -= receptor_currents(a, q, dt, c_m);
                    neuron_nt_update(a, q, vc, spiking_prev, dt, t_capture);
                }
            }
            spike = custom::spike_detection(vc, x, i_in, dt, c_m, g_gap) ? 1u : 0u;
            if (spike) custom::on_spike(vc, x, i_in, dt, c_m, g_gap);
            v_new = vc;
#pragma unroll
            for (int k = 0; k < custom::NVARS; ++k) a.n.custom[k][q] = x[k];
        } else {                     // Hodgkin-Huxley
            // The six exponentials of the rates and the two powers of the gates through the MAIN paths of expf / powf (snn_math.hpp):
            // branch-free, so that their table loads are in flight together -- one round trip for the exponentials, two for the
            // powers, where the full functions made eleven, one after the other.  An argument outside a main path (a gate at
            // exactly 0 -- the first step -- or a voltage on its way to infinity) sets `special`: the full functions then.
            const float m_state = hh_m, h_state = hh_h, n_state = hh_n;
            float e_ma = 0.0f, e_mb = 0.0f, e_ha = 0.0f, e_hb = 0.0f, e_na_ = 0.0f, e_nb = 0.0f;
            bool special = true;
            if constexpr (HOIST) {
                special = false;
                e_ma = expf_glibc_main(-(v + 40.0f) / 10.0f, special); e_mb = expf_glibc_main(-(v + 65.0f) / 18.0f, special);
                e_ha = expf_glibc_main(-(v + 65.0f) / 20.0f, special); e_hb = expf_glibc_main(-(v + 35.0f) / 10.0f, special);
                e_na_ = expf_glibc_main(-(v + 55.0f) / 10.0f, special); e_nb = expf_glibc_main(-(v + 65.0f) / 80.0f, special);
            }
            if (special) {
                e_ma = expf_glibc(-(v + 40.0f) / 10.0f); e_mb = expf_glibc(-(v + 65.0f) / 18.0f);
                e_ha = expf_glibc(-(v + 65.0f) / 20.0f); e_hb = expf_glibc(-(v + 35.0f) / 10.0f);
                e_na_ = expf_glibc(-(v + 55.0f) / 10.0f); e_nb = expf_glibc(-(v + 65.0f) / 80.0f);
            }
            const float m_a = 0.1f * ((v + 40.0f) / (1.0f - e_ma));
            const float m_b = 4.0f * e_mb;
            const float h_a = 0.07f * e_ha;
            const float h_b = 1.0f / (e_hb + 1.0f);
            const float m = gate_update(m_state, m_a, m_b, dt);
            const float h = gate_update(h_state, h_a, h_b, dt);
            const float n_a = 0.01f * (v + 55.0f) / (1.0f - e_na_);
            const float n_b = 0.125f * e_nb;
            const float ng = gate_update(n_state, n_a, n_b, dt);
            float m3 = 0.0f, n4 = 0.0f;
            bool special_pow = true;
            if constexpr (HOIST) {
                special_pow = false;
                m3 = powf_glibc_main(m, 3.0f, special_pow); n4 = powf_glibc_main(ng, 4.0f, special_pow);
            }
            if (special_pow) { m3 = pow3f_glibc(m); n4 = pow4f_glibc(ng); }
            const float i_na = m3 * h * hh_g_na * (v - hh_e_na);
            const float i_k = n4 * hh_g_k * (v - hh_e_k);

            const float i_kl = hh_g_kl * (v - hh_e_kl);

            a.n.m_alpha[q] = m_a; a.n.m_beta[q] = m_b; a.n.h_alpha[q] = h_a; a.n.h_beta[q] = h_b;
            a.n.n_alpha[q] = n_a; a.n.n_beta[q] = n_b;
            a.n.m_state[q] = m; a.n.h_state[q] = h; a.n.n_state[q] = ng;
            a.n.na_current[q] = i_na; a.n.k_current[q] = i_k; a.n.k_leak_current[q] = i_kl;

            const float i_ligand_gates = receptor_currents(a, q, dt, c_m);
            const float i_sum = i_in - (i_na + i_k + i_kl);
            v_new = v + (dt * i_sum / c_m - i_ligand_gates);

            neuron_nt_update(a, q, v_new, spiking_prev, dt, t_capture);

            const uint32_t increasing_right_now = v < v_new;
            const uint32_t threshold_crossed = v_new > hh_v_th;
            spike = threshold_crossed && hh_was_increasing && !increasing_right_now;
            a.n.was_increasing[q] = increasing_right_now;
        }

        chem_store(a, q, cs, t_capture);
        a.xout[v_at] = v_new;
        reinterpret_cast<uint32_t *>(a.xout)[s_at] = spike;
        if (a.xout2) {
            a.xout2[v_at] = v_new;
            reinterpret_cast<uint32_t *>(a.xout2)[s_at] = spike;
        }
        if (spike) a.n.last_firing_time[q] = (int32_t)clock;     // neuron/mod.rs:964-966, 2555-2557
        if (vhist_row) vhist_row[q] = v_new;
        if (a.spike_counts && spike) a.spike_counts[q] += 1;
        if (v_stored) *v_stored = v_new;
    }
    return spike;
}

// The loads update_neuron_at<MODEL, ., ., CHEM_OK> starts with, issued EARLY by a kernel that has work between its own start and
// the update (k_step_resident_q: the weights, the products, the turns): every value folded into one word the caller throws away
// once the loads have landed.  A launch starts with cold caches, so the update's first round trip goes to memory (about 2 500
// shader clocks measured in that kernel); requested two phases earlier its cache lines are in the CU's L1 when the update asks.
// Nothing read here is written by this launch before the update reads it.
struct UpdateTouch {
    RcLoaded l;
    uint32_t ntw[K_TYPES][6];
    uint32_t word[3], hh[11];
    bool rc = false, nt = false;
};
// ... the loads (no value is looked at here: nothing waits) ...
template <int MODEL, bool CHEM_OK>
__device__ __forceinline__ void update_touch_load(const UpdateArgs &a, uint32_t ql, UpdateTouch &t)
{
    const uint32_t q = a.rows.global_of(ql);
    t.word[0] = reinterpret_cast<const uint32_t *>(a.n.xbuf)[a.n.xl.at(q, PLANE_SPIKE)];
    t.word[1] = t.word[2] = 0u;
    if (MODEL == CUSTOM_MODEL) return;
    const bool fused_rc = !(SNN_HAVE_CUSTOM_RECEPTORS && a.model_is_custom) && !(SNN_HAVE_CUSTOM_RC && a.rc_kind == CUSTOM_KINETICS);
    const bool fused_nt = !(SNN_HAVE_CUSTOM_NT && a.nt_kind == CUSTOM_KINETICS);
    t.rc = CHEM_OK && a.chemical && fused_rc;
    t.nt = fused_nt;
    if (t.rc) chem_receptors_load(a, q, ql, t.l);
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        // chem_nt_load's loads (it tests the flag where it loads it: a wait)
#pragma unroll
        for (int j = 0; j < 6; ++j) t.ntw[k][j] = 0u;
        if (!t.nt || !a.has_nt || !(a.live_mask >> k & 1u)) continue;
        const size_t i = (size_t)k * a.n.n_pad + q;
        t.ntw[k][0] = a.n.nt_flags[i];
        t.ntw[k][1] = __float_as_uint(a.n.xbuf[a.n.xl.at(q, PLANE_T0 + k)]);
        t.ntw[k][2] = __float_as_uint(a.n.nt_t_max[i]); t.ntw[k][3] = __float_as_uint(a.n.nt_clearance[i]);
        t.ntw[k][4] = __float_as_uint(a.n.nt_v_p[i]); t.ntw[k][5] = __float_as_uint(a.n.nt_k_p[i]);
    }
    if (a.electrical) t.word[1] = a.n_in[ql];
    if (MODEL == 0) t.word[2] = __float_as_uint(a.n.w_value[q]);
    if (MODEL == 2) {
        const float *hh[10] = {a.n.m_state, a.n.h_state, a.n.n_state, a.n.g_na, a.n.e_na, a.n.g_k, a.n.e_k, a.n.g_k_leak, a.n.e_k_leak, a.n.v_th};
#pragma unroll
        for (int i = 0; i < 10; ++i) t.hh[i] = __float_as_uint(hh[i][q]);
        t.hh[10] = a.n.was_increasing[q];
    }
}
// ... and the fold, wherever the caller has to wait for older loads anyway
template <int MODEL>
__device__ __forceinline__ uint32_t update_touch_fold(const UpdateArgs &a, const UpdateTouch &t)
{
    uint32_t h = t.word[0] ^ t.word[1] ^ t.word[2];
    if (MODEL == CUSTOM_MODEL) return h;
    if (t.rc) {
        h ^= __float_as_uint(t.l.mg);
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k)
            h ^= t.l.fl[k] ^ t.l.cnt[k] ^ __float_as_uint(t.l.r[k]) ^ __float_as_uint(t.l.al[k]) ^ __float_as_uint(t.l.be[k]) ^
                 __float_as_uint(t.l.g[k]) ^ __float_as_uint(t.l.e[k]);
    }
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
#pragma unroll
        for (int j = 0; j < 6; ++j) h ^= t.ntw[k][j];
    }
    if (MODEL == 2) {
#pragma unroll
        for (int i = 0; i < 11; ++i) h ^= t.hh[i];
    }
    return h;
}

template <int MODEL, class Sums, bool CHEM_OK = true>
__device__ __forceinline__ uint32_t update_neuron(const UpdateArgs &a, uint32_t ql, const Sums &sums)
{
    return update_neuron_at<MODEL, Sums, true, CHEM_OK>(a, ql, sums, a.clock, a.vhist_row);
}

// Sums held in registers (the one-launch sparse step)
struct RegisterSums {
    float i, t[K_TYPES];
    __device__ __forceinline__ float elec() const { return i; }
    __device__ __forceinline__ float chem(int k) const { return t[k]; }
};

// The second-level sums of a network with chemical synapses, all planes at once: the loads of a batch of every plane the
// step needs (gap junctions + the live transmitter types) are in flight together, the adds of each plane stay strictly
// ascending from 0.0f.  One plane after the other (GlobalSums) the update of configs[2] (64 chunks, two planes) waited for 8
// dependent round trips of 16 loads; this way for 2 of 64.
__device__ __forceinline__ RegisterSums combine_all_planes(const UpdateArgs &a, uint32_t ql)
{
    constexpr uint32_t B = 32;
    RegisterSums out{};
    const float *plane[1 + K_TYPES];
    bool on[1 + K_TYPES];
    plane[0] = a.part_i + ql;
    on[0] = a.electrical != 0;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        plane[1 + k] = a.part_t + (size_t)k * a.n_chunks * a.ld + ql;
        on[1 + k] = (a.live_mask >> k & 1u) != 0u;                       // (the planes of the other types hold zeros)
    }
    float sum[1 + K_TYPES] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t c = 0;
    for (; c + B <= a.n_chunks; c += B) {
        float v[1 + K_TYPES][B];
#pragma unroll
        for (int pl = 0; pl < 1 + K_TYPES; ++pl)
            if (on[pl]) {                                                // launch-uniform
#pragma unroll
                for (uint32_t u = 0; u < B; ++u) v[pl][u] = plane[pl][(size_t)(c + u) * a.ld];
            }
#pragma unroll
        for (int pl = 0; pl < 1 + K_TYPES; ++pl)
            if (on[pl]) {
#pragma unroll
                for (uint32_t u = 0; u < B; ++u) sum[pl] += v[pl][u];
            }
    }
    for (; c < a.n_chunks; ++c) {
#pragma unroll
        for (int pl = 0; pl < 1 + K_TYPES; ++pl)
            if (on[pl]) sum[pl] += plane[pl][(size_t)c * a.ld];
    }
    out.i = sum[0];
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) out.t[k] = sum[1 + k];
    return out;
}

// what follows a wavefront's neuron updates: the handle's own slot of the wire buffer (dense shard handles) and the raster word.
// ql = the local column of this lane; every lane of the wavefront calls it (ballots)
__device__ __forceinline__ void update_epilogue(const UpdateArgs &a, uint32_t ql, bool active, uint32_t spike, float v_new)
{
    if (a.wire_out) {
        // entries past the shard's neurons (slot padding) stay zero: nothing ever writes them
        if (active) {
            const uint32_t q = a.rows.global_of(ql);
            const uint32_t *x = reinterpret_cast<const uint32_t *>(a.xout);
            for (uint32_t pl = 0; pl < a.wire_planes; ++pl)
                a.wire_out[(size_t)pl * a.wire_count + ql] = a.wire_plane_id[pl] == PLANE_V ? __float_as_uint(v_new)
                                                                                           : x[a.n.xl.at(q, (int)a.wire_plane_id[pl])];
        }
        const unsigned long long bits = __ballot(spike != 0);
        const uint32_t lane = threadIdx.x & 63u, w0 = (ql >> 6) * 2, n_words = (a.wire_count + 31) / 32;
        uint32_t *out = a.wire_out + (size_t)a.wire_planes * a.wire_count;
        if (lane == 0 && w0 < n_words) out[w0] = (uint32_t)bits;
        if (lane == 32 && w0 + 1 < n_words) out[w0 + 1] = (uint32_t)(bits >> 32);
    }

    // spike raster: one 64-bit ballot word per wavefront = one aligned 64-block of the global index space (shard
    // boundaries are multiples of 64; a range-set shard maps every wavefront to such a block, holes contribute 0)
    if (a.spike_row) {
        const unsigned long long word = __ballot(spike != 0);
        // a raster row holds n_pad / 64 words; ld may exceed the padded population by one wavefront (row de-alignment)
        if ((threadIdx.x & 63) == 0 && ql < a.ld) {
            const uint32_t g = a.rows.block ? (ql < a.n_loc ? a.rows.block[ql >> 6] * 64u : a.n.n_pad) : a.q0 + ql;
            if (g < a.n.n_pad) a.spike_row[g >> 6] = word;
        }
    }
}

// ALL_PLANES: a dense handle with chemical synapses (its own instantiation: the batch registers of combine_all_planes would
// otherwise cost the electrical-only update its occupancy)
template <int MODEL, bool ALL_PLANES = false>
__global__ __launch_bounds__(256) void k_update(const UpdateArgs a)
{
    warm_kernel_arguments<sizeof(UpdateArgs)>();                         // (snn_layout.hpp: the arguments' lines in one round trip)
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;           // blockDim.x = 64 or 256
    const bool active = ql < a.n_loc && a.rows.active(ql, a.n_loc);
    float v_new = 0.0f;
    uint32_t spike = 0u;
    if (active) {
        if (ALL_PLANES) spike = update_neuron_at<MODEL>(a, ql, combine_all_planes(a, ql), a.clock, a.vhist_row, &v_new);
        else spike = update_neuron_at<MODEL>(a, ql, GlobalSums{a, ql}, a.clock, a.vhist_row, &v_new);
    }
    update_epilogue(a, ql, active, spike, v_new);
}

// The update of a dense handle with chemical synapses, WIDE (round 6): a workgroup is 64 columns x 4 wavefronts; wavefront s
// requests ITS quarter of every live plane's chunk partials at once (configs[2]: 64 chunks, two planes -- 32 loads per lane, one
// round trip where k_update<MODEL, true> made two of 64), the running sums pass from wavefront to wavefront through LDS -- wavefront
// 0 adds chunks [0, per) from 0.0f, wavefront 1 continues with [per, 2 per) ...: the canonical ascending order, add for add -- and
// wavefront 3 updates the neurons.  While the partials are on their way the other three wavefronts TOUCH the arrays the update is
// about to read (TouchList, one load per array and lane, result dropped): the update's loads then hit the CU's cache instead of
// making a second trip to memory behind the first.  Four times the workgroups of k_update (configs[2]: 256, one per CU).
struct TouchList {
    const uint32_t *by_neuron[56];     // arrays indexed by the global neuron
    const uint32_t *by_column[8];      // ... by the local column (static counts)
    uint32_t n_neuron, n_column;
};

template <int MODEL>
__global__ __launch_bounds__(256) void k_update_wide(const UpdateArgs a, const TouchList touch)
{
    constexpr uint32_t B = 16;
    __shared__ float run[1 + K_TYPES][64];
    const uint32_t lane = threadIdx.x & 63u, seg = threadIdx.x >> 6;
    const uint32_t ql = blockIdx.x * 64u + lane;                         // < ld: the grid is ld / 64 workgroups
    const uint32_t per = (a.n_chunks + 3u) / 4u;
    const uint32_t c0 = min(seg * per, a.n_chunks), c1 = min(c0 + per, a.n_chunks);
    const float *plane[1 + K_TYPES];
    bool on[1 + K_TYPES];
    plane[0] = a.part_i + ql;
    on[0] = a.electrical != 0;
#pragma unroll
    for (int k = 0; k < K_TYPES; ++k) {
        plane[1 + k] = a.part_t + (size_t)k * a.n_chunks * a.ld + ql;
        on[1 + k] = a.chemical && (a.live_mask >> k & 1u) != 0u;         // (the planes of the other types hold zeros)
    }
    // the first batch of this wavefront's partials: requested before anything is waited for
    float v[1 + K_TYPES][B];
#pragma unroll
    for (int pl = 0; pl < 1 + K_TYPES; ++pl)
        if (on[pl]) {                                                    // launch-uniform
#pragma unroll
            for (uint32_t u = 0; u < B; ++u) v[pl][u] = plane[pl][(size_t)min(c0 + u, a.n_chunks - 1u) * a.ld];
        }
    const bool active = ql < a.n_loc && a.rows.active(ql, a.n_loc);
    if (seg != 3u && active) {
        const uint32_t q = a.rows.global_of(ql);
        for (uint32_t i = seg; i < touch.n_neuron; i += 3u) { const uint32_t x = touch.by_neuron[i][q]; asm volatile("" ::"v"(x)); }
        for (uint32_t i = seg; i < touch.n_column; i += 3u) { const uint32_t x = touch.by_column[i][ql]; asm volatile("" ::"v"(x)); }
    }
    float sum[1 + K_TYPES] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t turn = 0; turn < 4u; ++turn) {
        if (seg == turn) {
            if (turn) {
#pragma unroll
                for (int pl = 0; pl < 1 + K_TYPES; ++pl) sum[pl] = run[pl][lane];
            }
            for (uint32_t c = c0; c < c1; c += B) {
                if (c != c0) {                                           // (more than B chunks per wavefront: further batches)
#pragma unroll
                    for (int pl = 0; pl < 1 + K_TYPES; ++pl)
                        if (on[pl]) {
#pragma unroll
                            for (uint32_t u = 0; u < B; ++u) v[pl][u] = plane[pl][(size_t)min(c + u, a.n_chunks - 1u) * a.ld];
                        }
                }
#pragma unroll
                for (int pl = 0; pl < 1 + K_TYPES; ++pl)
                    if (on[pl]) {
#pragma unroll
                        for (uint32_t u = 0; u < B; ++u)
                            if (c + u < c1) sum[pl] += v[pl][u];         // (wave-uniform bound)
                    }
            }
            if (turn < 3u) {
#pragma unroll
                for (int pl = 0; pl < 1 + K_TYPES; ++pl) run[pl][lane] = sum[pl];
            }
        }
        if (turn < 3u) __syncthreads();
    }
    if (seg != 3u) return;
    float v_new = 0.0f;
    uint32_t spike = 0u;
    if (active) {
        RegisterSums s;
        s.i = sum[0];
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) s.t[k] = sum[1 + k];
        spike = update_neuron_at<MODEL>(a, ql, s, a.clock, a.vhist_row, &v_new);
    }
    update_epilogue(a, ql, active, spike, v_new);
}

} // namespace snn

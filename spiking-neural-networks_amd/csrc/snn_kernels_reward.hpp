// Reward modulation: the dopamine ODE, the per-synapse R-STDP update as a standalone streaming pass and fused into
// the synaptic-input pass.
#pragma once
#include "snn_kernels_inputs.hpp"

namespace snn {

// ---- reward modulation ------------------------------------------------------------------------------
// RewardModulatedLattice (neuron/mod.rs:2719-3417) with RewardModulatedSTDP + TraceRSTDP (plasticity/mod.rs:126-242).
// Per lattice: RM_STRIDE floats {dopamine, tau_d, tau_c, a_plus, a_minus, tau_plus, tau_minus, dt, exp(-dt/tau_c),
// dopamine before the latest reward}.  The last slot serves the deferred weight update (k_inputs_rstdp): the update
// of step t is applied at the start of step t+1 and must use the dopamine of step t even if the reward of step t+1
// has been applied in between.
constexpr int RM_STRIDE = 10;
constexpr int RM_DOPAMINE = 0, RM_DOPAMINE_BEFORE = 9;

// RewardModulatedSTDP::update (plasticity/mod.rs:199-201) on every modulated lattice; reward < 0 or > 0 alike.
// `refresh_only`: recompute the cached trace decay after the parameters changed.
__global__ void k_modulator_update(float *rm, const uint32_t *rm_on, uint32_t n_lattices, float reward, int refresh_only)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lattices) return;
    float *m = rm + (size_t)l * RM_STRIDE;
    m[8] = expf_glibc(-m[7] / m[2]);
    if (refresh_only || !rm_on[l]) return;
    m[RM_DOPAMINE_BEFORE] = m[RM_DOPAMINE];
    m[RM_DOPAMINE] = m[RM_DOPAMINE] * expf_glibc(-m[7] / m[1]) + m[1] * reward;
}

// The two update_weight visits every internal edge of a modulated lattice receives per step (do_update is always
// true, plasticity/mod.rs:239-241), deferred form: both visits see the same delta
//   dw = 0 + delta; w += c * dopamine; dw += delta; c = c * exp(-dt / tau_c) + tau_c * dw; w += c * dopamine
__device__ __forceinline__ void rstdp_edge(float &w, float &c, int32_t tp, int32_t tq, const float *m, int dop)
{
    const float delta_w = stdp_delta(tp, tq, m[3], m[4], m[5], m[6], m[7]);
    const float dopamine = m[dop];
    float dw = 0.0f;
    dw += delta_w;
    w += c * dopamine;
    dw += delta_w;
    c = c * m[8] + m[2] * dw;
    w += c * dopamine;
}

struct RewardArgs {
    float *W, *C;                          // weights and TraceRSTDP::c, both [n_tot rows][ld]
    uint32_t ld, n_loc, q0, n_neurons;
    const int32_t *last_firing_time;
    const uint32_t *lattice_slot;
    const float *rm;
    const uint32_t *rm_on;
    int dop;                               // RM_DOPAMINE, or RM_DOPAMINE_BEFORE when a newer reward is already in
};

// One streaming read-modify-write pass over the neuron rows of W and C: 16 B per synapse.
// Thread = one column, one unit (4 consecutive rows, dwordx4) per iteration, row groups grid-strided over blockIdx.y;
// the presynaptic side (the 4 rows of a group) is wave-uniform, the postsynaptic side lives in registers.
__global__ __launch_bounds__(256) void k_rstdp_dense(const RewardArgs a)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n_loc) return;
    const uint32_t q = a.q0 + c;
    const int32_t tq = a.last_firing_time[q];
    const uint32_t sq = a.lattice_slot[q];
    const uint32_t groups = (a.n_neurons + 3u) >> 2;
    for (uint32_t g = blockIdx.y; g < groups; g += gridDim.y) {
        uint32_t sp[4];
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = g * 4 + k;
            sp[k] = (p < a.n_neurons && a.rm_on[a.lattice_slot[p]]) ? a.lattice_slot[p] : 0xFFFFFFFFu;
            any = any || sp[k] == sq;
        }
        if (!any) continue;
        v4f *wp = reinterpret_cast<v4f *>(a.W) + (size_t)g * a.ld + c;
        v4f *cp = reinterpret_cast<v4f *>(a.C) + (size_t)g * a.ld + c;
        v4f w = *wp, tr = *cp;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (sp[k] != sq || w[k] != w[k]) continue;       // other lattice / not modulated / absent edge (NaN)
            float wk = w[k], ck = tr[k];
            rstdp_edge(wk, ck, a.last_firing_time[g * 4 + k], tq, a.rm + (size_t)sq * RM_STRIDE, a.dop);
            w[k] = wk; tr[k] = ck;
        }
        *wp = w;
        *cp = tr;
    }
}

// RewardModulatedLatticeNetwork::update_weights_from_neurons_across_reward_lattices, the incoming half (neuron/mod.rs:4859-4924):
// every neuron q of a modulated lattice visits, once per step, its connections from OTHER lattices and from spike-train cells --
//   kind 2 (RewardModulatedConnection::Weight), source in a plain neuron lattice lp: w += lp's STDP delta (:4869-4883);
//   kind 1 (RewardModulatedConnection::RewardModulatedWeight): one visit of q's lattice's modulator (plasticity/mod.rs:203-237):
//       dw += delta; every second visit  c = c * exp(-dt / tau_c) + tau_c * dw, dw = 0;  w += c * dopamine.
// One visit per step: TraceRSTDP::dw lives across steps (`pending`, the layout of W), and the counter is the same for every such
// connection of a lattice (bit l of `second`).  Thread = one column, 4 rows per unit, as k_rstdp_dense.
struct RewardCrossArgs {
    float *W, *C, *P;                      // weights, TraceRSTDP::c, TraceRSTDP::dw
    uint32_t ld, n_loc, q0, n_neurons, n_tot, n_lattices;
    const int32_t *last_firing_time, *st_last_firing_time;
    const uint32_t *lattice_slot, *st_lattice_slot;
    const float *rm, *stdp;
    const uint32_t *rm_on;
    const uint8_t *conn_kind;
    unsigned long long second;             // bit l: this is the second of a pair of visits for lattice l
};

__global__ __launch_bounds__(256) void k_reward_cross(const RewardCrossArgs a)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n_loc) return;
    const uint32_t q = a.q0 + c;
    const uint32_t sq = a.lattice_slot[q];
    if (!a.rm_on[sq]) return;
    const int32_t tq = a.last_firing_time[q];
    const float *m = a.rm + (size_t)sq * RM_STRIDE;
    const bool second = (a.second >> sq & 1ull) != 0ull;
    const uint32_t groups = (a.n_tot + 3u) >> 2;
    for (uint32_t g = blockIdx.y; g < groups; g += gridDim.y) {
        uint32_t kind[4];
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = g * 4 + k;
            kind[k] = 0;
            if (p < a.n_tot) {
                const uint32_t source = p < a.n_neurons ? a.lattice_slot[p] : a.n_lattices + a.st_lattice_slot[p - a.n_neurons];
                if (!(p < a.n_neurons && source == sq)) kind[k] = a.conn_kind[(size_t)source * a.n_lattices + sq];
            }
            any = any || kind[k] != 0;
        }
        if (!any) continue;
        v4f *wp = reinterpret_cast<v4f *>(a.W) + (size_t)g * a.ld + c;
        v4f *cp = reinterpret_cast<v4f *>(a.C) + (size_t)g * a.ld + c;
        v4f *pp = reinterpret_cast<v4f *>(a.P) + (size_t)g * a.ld + c;
        v4f w = *wp, tr = *cp, pd = *pp;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = g * 4 + k;
            if (kind[k] == 0 || w[k] != w[k]) continue;                      // a plain network's edge / absent edge (NaN)
            const int32_t tp = p < a.n_neurons ? a.last_firing_time[p] : a.st_last_firing_time[p - a.n_neurons];
            if (kind[k] == 2) {
                if (p >= a.n_neurons || a.rm_on[a.lattice_slot[p]]) continue;
                const float *prm = a.stdp + PL_STRIDE * a.lattice_slot[p];
                w[k] = w[k] + stdp_delta(tp, tq, prm[0], prm[1], prm[2], prm[3], prm[4]);
                continue;
            }
            float dw = pd[k], ck = tr[k];
            dw += stdp_delta(tp, tq, m[3], m[4], m[5], m[6], m[7]);
            if (second) {
                ck = ck * m[8] + m[2] * dw;
                dw = 0.0f;
            }
            w[k] = w[k] + ck * m[RM_DOPAMINE];
            pd[k] = dw; tr[k] = ck;
        }
        *wp = w; *cp = tr; *pp = pd;
    }
}

// The same update FUSED into the next step's synaptic-input pass: W and the trace are read, updated, written back,
// and the fresh weight feeds the input sums -- 16 B per synapse per step for a reward-modulated lattice instead of
// 16 (k_rstdp_dense) + 4 (k_inputs_dense).  The host defers the update of step t to the start of step t+1 (nothing
// reads W in between; any host access flushes it with k_rstdp_dense).  Staging, shapes and the accumulation are
// those of k_inputs_dense (kept separate so that the plain input pass stays untouched); single register buffer.
struct RstdpInputsArgs {
    InputsArgs in;
    float *W, *C;
    const int32_t *last_firing_time;
    const uint32_t *lattice_slot;
    const float *rm;
    const uint32_t *rm_on;
    int dop;
};

template <bool ELEC, bool CHEM, int STREAM>
__global__ __launch_bounds__(InputsShape<STREAM>::THREADS) void k_inputs_rstdp(const RstdpInputsArgs ra)
{
    using S = InputsShape<STREAM>;
    constexpr int VEC = S::VEC;
    constexpr uint32_t GB = 2;                 // row groups (8 rows) of W and of the traces in flight per lane
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const InputsArgs &a = ra.in;

    __shared__ float s_val[CHUNK];
    __shared__ uint32_t s_kind[CHUNK];
    __shared__ float s_t[CHEM ? K_TYPES : 1][CHUNK];
    __shared__ uint32_t s_mod[CHUNK];          // lattice slot of a row whose lattice is modulated, else NONE
    __shared__ int32_t s_lft[CHUNK];

    const uint32_t chunk = blockIdx.y;
    const uint32_t p0 = chunk * CHUNK;
    const uint32_t rows = min((uint32_t)CHUNK, a.n_tot - p0);
    const uint32_t tid = threadIdx.x;

    for (uint32_t i = tid; i < rows; i += S::THREADS) {
        const uint32_t p = p0 + i;
        float val;
        uint32_t kind, mod = NONE;
        int32_t lft = -1;
        if (p < a.n_neurons) {
            val = a.xbuf[a.xl.at(p, PLANE_V)];
            kind = KIND_NEURON;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    kind |= a.nt_flags[(size_t)k * a.n_pad + p] ? (0x100u << k) : 0u;
                    s_t[k][i] = a.xbuf[a.xl.at(p, PLANE_T0 + k)];
                }
            }
            const uint32_t slot = ra.lattice_slot[p];
            if (ra.rm_on[slot]) mod = slot;
            lft = ra.last_firing_time[p];
        } else {
            const uint32_t s = p - a.n_neurons;
            val = a.st_value[s];
            kind = (a.st_last_firing_time[s] < 0) ? KIND_ST_SILENT : KIND_ST_FIRED;
            if (CHEM) {
#pragma unroll
                for (int k = 0; k < K_TYPES; ++k) {
                    kind |= a.st_nt_flags[(size_t)k * a.c_pad + s] ? (0x100u << k) : 0u;
                    s_t[k][i] = a.st_nt_t[(size_t)k * a.c_pad + s];
                }
            }
        }
        s_val[i] = val;
        s_kind[i] = kind;
        s_mod[i] = mod;
        s_lft[i] = lft;
    }
    const uint32_t groups = (rows + 3u) >> 2;
    for (uint32_t i = rows + tid; i < groups * 4; i += S::THREADS) {     // padding rows of the last group (weights NaN)
        s_val[i] = 0.0f;
        s_kind[i] = KIND_NEURON;
        s_mod[i] = NONE;
        s_lft[i] = -1;
    }
    __syncthreads();

    const uint32_t tile = (blockIdx.x + blockIdx.y) % gridDim.x;
    const uint32_t ql = tile * S::TILE + tid;                // the lane's column j is ql + j * THREADS
    if (ql >= a.n_loc) return;
    bool colv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) colv[j] = ql + (uint32_t)j * S::THREADS < a.n_loc;

    float vq[VEC], gq[VEC];
    int32_t tq[VEC];
    uint32_t sq[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t q = ql + (uint32_t)j * S::THREADS;
        const bool in = q < a.n_loc;
        const uint32_t gqi = a.q0 + (in ? q : a.n_loc - 1);
        vq[j] = (ELEC && in) ? a.xbuf[a.xl.at(gqi, PLANE_V)] : 0.0f;
        gq[j] = (ELEC && in) ? a.gap_conductance[gqi] : 0.0f;
        tq[j] = ra.last_firing_time[gqi];
        sq[j] = in ? ra.lattice_slot[gqi] : NONE - 1;
    }

    float acc[VEC];
    float tacc[CHEM ? K_TYPES : 1][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < (CHEM ? K_TYPES : 1); ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) tacc[k][j] = 0.0f;

    const size_t ld = a.ld;
    // wave-uniform bases + the lane's fixed index (see k_inputs_dense); columns past the shard's width read slack
    v4f *wbase = reinterpret_cast<v4f *>(ra.W) + (size_t)(p0 >> 2) * ld + (size_t)tile * S::TILE;
    v4f *cbase = reinterpret_cast<v4f *>(ra.C) + (size_t)(p0 >> 2) * ld + (size_t)tile * S::TILE;

    // one presynaptic row of the input sums (weights already updated)
    auto row = [&](uint32_t r, const float (&w)[VEC]) {
        const uint32_t kind = __builtin_amdgcn_readfirstlane(s_kind[r]);
        if (ELEC) {
            const float vp = s_val[r];
            const uint32_t src = kind & 3u;
            if (src == KIND_NEURON) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * (vp - vq[j]), w[j]);
            } else if (src == KIND_ST_SILENT) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], vp, w[j]);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = acc_if_edge(acc[j], gq[j] * vp, w[j]);
            }
        }
        if (CHEM) {
#pragma unroll
            for (int k = 0; k < K_TYPES; ++k) {
                if (kind & (0x100u << k)) {
                    const float t = s_t[k][r];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) tacc[k][j] = acc_if_edge(tacc[k][j], t, w[j]);
                }
            }
        }
    };

    // one row group: the reward-modulated update of its rows (units of modulated rows are rewritten in full), then the sums
    auto group = [&](uint32_t grp, v4f (&w)[VEC], v4f (&c)[VEC]) {
        bool dirty = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t r = grp * 4 + k;
            const uint32_t mod = __builtin_amdgcn_readfirstlane(s_mod[r]);
            if (mod != NONE) {
                const float *m = ra.rm + (size_t)mod * RM_STRIDE;
                const int32_t tp = s_lft[r];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float wk = w[j][k], ck = c[j][k];
                    if (sq[j] == mod && wk == wk) rstdp_edge(wk, ck, tp, tq[j], m, ra.dop);
                    w[j][k] = wk; c[j][k] = ck;
                }
                dirty = true;
            }
        }
        if (dirty) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (!colv[j]) continue;
                __builtin_nontemporal_store(w[j], wbase + (size_t)grp * ld + j * S::THREADS + tid);
                __builtin_nontemporal_store(c[j], cbase + (size_t)grp * ld + j * S::THREADS + tid);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float wr[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) wr[j] = w[j][k];
            row(grp * 4 + k, wr);
        }
    };

    uint32_t g = 0;
    for (; g + GB <= groups; g += GB) {
        v4f wb[GB][VEC], cb[GB][VEC];
#pragma unroll
        for (uint32_t u = 0; u < GB; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) wb[u][j] = __builtin_nontemporal_load(wbase + (size_t)(g + u) * ld + j * S::THREADS + tid);
#pragma unroll
        for (uint32_t u = 0; u < GB; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) cb[u][j] = __builtin_nontemporal_load(cbase + (size_t)(g + u) * ld + j * S::THREADS + tid);
#pragma unroll
        for (uint32_t u = 0; u < GB; ++u) group(g + u, wb[u], cb[u]);
    }
    for (; g < groups; ++g) {
        v4f w[VEC], c[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            w[j] = __builtin_nontemporal_load(wbase + (size_t)g * ld + j * S::THREADS + tid);
            c[j] = __builtin_nontemporal_load(cbase + (size_t)g * ld + j * S::THREADS + tid);
        }
        group(g, w, c);
    }

    if (ELEC) {
        float *dst = a.part_i + (size_t)chunk * a.ld + ql;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (colv[j]) dst[(uint32_t)j * S::THREADS] = acc[j];
    }
    if (CHEM) {
#pragma unroll
        for (int k = 0; k < K_TYPES; ++k) {
            float *dst = a.part_t + ((size_t)k * a.n_chunks + chunk) * a.ld + ql;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (colv[j]) dst[(uint32_t)j * S::THREADS] = tacc[k][j];
        }
    }
}

} // namespace snn

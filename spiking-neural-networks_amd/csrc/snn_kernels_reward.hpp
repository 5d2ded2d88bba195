// Reward modulation: the dopamine ODE, the per-synapse R-STDP update as a standalone streaming pass and fused into
// the synaptic-input pass.
#pragma once
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_misc.hpp"

namespace snn {

// ---- reward modulation ------------------------------------------------------------------------------
// RewardModulatedLattice (neuron/mod.rs:2719-3417) with RewardModulatedSTDP + TraceRSTDP (plasticity/mod.rs:126-242).
// Per lattice: RM_STRIDE floats {dopamine, tau_d, tau_c, a_plus, a_minus, tau_plus, tau_minus, dt, exp(-dt/tau_c),
// dopamine before the latest reward}.  The last slot serves the deferred weight update (k_inputs_rstdp): the update
// of step t is applied at the start of step t+1 and must use the dopamine of step t even if the reward of step t+1
// has been applied in between.
constexpr int RM_STRIDE = 10;
constexpr int RM_DOPAMINE = 0, RM_DOPAMINE_BEFORE = 9;
// rm_on[lattice]: bit 0 RewardModulatedLattice::do_modulation (neuron/mod.rs:2744), bit 1 the lattice IS a reward-modulated
// lattice (held by the network's reward_modulated_lattices map, :3419-3453): set by snn_set_reward_modulator whatever its
// do_modulation argument.  A modulated lattice with do_modulation off updates no weight of its own and is never visited, but its
// modulator takes every reward and it stays a modulated PARTNER of other lattices' visits.
constexpr uint32_t RM_DO_MODULATION = 1u, RM_IS_MODULATED = 2u;

// RewardModulatedSTDP::update (plasticity/mod.rs:199-201) on every modulated lattice; reward < 0 or > 0 alike.
// `refresh_only`: recompute the cached trace decay after the parameters changed.
__global__ void k_modulator_update(float *rm, const uint32_t *rm_on, uint32_t n_lattices, float reward, int refresh_only)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lattices) return;
    float *m = rm + (size_t)l * RM_STRIDE;
    m[8] = expf_glibc(-m[7] / m[2]);
    if (refresh_only || !(rm_on[l] & RM_IS_MODULATED)) return;          // (a paused modulator still takes the reward, neuron/mod.rs:5287-5291)
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
            sp[k] = (p < a.n_neurons && (a.rm_on[a.lattice_slot[p]] & RM_DO_MODULATION)) ? a.lattice_slot[p] : 0xFFFFFFFFu;
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

// The connections BETWEEN lattices of a RewardModulatedLatticeNetwork (snn_set_connection_kind: 1 = RewardModulatedConnection::
// RewardModulatedWeight, 2 = ::Weight).  post_neuron_update_step (neuron/mod.rs:5030-5043) visits the spiking neurons of plastic
// plain lattices (update_weights_from_neurons_across_lattices, :4707-4802), then every neuron of the reward-modulated lattices
// (_across_reward_lattices, :4855-4977).  A visit of z handles, for each partner o in another lattice,
//   incoming o -> z:  Weight: z plain: z's lattice's STDP delta; z modulated: o's lattice's STDP delta when o sits in a plain
//                     lattice;  RewardModulatedWeight: one visit of the modulator of z's lattice (z modulated) or o's (z plain);
//   outgoing z -> o:  the reference looks up the REVERSE connection o -> z (:4768-4771, :4929-4932), applies the rule once more
//                     to that copy with (pre = z, post = o) and stores it as z -> o (weight, trace, dw and counter).
// The two connections of a PAIR of neurons are touched by the visits of those two neurons only, so a thread owns one pair (x < y)
// and replays its visits in the reference's order: plain visits first, then modulated ones, each by neuron index.  Cells are
// never visited and have no incoming connections: a pair (x, cell) is the incoming half of x alone.
// One modulator visit (plasticity/mod.rs:203-237): dw += delta; counter 0 -> 1; counter 1: c = c * exp(-dt / tau_c) + tau_c * dw,
// dw = 0, counter = 0; then w += c * dopamine.  dw (`P`) and the counter (`K`, 0.0 / 1.0) are per connection, in the layout of W.
struct RewardCrossArgs {
    StdpArgs s;                            // W, ld, sizes, firing times, lattice slots, STDP table, do_plasticity, spike plane, kinds
    float *C, *P, *K;                      // TraceRSTDP::c, ::dw, ::counter
    const float *rm;
    const uint32_t *rm_on;
    uint32_t *bad;                         // k_reward_cross_check: the highest refusal class found
};

struct CrossEdge {
    float w, c, dw, k;
    bool exists;
    uint32_t kind;
};

__device__ __forceinline__ size_t cross_at(const StdpArgs &s, uint32_t pre, uint32_t post)
{
    return ((size_t)(pre >> 2) * s.ld + post) * 4u + (pre & 3u);
}
__device__ __forceinline__ uint32_t cross_kind(const StdpArgs &s, uint32_t pre, uint32_t post_slot)
{
    const uint32_t source = pre < s.n_neurons ? s.lattice_slot[pre] : s.n_lattices + s.st_lattice_slot[pre - s.n_neurons];
    return s.conn_kind[(size_t)source * s.n_lattices + post_slot];
}
__device__ __forceinline__ void cross_trace_visit(CrossEdge &e, const float *m, int t_pre, int t_post)
{
    e.dw += stdp_delta(t_pre, t_post, m[3], m[4], m[5], m[6], m[7]);
    if (e.k == 0.0f) {
        e.k = 1.0f;
    } else {
        e.c = e.c * m[8] + m[2] * e.dw;
        e.k = 0.0f;
        e.dw = 0.0f;
    }
    e.w = e.w + e.c * m[RM_DOPAMINE];
}

// one visit of z (lattice lz) with partner o: `in` = o -> z, `out` = z -> o
__device__ __forceinline__ void cross_visit(const RewardCrossArgs &a, CrossEdge &in, CrossEdge &out, uint32_t lz, uint32_t lo,
                                            bool o_is_neuron, int tz, int to)
{
    const bool mod_z = (a.rm_on[lz] & RM_IS_MODULATED) != 0u, mod_o = o_is_neuron && (a.rm_on[lo] & RM_IS_MODULATED) != 0u;
    const float *plain = a.s.stdp + PL_STRIDE * (mod_z ? lo : lz);
    const float *m = a.rm + (size_t)(mod_z ? lz : lo) * RM_STRIDE;
    if (in.exists && in.kind == 2u && (!mod_z || (o_is_neuron && !mod_o)))
        in.w = in.w + stdp_delta(to, tz, plain[0], plain[1], plain[2], plain[3], plain[4]);
    if (in.exists && in.kind == 1u) cross_trace_visit(in, m, to, tz);
    if (!(out.exists && in.exists && o_is_neuron)) return;
    if (out.kind == 2u && (!mod_z || !mod_o)) out.w = in.w + stdp_delta(tz, to, plain[0], plain[1], plain[2], plain[3], plain[4]);
    if (out.kind == 1u) {
        CrossEdge copy = in;
        cross_trace_visit(copy, m, tz, to);
        out.w = copy.w; out.c = copy.c; out.dw = copy.dw; out.k = copy.k;
    }
}

__global__ __launch_bounds__(256) void k_reward_cross(const RewardCrossArgs a)
{
    const StdpArgs &s = a.s;
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= s.n_neurons) return;
    const uint32_t lx = s.lattice_slot[x];
    const int tx = s.last_firing_time[x];
    // which map of the reference's network holds a lattice (modulated or plain) decides its role; whether the neurons of a modulated
    // lattice are visited is its do_modulation (neuron/mod.rs:5113)
    const bool x_is_mod = (a.rm_on[lx] & RM_IS_MODULATED) != 0u, x_mod = x_is_mod && (a.rm_on[lx] & RM_DO_MODULATION) != 0u;
    const bool x_plain = !x_is_mod && s.do_plasticity[lx] && reinterpret_cast<const uint32_t *>(s.xbuf)[s.xl.at(x, PLANE_SPIKE)] != 0u;
    for (uint32_t y = blockIdx.y; y < s.n_tot; y += gridDim.y) {
        if (y <= x) continue;
        const bool y_neuron = y < s.n_neurons;
        const uint32_t ly = y_neuron ? s.lattice_slot[y] : 0u;
        if (y_neuron && ly == lx) continue;
        CrossEdge yx{}, xy{};
        const size_t i_yx = cross_at(s, y, x), i_xy = y_neuron ? cross_at(s, x, y) : 0;
        yx.kind = cross_kind(s, y, lx);
        yx.w = s.W[i_yx];
        yx.exists = yx.kind != 0u && yx.w == yx.w;                     // (NaN = no such connection)
        if (y_neuron) {
            xy.kind = cross_kind(s, x, ly);
            xy.w = s.W[i_xy];
            xy.exists = xy.kind != 0u && xy.w == xy.w;
        }
        if (!yx.exists && !xy.exists) continue;
        const bool y_is_mod = y_neuron && (a.rm_on[ly] & RM_IS_MODULATED) != 0u, y_mod = y_is_mod && (a.rm_on[ly] & RM_DO_MODULATION) != 0u;
        const bool y_plain = y_neuron && !y_is_mod && s.do_plasticity[ly] &&
                             reinterpret_cast<const uint32_t *>(s.xbuf)[s.xl.at(y, PLANE_SPIKE)] != 0u;
        if (!(x_mod || x_plain || y_mod || y_plain)) continue;
        if (yx.exists) { yx.c = a.C[i_yx]; yx.dw = a.P[i_yx]; yx.k = a.K[i_yx]; }
        if (xy.exists) { xy.c = a.C[i_xy]; xy.dw = a.P[i_xy]; xy.k = a.K[i_xy]; }
        const int ty = y_neuron ? s.last_firing_time[y] : s.st_last_firing_time[y - s.n_neurons];
        if (x_plain) cross_visit(a, yx, xy, lx, ly, y_neuron, tx, ty);
        if (y_plain) cross_visit(a, xy, yx, ly, lx, true, ty, tx);
        if (x_mod) cross_visit(a, yx, xy, lx, ly, y_neuron, tx, ty);
        if (y_mod) cross_visit(a, xy, yx, ly, lx, true, ty, tx);
        if (yx.exists) { s.W[i_yx] = yx.w; a.C[i_yx] = yx.c; a.P[i_yx] = yx.dw; a.K[i_yx] = yx.k; }
        if (xy.exists) { s.W[i_xy] = xy.w; a.C[i_xy] = xy.c; a.P[i_xy] = xy.dw; a.K[i_xy] = xy.k; }
    }
}

// Where the reference's visits are defined (it unwraps None elsewhere); the highest class found goes to *bad:
//   1 a connection u -> v of a visited lattice (modulated, or plain with do_plasticity) without its reverse v -> u of the same kind
//     (neuron/mod.rs:4768-4771, :4929-4932);
//   2 RewardModulatedWeight with no modulator on either side while one side is plastic, or from a spike train into a plastic
//     plain lattice (:4743, :4789);
//   3 Weight between a plastic plain lattice and a reward-modulated one (:4729-4733, :4778);
//   4 a plastic plain lattice with the BCM rule on such a connection.
__global__ __launch_bounds__(256) void k_reward_cross_check(const RewardCrossArgs a)
{
    const StdpArgs &s = a.s;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= s.n_neurons) return;
    const uint32_t lq = s.lattice_slot[q];
    const bool mod_q = (a.rm_on[lq] & RM_IS_MODULATED) != 0u, plastic_q = !mod_q && s.do_plasticity[lq] != 0u;
    uint32_t bad = 0;
    for (uint32_t p = blockIdx.y; p < s.n_tot; p += gridDim.y) {
        const uint32_t kind = cross_kind(s, p, lq);
        const float w = s.W[cross_at(s, p, q)];
        if (kind == 0u || w != w) continue;
        if (plastic_q && s.stdp[PL_STRIDE * lq + 5] != 0.0f) bad = max(bad, 4u);
        if (p >= s.n_neurons) {
            if (kind == 1u && plastic_q) bad = max(bad, 2u);
            continue;
        }
        const uint32_t lp = s.lattice_slot[p];
        if (lp == lq) continue;
        const bool mod_p = (a.rm_on[lp] & RM_IS_MODULATED) != 0u, plastic_p = !mod_p && s.do_plasticity[lp] != 0u;
        if (plastic_p && s.stdp[PL_STRIDE * lp + 5] != 0.0f) bad = max(bad, 4u);
        if (kind == 1u && !mod_p && !mod_q && (plastic_p || plastic_q)) bad = max(bad, 2u);
        if (kind == 2u && ((plastic_p && mod_q) || (plastic_q && mod_p))) bad = max(bad, 3u);
        if ((mod_p && (a.rm_on[lp] & RM_DO_MODULATION)) || plastic_p) {          // p's lattice is visited: its outgoing half needs q -> p
            const float r = s.W[cross_at(s, q, p)];
            if (r != r || s.conn_kind[(size_t)lq * s.n_lattices + lp] != kind) bad = max(bad, 1u);
        }
    }
    if (bad) atomicMax(a.bad, bad);
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
            if (ra.rm_on[slot] & RM_DO_MODULATION) mod = slot;
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

// C ABI of the MI355X-native spiking-lattice stepper (include/snn_amd.h).
// Host side of the handle: index space, device allocations, attribute registry (the reference's
// HashMap<String, BufferGPU> of IterateAndSpikeGPU::convert_to_gpu, neuron/iterate_and_spike/
// mod.rs:3156-3189), graph import/export, the step loop of run_lattice / run_lattices
// (neuron/gpu_lattices/mod.rs:791-896, 2284-2583) and histories.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/snn_amd.h"
#include "snn_kernels_csr.hpp"
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_misc.hpp"
#include "snn_kernels_resident.hpp"
#include "snn_kernels_update.hpp"
#include "snn_layout.hpp"

using namespace snn;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr, code)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((code), std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

enum AttrType { T_F32 = 0, T_U32 = 1, T_I32 = 2 };
enum AttrStore { S_PLAIN = 0, S_PLAIN_K = 1, S_XPLANE = 2, S_XPLANE_K = 3 };

struct Attr {
    AttrType type;
    AttrStore store;
    void *base;        // S_PLAIN / S_PLAIN_K: device array; planes: unused
    int plane;         // S_XPLANE / S_XPLANE_K
    uint32_t pad;      // stride between types for S_PLAIN_K
    int dirties;       // 1: invalidates the static per-column counts
};

struct LatticeInfo {
    uint32_t id, rows, cols, first, count, slot;
    bool spike_train;
};

} // namespace

struct snn_network {
    int device = 0;
    hipStream_t stream = nullptr;          // the stream every launch goes to
    hipStream_t own_stream = nullptr;      // created with the handle
    bool external_stream = false;          // snn_set_stream adopted a caller's stream
    int model = 0, nt_kind = 0, rc_kind = 0, st_kind = 0;
    bool finalized = false;
    int electrical = 1, chemical = 0;
    long long clock = 0;

    std::vector<LatticeInfo> lattices;      // neuron lattices, ascending id after finalize
    std::vector<LatticeInfo> st_lattices;   // spike-train lattices
    std::vector<long long> st_clock;        // own clocks of the spike-train lattices
    std::vector<float> stdp_host;           // [n_lattices][5]
    std::vector<uint32_t> plast_host;       // [n_lattices]
    bool any_plasticity = false;
    // reward modulation (RewardModulatedLattice): per-lattice modulator table + per-edge trace, allocated on first use
    bool any_modulation = false;
    std::vector<float> rm_host;            // [n_lattices][RM_STRIDE]
    std::vector<uint32_t> rm_on_host;
    float *rm_dev = nullptr;
    uint32_t *rm_on_dev = nullptr;
    float *trace = nullptr;                // dense: [n_tot][ld]; CSR: [sell_entries]

    uint32_t nn = 0, nc = 0, n_tot = 0, n_pad = 0, c_pad = 0;
    uint32_t q0 = 0, q1 = 0, n_loc = 0, ld = 0, n_chunks = 0;
    XLayout xl{0, 1};

    std::vector<void *> allocs;
    // sparse form (CSR by local postsynaptic row); the arrays are replaced by every snn_set_graph_csr
    bool csr = false;
    uint64_t nnz = 0;
    // device: SELL-64 rows (slice_ptr / pre / w / row_len) + per-CSR-edge slot, local row and the transpose index
    uint32_t *csr_ptr = nullptr, *csr_pre = nullptr, *csr_post = nullptr, *csr_t_ptr = nullptr, *csr_t_edge = nullptr;
    uint32_t *csr_row_len = nullptr, *csr_edge_slot = nullptr;
    float *csr_w = nullptr;
    uint64_t sell_entries = 0;
    std::vector<uint32_t> edge_slot_host;   // CSR edge -> SELL entry (for snn_get_graph_csr)
    float *W = nullptr;
    float *xbuf = nullptr;
    float *part_i = nullptr, *part_t = nullptr;
    uint32_t *n_in = nullptr, *tcount = nullptr;
    bool counts_dirty = true;
    NeuronArrays na{};
    CellArrays ca{};
    uint32_t *lattice_slot = nullptr;
    float *stdp_dev = nullptr;
    uint32_t *plast_dev = nullptr;
    uint32_t *spike_list = nullptr, *spike_count = nullptr;
    long long *st_clock_dev = nullptr;
    long long run_step_offset = 0;
    bool run_active = false;        // a (possibly externally driven) run is open: device clocks are ahead of st_clock
    // fused small-lattice step (k_step_resident): two shadow copies of the exchange buffer + per-tile tickets
    float *shadow[2] = {nullptr, nullptr};
    int shadow_cur = 0;
    bool shadow_valid = false;      // shadow[shadow_cur] == exchange buffer
    int fused_step = 1;             // 0: always take the two-kernel path (SNN_AMD_FUSED_STEP=0)
    bool view_dirty = true;         // spike-train gap-junction values must be refreshed before the next inputs
    bool local_inputs_done = false; // this step's LOCAL chunk partials are already enqueued

    std::map<std::string, Attr> neuron_attrs, cell_attrs;

    // histories
    int want_vhist = 0, want_raster = 0;
    // reduced histories: per-lattice average voltage / EEG value per step, per-neuron spike totals
    int want_avg = 0, want_eeg = 0, want_counts = 0;
    float eeg_ref = 0.007f, eeg_dist = 0.8f, eeg_cond = 251.0f;     // EEGHistory defaults, neuron/mod.rs:246-255
    float *summ_avg = nullptr, *summ_eeg = nullptr;                 // [cap][n_lattices]
    uint32_t *spike_counts = nullptr, *lat_first_dev = nullptr, *lat_count_dev = nullptr;
    uint64_t hist_steps = 0, hist_cap = 0;
    std::vector<std::vector<float>> preset_host;   // PresetSpikeTrain firing times per cell
    float *preset_times_dev = nullptr;
    uint64_t hist_tick = 0;                // steps seen since the record was (re)started
    uint32_t hist_every = 1;               // a row is stored when hist_tick % hist_every == 0
    float *vhist = nullptr, *st_vhist = nullptr;
    unsigned long long *raster = nullptr;

    // profiling of the synaptic-input kernel
    int profile = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_counts;     // 1: the launch closes a pass over the graph, 0: first half of a split pass
    size_t ev_used = 0;
    uint64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace {
inline bool recording(const snn_network *net)
{
    return net->want_vhist || net->want_raster || net->want_avg || net->want_eeg;
}
// does the step being computed store its history rows (strided capture: every hist_every-th step)
inline bool record_now(const snn_network *net)
{
    return recording(net) && net->hist_tick % net->hist_every == 0;
}
} // namespace

namespace {

int dev_alloc(snn_network *net, void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0) bytes = 256;
    HIP_TRY(hipMalloc(out, bytes), SNN_ERR_BUFFER_CREATE);
    net->allocs.push_back(*out);
    return SNN_OK;
}

template <typename T>
int dev_alloc_t(snn_network *net, T **out, size_t count)
{
    return dev_alloc(net, reinterpret_cast<void **>(out), count * sizeof(T));
}

int fill_f32(snn_network *net, float *p, size_t n, float v)
{
    if (n == 0) return SNN_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, net->stream, p, n, v);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}
int fill_u32(snn_network *net, uint32_t *p, size_t n, uint32_t v)
{
    if (n == 0) return SNN_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, net->stream, p, n, v);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

const LatticeInfo *find_lattice(const snn_network *net, uint32_t id)
{
    for (const auto &l : net->lattices) if (l.id == id) return &l;
    for (const auto &l : net->st_lattices) if (l.id == id) return &l;
    return nullptr;
}

void reg(std::map<std::string, Attr> &m, const char *name, AttrType t, AttrStore s, void *base, int plane,
         uint32_t pad, int dirties = 0)
{
    m[name] = Attr{t, s, base, plane, pad, dirties};
}

// Allocate one f32 per-neuron array, fill with `def`, register under `name`.
int neuron_f32(snn_network *net, float **field, const char *name, float def)
{
    int rc = dev_alloc_t(net, field, net->n_pad);
    if (rc) return rc;
    rc = fill_f32(net, *field, net->n_pad, def);
    if (rc) return rc;
    if (name) reg(net->neuron_attrs, name, T_F32, S_PLAIN, *field, 0, 0);
    return SNN_OK;
}
int cell_f32(snn_network *net, float **field, const char *name, float def)
{
    int rc = dev_alloc_t(net, field, net->c_pad);
    if (rc) return rc;
    rc = fill_f32(net, *field, net->c_pad, def);
    if (rc) return rc;
    if (name) reg(net->cell_attrs, name, T_F32, S_PLAIN, *field, 0, 0);
    return SNN_OK;
}
// [3][pad] block with per-type defaults
int typed_f32(snn_network *net, float **field, uint32_t pad, float d0, float d1, float d2)
{
    int rc = dev_alloc_t(net, field, (size_t)K_TYPES * pad);
    if (rc) return rc;
    const float d[3] = {d0, d1, d2};
    for (int k = 0; k < K_TYPES; ++k) {
        rc = fill_f32(net, *field + (size_t)k * pad, pad, d[k]);
        if (rc) return rc;
    }
    return SNN_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

int build_state(snn_network *net)
{
    NeuronArrays &n = net->na;
    CellArrays &c = net->ca;
    const uint32_t np = net->n_pad, cp = net->c_pad;
    auto &A = net->neuron_attrs;
    auto &CA = net->cell_attrs;

    // exchanged planes
    TRY(dev_alloc_t(net, &net->xbuf, (size_t)net->xl.n_shards * NUM_PLANES * net->xl.stride));
    HIP_TRY(hipMemsetAsync(net->xbuf, 0, (size_t)net->xl.n_shards * NUM_PLANES * net->xl.stride * 4, net->stream),
            SNN_ERR_BUFFER_WRITE);
    n.xbuf = net->xbuf;
    n.xl = net->xl;
    n.n_pad = np;
    reg(A, "current_voltage", T_F32, S_XPLANE, nullptr, PLANE_V, 0);
    reg(A, "is_spiking", T_U32, S_XPLANE, nullptr, PLANE_SPIKE, 0);
    reg(A, "neurotransmitters$t", T_F32, S_XPLANE_K, nullptr, PLANE_T0, 0);

    // reference defaults: Izhikevich integrate_and_fire/mod.rs:1198-1220, LIF :149-171,
    // Hodgkin-Huxley hodgkin_huxley/mod.rs:80-98 + ion_channels/mod.rs:23-31, 205-215, 255-264, 299-307
    const bool izh = net->model == SNN_MODEL_IZHIKEVICH, lif = net->model == SNN_MODEL_LIF;
    const bool qif = net->model == SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE, slif = net->model == SNN_MODEL_SIMPLE_LIF;
    const bool alif = net->model == SNN_MODEL_ADAPTIVE_LIF, aelif = net->model == SNN_MODEL_ADAPTIVE_EXP_LIF;
    const bool adp = alif || aelif, lizh = net->model == SNN_MODEL_LEAKY_IZHIKEVICH;
    const float v0 = (lif || qif || slif || adp) ? -75.0f : -65.0f;
    {
        // initial voltage into plane V of every shard slot
        for (uint32_t s = 0; s < net->xl.n_shards; ++s)
            TRY(fill_f32(net, net->xbuf + ((size_t)s * NUM_PLANES + PLANE_V) * net->xl.stride, net->xl.stride, v0));
    }
    TRY(neuron_f32(net, &n.gap_conductance, "gap_conductance", slif ? 10.0f : 7.0f));
    TRY(neuron_f32(net, &n.dt, "dt", net->model == SNN_MODEL_HODGKIN_HUXLEY ? 0.01f : 0.1f));
    TRY(neuron_f32(net, &n.c_m, "c_m", net->model == SNN_MODEL_HODGKIN_HUXLEY ? 1.0f : 100.0f));
    TRY(neuron_f32(net, &n.v_th, "v_th", (izh || lizh) ? 30.0f : ((lif || qif || slif || adp) ? -55.0f : 0.0f)));
    TRY(dev_alloc_t(net, &n.last_firing_time, np));
    HIP_TRY(hipMemsetAsync(n.last_firing_time, 0xFF, (size_t)np * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    reg(A, "last_firing_time", T_I32, S_PLAIN, n.last_firing_time, 0, 0);

    const bool izh_like = izh || lizh, lif_like = lif || adp;
    TRY(neuron_f32(net, &n.w_value, (izh_like || adp) ? "w_value" : nullptr, adp ? 0.0f : 30.0f));
    TRY(neuron_f32(net, &n.a, izh_like ? "a" : nullptr, 0.02f));
    TRY(neuron_f32(net, &n.b, izh_like ? "b" : nullptr, 0.2f));
    TRY(neuron_f32(net, &n.c, izh_like ? "c" : nullptr, -55.0f));
    TRY(neuron_f32(net, &n.d, izh_like ? "d" : nullptr, 8.0f));
    TRY(neuron_f32(net, &n.tau_m, (izh_like || lif_like || qif) ? "tau_m" : nullptr, izh ? 1.0f : (qif ? 100.0f : 10.0f)));

    TRY(neuron_f32(net, &n.v_reset, (lif_like || qif || slif) ? "v_reset" : nullptr, -75.0f));
    TRY(neuron_f32(net, &n.refractory_count, (lif_like || qif) ? "refractory_count" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.tref, (lif_like || qif) ? "tref" : nullptr, 10.0f));
    TRY(neuron_f32(net, &n.leak_constant, lif_like ? "leak_constant" : nullptr, -1.0f));
    TRY(neuron_f32(net, &n.integration_constant, (lif_like || qif) ? "integration_constant" : nullptr, 1.0f));
    // adaptive models, integrate_and_fire/mod.rs:969-996, 1105-1130
    TRY(neuron_f32(net, &n.adp_alpha, adp ? "alpha" : nullptr, 6.0f));
    TRY(neuron_f32(net, &n.adp_beta, adp ? "beta" : nullptr, 10.0f));
    TRY(neuron_f32(net, &n.slope_factor, aelif ? "slope_factor" : nullptr, 1.0f));
    // reference buffer names of the two models with a reference GPU implementation
    // (integrate_and_fire/mod.rs:729-773, 1700-1740)
    TRY(neuron_f32(net, &n.qif_alpha, qif ? "alpha" : nullptr, 1.0f));
    TRY(neuron_f32(net, &n.qif_v_c, qif ? "v_c" : nullptr, -60.0f));
    TRY(neuron_f32(net, &n.slif_g, slif ? "g" : nullptr, -0.1f));
    TRY(neuron_f32(net, &n.slif_e, slif ? "e" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.e_l, (lif_like || lizh) ? "e_l" : nullptr, lizh ? -65.0f : -75.0f));
    TRY(neuron_f32(net, &n.g_l, lif_like ? "g_l" : nullptr, 10.0f));

    const bool hh = net->model == SNN_MODEL_HODGKIN_HUXLEY;
    TRY(neuron_f32(net, &n.m_state, hh ? "na_channel$m$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_state, hh ? "na_channel$h$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_state, hh ? "k_channel$n$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.m_alpha, hh ? "na_channel$m$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.m_beta, hh ? "na_channel$m$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_alpha, hh ? "na_channel$h$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_beta, hh ? "na_channel$h$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_alpha, hh ? "k_channel$n$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_beta, hh ? "k_channel$n$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.g_na, hh ? "na_channel$g_na" : nullptr, 120.0f));
    TRY(neuron_f32(net, &n.e_na, hh ? "na_channel$e_na" : nullptr, 50.0f));
    TRY(neuron_f32(net, &n.g_k, hh ? "k_channel$g_k" : nullptr, 36.0f));
    TRY(neuron_f32(net, &n.e_k, hh ? "k_channel$e_k" : nullptr, -77.0f));
    TRY(neuron_f32(net, &n.g_k_leak, hh ? "k_leak_channel$g_k_leak" : nullptr, 0.3f));
    TRY(neuron_f32(net, &n.e_k_leak, hh ? "k_leak_channel$e_k_leak" : nullptr, -55.0f));
    TRY(neuron_f32(net, &n.na_current, hh ? "na_channel$current" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.k_current, hh ? "k_channel$current" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.k_leak_current, hh ? "k_leak_channel$current" : nullptr, 0.0f));
    TRY(dev_alloc_t(net, &n.was_increasing, np));
    TRY(fill_u32(net, n.was_increasing, np, 0));
    if (hh) reg(A, "was_increasing", T_U32, S_PLAIN, n.was_increasing, 0, 0);

    // neurotransmitters (iterate_and_spike/mod.rs:136-145, 174-182) -- absent by default (flags 0)
    TRY(typed_f32(net, &n.nt_t_max, np, 1.0f, 1.0f, 1.0f));
    // clearance_constant of the Approximate kinetics / decay_constant of ExponentialDecay (:336-343) share storage
    const float nt_c = net->nt_kind == SNN_NT_EXPONENTIAL_DECAY ? 2.0f : 0.01f;
    TRY(typed_f32(net, &n.nt_clearance, np, nt_c, nt_c, nt_c));
    TRY(typed_f32(net, &n.nt_v_p, np, 2.0f, 2.0f, 2.0f));
    TRY(typed_f32(net, &n.nt_k_p, np, 5.0f, 5.0f, 5.0f));
    TRY(dev_alloc_t(net, &n.nt_flags, (size_t)K_TYPES * np));
    TRY(fill_u32(net, n.nt_flags, (size_t)K_TYPES * np, 0));
    reg(A, "neurotransmitters$t_max", T_F32, S_PLAIN_K, n.nt_t_max, 0, np);
    reg(A, "neurotransmitters$clearance_constant", T_F32, S_PLAIN_K, n.nt_clearance, 0, np);
    reg(A, "neurotransmitters$decay_constant", T_F32, S_PLAIN_K, n.nt_clearance, 0, np);
    reg(A, "neurotransmitters$v_p", T_F32, S_PLAIN_K, n.nt_v_p, 0, np);
    reg(A, "neurotransmitters$k_p", T_F32, S_PLAIN_K, n.nt_k_p, 0, np);
    reg(A, "neurotransmitters$flags", T_U32, S_PLAIN_K, n.nt_flags, 0, np, 1);

    // receptors (iterate_and_spike/mod.rs:1085-1094, 1115-1125, 1148-1157, 417-425)
    TRY(typed_f32(net, &n.rc_g, np, 1.0f, 0.6f, 1.2f));
    TRY(typed_f32(net, &n.rc_e, np, 0.0f, 0.0f, -80.0f));
    TRY(typed_f32(net, &n.rc_mg, np, 0.0f, 0.3f, 0.0f));
    TRY(typed_f32(net, &n.rc_r, np, 0.0f, 0.0f, 0.0f));
    TRY(typed_f32(net, &n.rc_alpha, np, 1.0f, 1.0f, 1.0f));
    // ExponentialDecayReceptor (:501-533): r_max lives in the alpha array, decay_constant in the beta array
    const float rc_b = net->rc_kind == SNN_RC_EXPONENTIAL_DECAY ? 2.0f : 1.0f;
    TRY(typed_f32(net, &n.rc_beta, np, rc_b, rc_b, rc_b));
    TRY(typed_f32(net, &n.rc_current, np, 0.0f, 0.0f, 0.0f));
    TRY(dev_alloc_t(net, &n.rc_flags, (size_t)K_TYPES * np));
    TRY(fill_u32(net, n.rc_flags, (size_t)K_TYPES * np, 0));
    reg(A, "receptors$flags", T_U32, S_PLAIN_K, n.rc_flags, 0, np);
    static const char *TN[3] = {"AMPA", "NMDA", "GABA"};
    for (int k = 0; k < K_TYPES; ++k) {
        const std::string p = std::string("receptors$") + TN[k];
        reg(A, (p + "_g").c_str(), T_F32, S_PLAIN, n.rc_g + (size_t)k * np, 0, 0);
        reg(A, (p + "_e").c_str(), T_F32, S_PLAIN, n.rc_e + (size_t)k * np, 0, 0);
        reg(A, (p + "_current").c_str(), T_F32, S_PLAIN, n.rc_current + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$r").c_str(), T_F32, S_PLAIN, n.rc_r + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$alpha").c_str(), T_F32, S_PLAIN, n.rc_alpha + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$beta").c_str(), T_F32, S_PLAIN, n.rc_beta + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$r_max").c_str(), T_F32, S_PLAIN, n.rc_alpha + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$decay_constant").c_str(), T_F32, S_PLAIN, n.rc_beta + (size_t)k * np, 0, 0);
    }
    reg(A, "receptors$NMDA_mg", T_F32, S_PLAIN, n.rc_mg + (size_t)1 * np, 0, 0);

    // lattice slot per neuron + plasticity tables
    TRY(dev_alloc_t(net, &net->lattice_slot, np));
    TRY(fill_u32(net, net->lattice_slot, np, 0));
    for (const auto &l : net->lattices) TRY(fill_u32(net, net->lattice_slot + l.first, l.count, l.slot));
    const size_t nl = std::max<size_t>(1, net->lattices.size());
    {
        std::vector<uint32_t> lf(nl, 0), lc(nl, 0);
        for (const auto &l : net->lattices) { lf[l.slot] = l.first; lc[l.slot] = l.count; }
        TRY(dev_alloc_t(net, &net->lat_first_dev, nl));
        TRY(dev_alloc_t(net, &net->lat_count_dev, nl));
        HIP_TRY(hipMemcpy(net->lat_first_dev, lf.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipMemcpy(net->lat_count_dev, lc.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        TRY(dev_alloc_t(net, &net->spike_counts, np));
        TRY(fill_u32(net, net->spike_counts, np, 0));
    }
    net->stdp_host.assign(nl * 5, 0.0f);
    net->plast_host.assign(nl, 0);
    for (size_t l = 0; l < nl; ++l) {   // plasticity/mod.rs:29-39
        float *s = &net->stdp_host[l * 5];
        s[0] = 2.0f; s[1] = 2.0f; s[2] = 4.5f; s[3] = 4.5f; s[4] = 0.1f;
    }
    TRY(dev_alloc_t(net, &net->stdp_dev, nl * 5));
    TRY(dev_alloc_t(net, &net->plast_dev, nl));
    HIP_TRY(hipMemcpyAsync(net->stdp_dev, net->stdp_host.data(), nl * 5 * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpyAsync(net->plast_dev, net->plast_host.data(), nl * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    // RewardModulatedSTDP defaults, plasticity/mod.rs:176-189
    net->rm_host.assign(nl * RM_STRIDE, 0.0f);
    net->rm_on_host.assign(nl, 0);
    for (size_t l = 0; l < nl; ++l) {
        float *m = &net->rm_host[l * RM_STRIDE];
        m[0] = 0.0f; m[1] = 20.0f; m[2] = 0.0001f; m[3] = 2.0f; m[4] = 2.0f; m[5] = 4.5f; m[6] = 4.5f; m[7] = 0.1f;
    }
    TRY(dev_alloc_t(net, &net->rm_dev, nl * RM_STRIDE));
    TRY(dev_alloc_t(net, &net->rm_on_dev, nl));
    HIP_TRY(hipMemcpyAsync(net->rm_dev, net->rm_host.data(), nl * RM_STRIDE * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpyAsync(net->rm_on_dev, net->rm_on_host.data(), nl * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    TRY(dev_alloc_t(net, &net->spike_list, np));
    TRY(dev_alloc_t(net, &net->spike_count, 1));

    // spike-train cells (spike_train/mod.rs:299-313, 998-1013, 50-56)
    c.c_pad = cp;
    TRY(cell_f32(net, &c.current_voltage, "current_voltage", 0.0f));
    TRY(cell_f32(net, &c.v_th, "v_th", 30.0f));
    TRY(cell_f32(net, &c.v_resting, "v_resting", 0.0f));
    TRY(cell_f32(net, &c.dt, "dt", 0.1f));
    TRY(cell_f32(net, &c.k, "neural_refractoriness$k", 10000.0f));
    TRY(cell_f32(net, &c.chance_of_firing, net->st_kind == SNN_ST_POISSON ? "chance_of_firing" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.rate, net->st_kind == SNN_ST_RATE ? "rate" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.step, net->st_kind == SNN_ST_RATE ? "step" : (net->st_kind == SNN_ST_PRESET ? "internal_clock" : nullptr), 0.0f));
    TRY(dev_alloc_t(net, &c.counter, cp));
    TRY(fill_u32(net, c.counter, cp, 0));
    if (net->st_kind == SNN_ST_PRESET) reg(CA, "counter", T_U32, S_PLAIN, c.counter, 0, 0);
    {
        // no firing times until snn_set_firing_times: every cell's list is empty
        uint32_t *ptr = nullptr;
        TRY(dev_alloc_t(net, &ptr, (size_t)cp + 1));
        TRY(fill_u32(net, ptr, (size_t)cp + 1, 0));
        c.preset_ptr = ptr;
        c.preset_times = nullptr;
        net->preset_host.assign(net->nc, {});
    }
    TRY(cell_f32(net, &c.presyn_value, nullptr, 0.0f));
    TRY(dev_alloc_t(net, &c.seed, cp));
    if (cp) {
        hipLaunchKernelGGL(k_iota_u32, dim3((cp + 255) / 256), dim3(256), 0, net->stream, c.seed, (size_t)cp, 1u);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    if (net->st_kind == SNN_ST_POISSON) reg(CA, "seed", T_U32, S_PLAIN, c.seed, 0, 0);
    TRY(dev_alloc_t(net, &c.is_spiking, cp));
    TRY(fill_u32(net, c.is_spiking, cp, 0));
    reg(CA, "is_spiking", T_U32, S_PLAIN, c.is_spiking, 0, 0);
    TRY(dev_alloc_t(net, &c.last_firing_time, cp));
    HIP_TRY(hipMemsetAsync(c.last_firing_time, 0xFF, (size_t)std::max<uint32_t>(cp, 1) * 4, net->stream),
            SNN_ERR_BUFFER_WRITE);
    reg(CA, "last_firing_time", T_I32, S_PLAIN, c.last_firing_time, 0, 0);
    TRY(typed_f32(net, &c.nt_t, cp, 0.0f, 0.0f, 0.0f));
    TRY(typed_f32(net, &c.nt_t_max, cp, 1.0f, 1.0f, 1.0f));
    TRY(typed_f32(net, &c.nt_clearance, cp, nt_c, nt_c, nt_c));
    TRY(typed_f32(net, &c.nt_v_p, cp, 2.0f, 2.0f, 2.0f));
    TRY(typed_f32(net, &c.nt_k_p, cp, 5.0f, 5.0f, 5.0f));
    TRY(dev_alloc_t(net, &c.nt_flags, (size_t)K_TYPES * cp));
    TRY(fill_u32(net, c.nt_flags, (size_t)K_TYPES * cp, 0));
    reg(CA, "neurotransmitters$t", T_F32, S_PLAIN_K, c.nt_t, 0, cp);
    reg(CA, "neurotransmitters$t_max", T_F32, S_PLAIN_K, c.nt_t_max, 0, cp);
    reg(CA, "neurotransmitters$clearance_constant", T_F32, S_PLAIN_K, c.nt_clearance, 0, cp);
    reg(CA, "neurotransmitters$decay_constant", T_F32, S_PLAIN_K, c.nt_clearance, 0, cp);
    reg(CA, "neurotransmitters$v_p", T_F32, S_PLAIN_K, c.nt_v_p, 0, cp);
    reg(CA, "neurotransmitters$k_p", T_F32, S_PLAIN_K, c.nt_k_p, 0, cp);
    reg(CA, "neurotransmitters$flags", T_U32, S_PLAIN_K, c.nt_flags, 0, cp, 1);
    TRY(dev_alloc_t(net, &c.lattice_slot, cp));
    TRY(fill_u32(net, c.lattice_slot, cp, 0));
    for (const auto &l : net->st_lattices)
        TRY(fill_u32(net, c.lattice_slot + (l.first - net->nn), l.count, l.slot));
    net->st_clock.assign(std::max<size_t>(1, net->st_lattices.size()), 0);
    TRY(dev_alloc_t(net, &net->st_clock_dev, net->st_clock.size()));

    // graph + partials + counts
    if (net->csr) net->n_chunks = 1;     // the CSR kernel writes the finished two-level sum
    TRY(dev_alloc_t(net, &net->W, net->csr ? 0 : (size_t)net->n_tot * net->ld));
    TRY(dev_alloc_t(net, &net->part_i, (size_t)net->n_chunks * net->ld));
    TRY(dev_alloc_t(net, &net->part_t, (size_t)K_TYPES * net->n_chunks * net->ld));
    TRY(dev_alloc_t(net, &net->n_in, net->ld));
    TRY(dev_alloc_t(net, &net->tcount, (size_t)K_TYPES * net->ld));
    net->counts_dirty = true;
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}

int end_run(snn_network *net);

// ---- attribute transfer ------------------------------------------------------------------------

// copy `count` 32-bit words between host and plane `plane` for global indices [first, first+count)
int xplane_copy(snn_network *net, int plane, uint32_t first, uint32_t count, void *host, bool to_device)
{
    uint32_t done = 0;
    while (done < count) {
        const uint32_t g = first + done;
        const uint32_t shard = g / net->xl.stride;
        const uint32_t in_shard = g - shard * net->xl.stride;
        const uint32_t seg = std::min(count - done, net->xl.stride - in_shard);
        float *dev = net->xbuf + net->xl.at(g, plane);
        char *h = static_cast<char *>(host) + (size_t)done * 4;
        if (to_device) HIP_TRY(hipMemcpy(dev, h, (size_t)seg * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        else HIP_TRY(hipMemcpy(h, dev, (size_t)seg * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
        done += seg;
    }
    return SNN_OK;
}

int attr_io(snn_network *net, uint32_t id, const char *name, AttrType type, void *host, size_t count, bool set)
{
    if (!net || !name || (!host && count)) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id " + std::to_string(id));
    auto &table = l->spike_train ? net->cell_attrs : net->neuron_attrs;
    auto it = table.find(name);
    if (it == table.end()) return fail(SNN_ERR_BAD_ATTR, std::string("unknown attribute '") + name + "'");
    const Attr &a = it->second;
    if (a.type != type) return fail(SNN_ERR_BAD_ATTR, std::string("attribute '") + name + "' has another scalar type");
    const bool typed = (a.store == S_PLAIN_K || a.store == S_XPLANE_K);
    const size_t expect = (size_t)l->count * (typed ? K_TYPES : 1);
    if (count != expect)
        return fail(SNN_ERR_DIM_MISMATCH, std::string("attribute '") + name + "': expected " +
                                              std::to_string(expect) + " values, got " + std::to_string(count));
    if (l->count == 0) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (set && l->spike_train) net->view_dirty = true;
    if (set) net->shadow_valid = false;
    const uint32_t first = l->spike_train ? l->first - net->nn : l->first;   // index inside its own arrays

    if (!typed) {
        if (a.store == S_PLAIN) {
            char *dev = static_cast<char *>(a.base) + (size_t)first * 4;
            if (set) HIP_TRY(hipMemcpy(dev, host, count * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
            else HIP_TRY(hipMemcpy(host, dev, count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
        } else {
            TRY(xplane_copy(net, a.plane, first, l->count, host, set));
        }
    } else {
        // host layout [cell*3 + k] (gpu_lattices/mod.rs:117-127) <-> device type-major planes
        std::vector<uint32_t> tmp(l->count);
        uint32_t *h = static_cast<uint32_t *>(host);
        for (int k = 0; k < K_TYPES; ++k) {
            if (set) for (uint32_t i = 0; i < l->count; ++i) tmp[i] = h[(size_t)i * K_TYPES + k];
            if (a.store == S_PLAIN_K) {
                char *dev = static_cast<char *>(a.base) + ((size_t)k * a.pad + first) * 4;
                if (set) HIP_TRY(hipMemcpy(dev, tmp.data(), (size_t)l->count * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
                else HIP_TRY(hipMemcpy(tmp.data(), dev, (size_t)l->count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
            } else {
                TRY(xplane_copy(net, a.plane + k, first, l->count, tmp.data(), set));
            }
            if (!set) for (uint32_t i = 0; i < l->count; ++i) h[(size_t)i * K_TYPES + k] = tmp[i];
        }
    }
    if (set && a.dirties) net->counts_dirty = true;
    return SNN_OK;
}

// ---- per-step launches -------------------------------------------------------------------------

SellGraph csr_graph(const snn_network *net)
{
    SellGraph g{};
    g.slice_ptr = net->csr_ptr; g.pre = net->csr_pre; g.w = net->csr_w; g.row_len = net->csr_row_len;
    g.edge_slot = net->csr_edge_slot; g.edge_post = net->csr_post;
    g.t_ptr = net->csr_t_ptr; g.t_edge = net->csr_t_edge;
    g.n_loc = net->n_loc; g.n_slices = (net->n_loc + 63) / 64;
    return g;
}

int ensure_counts(snn_network *net)
{
    if (!net->counts_dirty || net->n_loc == 0) { net->counts_dirty = false; return SNN_OK; }
    HIP_TRY(hipMemsetAsync(net->n_in, 0, (size_t)net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->tcount, 0, (size_t)K_TYPES * net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    if (net->csr) {
        if (net->csr_ptr) {
            CsrCountArgs a{};
            a.g = csr_graph(net);
            a.n_neurons = net->nn; a.ld = net->ld;
            a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
            a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
            a.n_in = net->n_in; a.tcount = net->tcount;
            hipLaunchKernelGGL(k_csr_count, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, a);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        }
    } else if (net->n_tot) {
        CountArgs a{};
        a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.n_neurons = net->nn; a.n_tot = net->n_tot;
        a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
        a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
        a.n_in = net->n_in; a.tcount = net->tcount;
        a.rows_per_block = 256;
        dim3 grid((net->n_loc + 255) / 256, (net->n_tot + 255) / 256);
        hipLaunchKernelGGL(k_graph_count, grid, dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->counts_dirty = false;
    return SNN_OK;
}

int launch_spike_trains(snn_network *net, int iterate, long long step_offset, long long view_clock)
{
    if (net->nc == 0) return SNN_OK;
    SpikeTrainArgs a{};
    a.c = net->ca; a.n_cells = net->nc; a.st_kind = net->st_kind; a.nt_kind = net->nt_kind;
    a.iterate = iterate; a.lattice_clock = net->st_clock_dev; a.step_offset = step_offset;
    a.view_clock = view_clock;
    a.vhist_row = (iterate && record_now(net) && net->want_vhist && net->st_vhist) ? net->st_vhist + (size_t)net->hist_steps * net->c_pad : nullptr;
    hipLaunchKernelGGL(k_spike_trains, dim3((net->nc + 255) / 256), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

enum InputsPart { INPUTS_ALL = 0, INPUTS_LOCAL = 1, INPUTS_REMOTE = 2 };

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
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = net->xbuf; a.xl = net->xl; a.gap_conductance = net->na.gap_conductance;
    a.st_value = net->ca.presyn_value; a.st_last_firing_time = net->ca.last_firing_time;
    a.st_nt_t = net->ca.nt_t; a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
    a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_chunks = net->n_chunks;
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
        net->ev_counts[net->ev_used] = (part == INPUTS_LOCAL) ? 0 : 1;   // LOCAL + REMOTE = one pass over W
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
    const int shape = resident ? 0 : (waves4 < 8192 ? 2 : 1);     // 8192 = 256 CUs x 32 wave slots
#define SNN_LAUNCH_SHAPE(E, C, SH)                                                                        \
    hipLaunchKernelGGL((k_inputs_dense<E, C, SH>),                                                       \
                       dim3((net->n_loc + InputsShape<SH>::TILE - 1) / InputsShape<SH>::TILE, grid_chunks), \
                       dim3(InputsShape<SH>::THREADS), 0, net->stream, a)
#define SNN_LAUNCH_INPUTS(E, C)                                                                          \
    do {                                                                                                 \
        if (shape == 1) SNN_LAUNCH_SHAPE(E, C, 1);                                                       \
        else if (shape == 2) SNN_LAUNCH_SHAPE(E, C, 2);                                                  \
        else SNN_LAUNCH_SHAPE(E, C, 0);                                                                  \
    } while (0)
    if (net->electrical && net->chemical) SNN_LAUNCH_INPUTS(true, true);
    else if (net->electrical) SNN_LAUNCH_INPUTS(true, false);
    else SNN_LAUNCH_INPUTS(false, true);
#undef SNN_LAUNCH_INPUTS
#undef SNN_LAUNCH_SHAPE
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    return SNN_OK;
}

int launch_update(snn_network *net)
{
    if (net->n_loc == 0) return SNN_OK;
    UpdateArgs a{};
    a.n = net->na;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_in = net->n_in; a.tcount = net->tcount;
    a.ld = net->ld; a.n_chunks = net->n_tot ? net->n_chunks : 0; a.q0 = net->q0; a.n_loc = net->n_loc;
    a.clock = net->clock;
    a.electrical = net->electrical; a.chemical = net->chemical; a.nt_kind = net->nt_kind; a.rc_kind = net->rc_kind;
    a.vhist_row = (record_now(net) && net->want_vhist && net->vhist) ? net->vhist + (size_t)net->hist_steps * net->n_pad : nullptr;
    a.spike_row = (record_now(net) && net->want_raster && net->raster) ? net->raster + (size_t)net->hist_steps * (net->n_pad / 64) : nullptr;
    a.spike_counts = net->want_counts ? net->spike_counts : nullptr;
    a.xout = net->xbuf; a.xout2 = nullptr;
    net->shadow_valid = false;            // the exchange buffer moves on without the shadows
    dim3 grid((net->ld + 255) / 256);
    switch (net->model) {
    case SNN_MODEL_LIF: hipLaunchKernelGGL((k_update<1>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_HODGKIN_HUXLEY: hipLaunchKernelGGL((k_update<2>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE: hipLaunchKernelGGL((k_update<3>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_SIMPLE_LIF: hipLaunchKernelGGL((k_update<4>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_ADAPTIVE_LIF: hipLaunchKernelGGL((k_update<5>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_ADAPTIVE_EXP_LIF: hipLaunchKernelGGL((k_update<6>), grid, dim3(256), 0, net->stream, a); break;
    case SNN_MODEL_LEAKY_IZHIKEVICH: hipLaunchKernelGGL((k_update<7>), grid, dim3(256), 0, net->stream, a); break;
    default: hipLaunchKernelGGL((k_update<0>), grid, dim3(256), 0, net->stream, a); break;
    }
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

int launch_plasticity(snn_network *net)
{
    if (!net->any_plasticity || net->nn == 0) return SNN_OK;
    StdpArgs a{};
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = net->xbuf; a.xl = net->xl;
    a.last_firing_time = net->na.last_firing_time; a.st_last_firing_time = net->ca.last_firing_time;
    a.lattice_slot = net->lattice_slot; a.stdp = net->stdp_dev; a.do_plasticity = net->plast_dev;
    a.spike_list = net->spike_list; a.spike_count = net->spike_count;
    HIP_TRY(hipMemsetAsync(net->spike_count, 0, 4, net->stream), SNN_ERR_BUFFER_WRITE);
    hipLaunchKernelGGL(k_spike_compact, dim3((net->nn + 255) / 256), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (net->n_loc == 0) return SNN_OK;
    if (net->csr) {
        if (!net->csr_ptr) return SNN_OK;
        CsrStdpArgs ca{};
        ca.g = csr_graph(net);
        ca.s = a;
        hipLaunchKernelGGL(k_stdp_csr_in, dim3(1024), dim3(64), 0, net->stream, ca);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        hipLaunchKernelGGL(k_stdp_csr_out, dim3(1024), dim3(64), 0, net->stream, ca);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    const unsigned sy = 64;   // spiking neurons processed concurrently; the rest grid-strides
    hipLaunchKernelGGL(k_stdp_columns, dim3((net->n_tot + 255) / 256, sy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    hipLaunchKernelGGL(k_stdp_rows, dim3((net->n_loc + 255) / 256, sy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// RewardModulatedLattice::update_weights_from_neurons for every modulated lattice (deferred form)
int launch_reward_modulation(snn_network *net)
{
    if (!net->any_modulation || net->nn == 0 || net->n_loc == 0 || !net->trace) return SNN_OK;
    if (net->csr) {
        if (!net->csr_ptr) return SNN_OK;
        CsrRewardArgs a{};
        a.g = csr_graph(net); a.c = net->trace; a.q0 = net->q0; a.n_neurons = net->nn;
        a.last_firing_time = net->na.last_firing_time; a.lattice_slot = net->lattice_slot;
        a.rm = net->rm_dev; a.rm_on = net->rm_on_dev;
        hipLaunchKernelGGL(k_rstdp_csr, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    RewardArgs a{};
    a.W = net->W; a.C = net->trace; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.n_neurons = net->nn;
    a.last_firing_time = net->na.last_firing_time; a.lattice_slot = net->lattice_slot;
    a.rm = net->rm_dev; a.rm_on = net->rm_on_dev;
    const unsigned gx = (net->n_loc + 1023) / 1024;
    const unsigned gy = std::max(1u, std::min<unsigned>(net->nn, 8192u / gx));     // ~8192 workgroups in flight
    hipLaunchKernelGGL(k_rstdp_dense, dim3(gx, gy), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// The synapse matrix is the one allocation whose HBM placement matters: on MI355X two 17 GB allocations of one
// process can differ by 5-6 % in the sustained rate of the input pass (stable per allocation, different from
// process to process).  For matrices >= 1 GiB a second candidate is allocated while the first is held, the real
// kernel is timed on both (one warm + one timed pass each, once per handle) and the faster allocation is kept.
int time_input_pass(snn_network *net, float *ms)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(SNN_ERR_QUEUE, "hipEventCreate failed");
    int rc = launch_inputs(net);
    if (rc == SNN_OK && hipEventRecord(e0, net->stream) != hipSuccess) rc = fail(SNN_ERR_QUEUE, "hipEventRecord failed");
    if (rc == SNN_OK) rc = launch_inputs(net);
    if (rc == SNN_OK && (hipEventRecord(e1, net->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                         hipEventElapsedTime(ms, e0, e1) != hipSuccess))
        rc = fail(SNN_ERR_WAIT, "placement timing failed");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int choose_matrix_placement(snn_network *net)
{
    const size_t count = net->csr ? 0 : (size_t)net->n_tot * net->ld;
    const size_t bytes = count * sizeof(float);
    if (bytes >= ((size_t)1 << 30) && net->n_loc) {
        const int prof = net->profile;
        net->profile = 0;
        float best_ms = 0.0f;
        int rc = time_input_pass(net, &best_ms);
        // up to four more candidates; every loser stays allocated until the end so that each new candidate is
        // forced into a different HBM region (a freed block would simply be handed out again)
        std::vector<void *> losers;
        for (int cand = 0; cand < 4 && rc == SNN_OK; ++cand) {
            size_t free_b = 0, total_b = 0;
            void *b = nullptr;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (bytes >> 2) ||
                hipMalloc(&b, bytes) != hipSuccess)
                break;
            float *a = net->W;
            float ms_b = 0.0f;
            net->W = static_cast<float *>(b);
            rc = time_input_pass(net, &ms_b);
            if (getenv("SNN_DEBUG_PLACEMENT"))
                fprintf(stderr, "[snn] matrix placement: held %p %.3f ms, candidate %p %.3f ms\n", (void *)a, best_ms, b, ms_b);
            if (rc == SNN_OK && ms_b < best_ms * 0.99f) {       // the candidate wins
                for (auto &p : net->allocs) if (p == a) p = b;
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
    return net->fused_step && !net->csr && net->xl.n_shards == 1 && net->n_loc && net->n_tot && !net->local_inputs_done &&
           net->n_chunks <= RESIDENT_MAX_CHUNKS && (size_t)net->n_tot * net->ld * 4 <= ((size_t)64 << 20);
}

int launch_step_resident(snn_network *net)
{
    const size_t xelems = (size_t)net->xl.n_shards * NUM_PLANES * net->xl.stride;
    if (!net->shadow[0]) {
        TRY(dev_alloc_t(net, &net->shadow[0], xelems));
        TRY(dev_alloc_t(net, &net->shadow[1], xelems));
        net->shadow_valid = false;
    }
    if (!net->shadow_valid) {
        // both shadows: entries the step never rewrites (absent transmitter types, padding) must agree everywhere
        for (int i = 0; i < 2; ++i)
            HIP_TRY(hipMemcpyAsync(net->shadow[i], net->xbuf, xelems * 4, hipMemcpyDeviceToDevice, net->stream),
                    SNN_ERR_BUFFER_WRITE);
        net->shadow_valid = true;
    }
    float *cur = net->shadow[net->shadow_cur], *next = net->shadow[net->shadow_cur ^ 1];
    ResidentArgs r{};
    InputsArgs &a = r.in;
    a.chunk_first = 0; a.hole_begin = net->n_chunks; a.hole_count = 0;
    a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.q0 = net->q0; a.n_neurons = net->nn; a.n_tot = net->n_tot;
    a.xbuf = cur; a.xl = net->xl; a.gap_conductance = net->na.gap_conductance;
    a.st_value = net->ca.presyn_value; a.st_last_firing_time = net->ca.last_firing_time;
    a.st_nt_t = net->ca.nt_t; a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
    a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
    a.part_i = net->part_i; a.part_t = net->part_t; a.n_chunks = net->n_chunks;
    UpdateArgs &u = r.up;
    u.n = net->na;
    u.n.xbuf = cur;
    u.part_i = net->part_i; u.part_t = net->part_t; u.n_in = net->n_in; u.tcount = net->tcount;
    u.ld = net->ld; u.n_chunks = net->n_chunks; u.q0 = net->q0; u.n_loc = net->n_loc;
    u.clock = net->clock;
    u.electrical = net->electrical; u.chemical = net->chemical; u.nt_kind = net->nt_kind; u.rc_kind = net->rc_kind;
    u.vhist_row = (record_now(net) && net->want_vhist && net->vhist) ? net->vhist + (size_t)net->hist_steps * net->n_pad : nullptr;
    u.spike_row = (record_now(net) && net->want_raster && net->raster) ? net->raster + (size_t)net->hist_steps * (net->n_pad / 64) : nullptr;
    u.spike_counts = net->want_counts ? net->spike_counts : nullptr;
    u.xout = net->xbuf; u.xout2 = next;
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
        net->ev_counts[net->ev_used] = 1;
        ++net->ev_used;
        HIP_TRY(hipEventRecord(e0, net->stream), SNN_ERR_QUEUE);
    }
    const dim3 grid((net->n_loc + 63) / 64), block(64 * net->n_chunks);
#define SNN_RESIDENT(M)                                                                                              \
    do {                                                                                                             \
        if (net->electrical && net->chemical) hipLaunchKernelGGL((k_step_resident<M, true, true>), grid, block, 0, net->stream, r);  \
        else if (net->electrical) hipLaunchKernelGGL((k_step_resident<M, true, false>), grid, block, 0, net->stream, r);             \
        else hipLaunchKernelGGL((k_step_resident<M, false, true>), grid, block, 0, net->stream, r);                                  \
    } while (0)
    switch (net->model) {
    case 1: SNN_RESIDENT(1); break;
    case 2: SNN_RESIDENT(2); break;
    case 3: SNN_RESIDENT(3); break;
    case 4: SNN_RESIDENT(4); break;
    case 5: SNN_RESIDENT(5); break;
    case 6: SNN_RESIDENT(6); break;
    case 7: SNN_RESIDENT(7); break;
    default: SNN_RESIDENT(0); break;
    }
#undef SNN_RESIDENT
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    if (net->profile) HIP_TRY(hipEventRecord(e1, net->stream), SNN_ERR_QUEUE);
    net->shadow_cur ^= 1;
    return SNN_OK;
}

// first half of a step: inputs from S(t) and the local neurons' update
int step_begin(snn_network *net)
{
    if (fused_step_applies(net)) return launch_step_resident(net);
    TRY(launch_inputs(net, net->local_inputs_done ? INPUTS_REMOTE : INPUTS_ALL));
    net->local_inputs_done = false;
    TRY(launch_update(net));
    return SNN_OK;
}

// second half: remote last_firing_time, plasticity, histories, clock, spike trains (steps 3-6)
int step_end(snn_network *net)
{
    if (net->xl.n_shards > 1 && net->nn) {
        hipLaunchKernelGGL(k_stamp_remote, dim3((net->nn + 255) / 256), dim3(256), 0, net->stream,
                           net->xbuf, net->xl, net->na.last_firing_time, net->nn, net->q0, net->n_loc, net->clock);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    TRY(launch_plasticity(net));
    TRY(launch_reward_modulation(net));
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
    TRY(launch_spike_trains(net, 1, net->run_step_offset, net->clock));
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
        (!net->want_avg || net->summ_avg) && (!net->want_eeg || net->summ_eeg))
        return SNN_OK;
    const uint64_t cap = std::max<uint64_t>(need, net->hist_cap + net->hist_cap / 2);   // geometric: O(T) copies overall
    auto regrow = [&](void **buf, size_t row_bytes, bool wanted) -> int {
        if (!wanted || row_bytes == 0) return SNN_OK;
        void *nb = nullptr;
        HIP_TRY(hipMalloc(&nb, std::max<size_t>(256, cap * row_bytes)), SNN_ERR_BUFFER_CREATE);
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
    net->hist_cap = cap;
    return SNN_OK;
}

// Opens a run (snn_run, or a sequence of externally driven steps): static counts, history capacity, the
// spike-train lattices' clocks on the device and -- only when cell state or the clock changed behind the
// stepper's back -- the spike-train gap-junction values for the current clock.
int begin_run(snn_network *net, uint64_t iterations)
{
    TRY(ensure_counts(net));
    TRY(grow_history(net, iterations));
    if (net->run_active) return SNN_OK;
    if (net->nc) {
        // pageable source: the copy is staged before the call returns, so the host vector may change afterwards
        HIP_TRY(hipMemcpyAsync(net->st_clock_dev, net->st_clock.data(), net->st_clock.size() * sizeof(long long),
                               hipMemcpyHostToDevice, net->stream), SNN_ERR_BUFFER_WRITE);
        if (net->view_dirty) TRY(launch_spike_trains(net, 0, 0, net->clock));
    }
    net->view_dirty = false;
    net->run_step_offset = 0;
    net->run_active = true;
    return SNN_OK;
}

// Closes the open run: waits for the stream and folds the steps done into the host-side lattice clocks.
int end_run(snn_network *net)
{
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
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dw), (size_t)hop * host_ld * 4), SNN_ERR_BUFFER_CREATE);
    if (hipMalloc(reinterpret_cast<void **>(&dc), (size_t)hop * host_ld * 4) != hipSuccess) {
        (void)hipFree(dw);
        return fail(SNN_ERR_BUFFER_CREATE, "staging allocation failed");
    }
    int rc = SNN_OK;
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
                               net->W, net->ld, net->n_loc, net->q0, pre_begin + r, rows, dw, dc, host_ld);
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
    (void)hipFree(dw);
    (void)hipFree(dc);
    if (set) net->counts_dirty = true;
    return rc;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int snn_abi_version(void) { return SNN_ABI_VERSION; }
const char *snn_last_error(void) { return g_last_error.c_str(); }

int snn_network_create(int device, int neuron_model, int nt_kinetics, int receptor_kinetics,
                       int spike_train_model, snn_network_t **out)
{
    if (!out) return fail(SNN_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (neuron_model < 0 || neuron_model > 7 || nt_kinetics < 0 || nt_kinetics > 3 || receptor_kinetics < 0 ||
        receptor_kinetics > 2 || spike_train_model < 0 || spike_train_model > 3)
        return fail(SNN_ERR_BAD_ARG, "unknown model / kinetics selector");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SNN_ERR_GET_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(SNN_ERR_GET_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    snn_network *net = new snn_network();
    net->device = device;
    if (const char *e = getenv("SNN_AMD_FUSED_STEP")) net->fused_step = (e[0] != '0');
    net->model = neuron_model; net->nt_kind = nt_kinetics; net->rc_kind = receptor_kinetics;
    net->st_kind = spike_train_model;
    if (hipStreamCreateWithFlags(&net->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete net;
        return fail(SNN_ERR_QUEUE, "hipStreamCreate failed");
    }
    net->stream = net->own_stream;
    *out = net;
    return SNN_OK;
}

int snn_network_destroy(snn_network_t *net)
{
    if (!net) return SNN_OK;
    (void)hipSetDevice(net->device);
    if (net->stream) (void)hipStreamSynchronize(net->stream);
    for (void *p : net->allocs) (void)hipFree(p);
    for (void *p : {(void *)net->csr_ptr, (void *)net->csr_pre, (void *)net->csr_post, (void *)net->csr_t_ptr,
                    (void *)net->csr_t_edge, (void *)net->csr_w, (void *)net->csr_row_len, (void *)net->csr_edge_slot})
        if (p) (void)hipFree(p);
    if (net->vhist) (void)hipFree(net->vhist);
    if (net->st_vhist) (void)hipFree(net->st_vhist);
    if (net->raster) (void)hipFree(net->raster);
    if (net->preset_times_dev) (void)hipFree(net->preset_times_dev);
    if (net->trace) (void)hipFree(net->trace);
    if (net->summ_avg) (void)hipFree(net->summ_avg);
    if (net->summ_eeg) (void)hipFree(net->summ_eeg);
    for (auto &e : net->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (net->own_stream) (void)hipStreamDestroy(net->own_stream);
    delete net;
    return SNN_OK;
}

static int add_lattice_impl(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols, bool st)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "lattices cannot be added after finalize");
    if (find_lattice(net, id))   // LatticeNetworkError::GraphIDAlreadyPresent, neuron/mod.rs:1669-1671
        return fail(SNN_ERR_BAD_ARG, "lattice id " + std::to_string(id) + " already present");
    if (st && net->st_kind == SNN_ST_NONE) return fail(SNN_ERR_BAD_STATE, "network was created without a spike-train model");
    if ((uint64_t)rows * cols > 0x7FFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "lattice too large");
    LatticeInfo l{id, rows, cols, 0, rows * cols, 0, st};
    (st ? net->st_lattices : net->lattices).push_back(l);
    return SNN_OK;
}

int snn_network_add_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols)
{
    return add_lattice_impl(net, id, rows, cols, false);
}
int snn_network_add_spike_train_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols)
{
    return add_lattice_impl(net, id, rows, cols, true);
}

static int finalize_impl(snn_network_t *net, bool whole, uint32_t post_begin, uint32_t post_end, uint32_t n_shards,
                         uint32_t stride)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "already finalized");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    auto by_id = [](const LatticeInfo &a, const LatticeInfo &b) { return a.id < b.id; };
    std::sort(net->lattices.begin(), net->lattices.end(), by_id);
    std::sort(net->st_lattices.begin(), net->st_lattices.end(), by_id);
    uint64_t off = 0;
    uint32_t slot = 0;
    for (auto &l : net->lattices) { l.first = (uint32_t)off; l.slot = slot++; off += l.count; }
    net->nn = (uint32_t)off;
    slot = 0;
    for (auto &l : net->st_lattices) { l.first = (uint32_t)off; l.slot = slot++; off += l.count; }
    if (off > 0x7FFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "network too large");
    net->n_tot = (uint32_t)off;
    net->nc = net->n_tot - net->nn;
    net->n_pad = std::max<uint32_t>(256, round_up(net->nn, 256));
    net->c_pad = std::max<uint32_t>(256, round_up(net->nc, 256));
    if (whole) {
        net->q0 = 0; net->q1 = net->nn;
        net->xl = XLayout{net->n_pad, 1};
    } else {
        if (post_begin > post_end || post_end > net->nn) return fail(SNN_ERR_DIM_MISMATCH, "shard range outside the population");
        if (stride == 0 || stride % 64 != 0 || n_shards == 0 || (uint64_t)stride * n_shards < net->nn)
            return fail(SNN_ERR_BAD_ARG, "shard stride must be a multiple of 64 covering the population");
        net->q0 = post_begin; net->q1 = post_end;
        net->xl = XLayout{stride, n_shards};
        net->n_pad = std::max<uint32_t>(net->n_pad, round_up(stride * n_shards, 256));
    }
    net->n_loc = net->q1 - net->q0;
    net->ld = std::max<uint32_t>(64, round_up(net->n_loc, 64));
    // A row stride that is a multiple of 4 KiB puts the same columns of consecutive rows on the same HBM
    // channels; 256 B of padding per row de-aligns them (measured at 256x256, same process: +2 % bandwidth).
    if (net->ld % 1024 == 0) net->ld += 64;
    net->n_chunks = (net->n_tot + CHUNK - 1) / CHUNK;
    int rc = build_state(net);
    if (rc) return rc;
    rc = choose_matrix_placement(net);
    if (rc) return rc;
    net->finalized = true;
    return SNN_OK;
}

int snn_network_finalize(snn_network_t *net) { return finalize_impl(net, true, 0, 0, 1, 0); }

int snn_network_finalize_shard(snn_network_t *net, uint32_t shard_index, uint32_t n_shards)
{
    // equal slots: stride = ceil(n_neurons / n_shards) rounded up to a wavefront (64 neurons); shard r owns
    // neurons [r*stride, min(n, (r+1)*stride)) -- trailing shards may be short or empty
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (n_shards == 0 || shard_index >= n_shards) return fail(SNN_ERR_BAD_ARG, "shard_index must be < n_shards");
    uint64_t nn = 0;
    for (const auto &l : net->lattices) nn += l.count;
    const uint32_t stride = std::max<uint32_t>(64, round_up((uint32_t)((nn + n_shards - 1) / n_shards), 64));
    const uint32_t begin = (uint32_t)std::min<uint64_t>(nn, (uint64_t)shard_index * stride);
    const uint32_t end = (uint32_t)std::min<uint64_t>(nn, (uint64_t)begin + stride);
    return finalize_impl(net, false, begin, end, n_shards, stride);
}

int snn_network_sizes(const snn_network_t *net, uint32_t *n_neurons, uint32_t *n_cells, uint32_t *post_begin,
                      uint32_t *post_end)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_neurons) *n_neurons = net->nn;
    if (n_cells) *n_cells = net->nc;
    if (post_begin) *post_begin = net->q0;
    if (post_end) *post_end = net->q1;
    return SNN_OK;
}

int snn_network_lattice_range(const snn_network_t *net, uint32_t id, uint32_t *first, uint32_t *count)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id");
    if (first) *first = l->first;
    if (count) *count = l->count;
    return SNN_OK;
}

int snn_set_attr_f32(snn_network_t *net, uint32_t id, const char *name, const float *src, size_t count)
{ return attr_io(net, id, name, T_F32, const_cast<float *>(src), count, true); }
int snn_get_attr_f32(snn_network_t *net, uint32_t id, const char *name, float *dst, size_t count)
{ return attr_io(net, id, name, T_F32, dst, count, false); }
int snn_set_attr_u32(snn_network_t *net, uint32_t id, const char *name, const uint32_t *src, size_t count)
{ return attr_io(net, id, name, T_U32, const_cast<uint32_t *>(src), count, true); }
int snn_get_attr_u32(snn_network_t *net, uint32_t id, const char *name, uint32_t *dst, size_t count)
{ return attr_io(net, id, name, T_U32, dst, count, false); }
int snn_set_attr_i32(snn_network_t *net, uint32_t id, const char *name, const int32_t *src, size_t count)
{ return attr_io(net, id, name, T_I32, const_cast<int32_t *>(src), count, true); }
int snn_get_attr_i32(snn_network_t *net, uint32_t id, const char *name, int32_t *dst, size_t count)
{ return attr_io(net, id, name, T_I32, dst, count, false); }

int snn_set_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *weights,
                       const uint32_t *connections)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    return graph_rows_io(net, pre_begin, pre_count, const_cast<float *>(weights),
                         const_cast<uint32_t *>(connections), net->nn, true);
}
int snn_get_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *weights, uint32_t *connections)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    return graph_rows_io(net, pre_begin, pre_count, weights, connections, net->nn, false);
}
int snn_set_graph_dense(snn_network_t *net, const float *weights, const uint32_t *connections, size_t n_tot)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_tot != net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "graph size does not match the network");
    return graph_rows_io(net, 0, net->n_tot, const_cast<float *>(weights), const_cast<uint32_t *>(connections),
                         net->n_tot, true);
}
int snn_get_graph_dense(snn_network_t *net, float *weights, uint32_t *connections, size_t n_tot)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_tot != net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "graph size does not match the network");
    return graph_rows_io(net, 0, net->n_tot, weights, connections, net->n_tot, false);
}

int snn_fill_graph_synthetic(snn_network_t *net, uint64_t seed, float lo, float hi, int with_diagonal)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "the synthetic dense graph needs a dense handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    if (net->n_tot && net->ld) {
        const unsigned gy = std::min<uint32_t>(net->n_tot, 4096);
        hipLaunchKernelGGL(k_graph_synthetic, dim3((net->ld + 255) / 256, gy), dim3(256), 0, net->stream, net->W,
                           net->ld, net->n_loc, net->q0, net->nn, net->n_tot, seed, lo, hi, with_diagonal);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    }
    net->counts_dirty = true;
    return SNN_OK;
}

int snn_network_use_csr(snn_network_t *net, int enable)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "the graph form is fixed at finalize");
    net->csr = enable != 0;
    return SNN_OK;
}

int snn_set_graph_csr(snn_network_t *net, const uint64_t *row_ptr, const uint32_t *pre_index, const float *weights,
                      uint64_t nnz)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph: call snn_network_use_csr before finalize");
    if (!row_ptr || (nnz && (!pre_index || !weights))) return fail(SNN_ERR_BAD_ARG, "null graph pointer");
    if (nnz >= 0xFFFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "more than 2^32-1 stored synapses per handle");
    const uint32_t n_loc = net->n_loc;
    if (row_ptr[0] != 0 || row_ptr[n_loc] != nnz) return fail(SNN_ERR_DIM_MISMATCH, "row_ptr must run from 0 to nnz");
    const uint32_t n_slices = (n_loc + 63) / 64;
    std::vector<uint32_t> slice_ptr((size_t)n_slices + 1, 0), row_len((size_t)n_slices * 64, 0), post(nnz),
        t_ptr((size_t)net->n_tot + 1, 0), t_edge(nnz), edge_slot(nnz);
    for (uint32_t q = 0; q < n_loc; ++q) {
        if (row_ptr[q + 1] < row_ptr[q]) return fail(SNN_ERR_DIM_MISMATCH, "row_ptr is not monotone");
        row_len[q] = (uint32_t)(row_ptr[q + 1] - row_ptr[q]);
        for (uint64_t e = row_ptr[q]; e < row_ptr[q + 1]; ++e) {
            if (pre_index[e] >= net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "presynaptic index out of range");
            if (e > row_ptr[q] && pre_index[e] <= pre_index[e - 1])
                return fail(SNN_ERR_BAD_ARG, "presynaptic indices of a row must be strictly ascending");
            post[e] = q;
            ++t_ptr[pre_index[e] + 1];
        }
    }
    // SELL-64: every slice (64 rows) is padded to its longest row
    uint64_t entries = 0;
    for (uint32_t sl = 0; sl < n_slices; ++sl) {
        uint32_t width = 0;
        for (uint32_t r = sl * 64; r < sl * 64 + 64; ++r) width = std::max(width, row_len[r]);
        slice_ptr[sl] = (uint32_t)entries;
        entries += (uint64_t)width * 64;
        if (entries >= 0xFFFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "sparse graph too large for 32-bit slots");
    }
    slice_ptr[n_slices] = (uint32_t)entries;
    std::vector<uint32_t> sell_pre(entries, SELL_PAD);
    std::vector<float> sell_w(entries, 0.0f);
    for (uint32_t q = 0; q < n_loc; ++q) {
        const uint32_t base = slice_ptr[q >> 6] + (q & 63u);
        for (uint64_t e = row_ptr[q]; e < row_ptr[q + 1]; ++e) {
            const uint32_t slot = base + (uint32_t)(e - row_ptr[q]) * 64;
            sell_pre[slot] = pre_index[e];
            sell_w[slot] = weights[e];
            edge_slot[e] = slot;
        }
    }
    for (size_t p = 0; p < net->n_tot; ++p) t_ptr[p + 1] += t_ptr[p];
    {
        std::vector<uint32_t> fill(t_ptr.begin(), t_ptr.end() - 1);
        for (uint64_t e = 0; e < nnz; ++e) t_edge[fill[pre_index[e]]++] = (uint32_t)e;
    }
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    for (void **p : {(void **)&net->csr_ptr, (void **)&net->csr_pre, (void **)&net->csr_post, (void **)&net->csr_t_ptr,
                     (void **)&net->csr_t_edge, (void **)&net->csr_w, (void **)&net->csr_row_len,
                     (void **)&net->csr_edge_slot}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, std::max<size_t>(bytes, 256)), SNN_ERR_BUFFER_CREATE);
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        return SNN_OK;
    };
    TRY(up((void **)&net->csr_ptr, slice_ptr.data(), slice_ptr.size() * 4));
    TRY(up((void **)&net->csr_pre, sell_pre.data(), entries * 4));
    TRY(up((void **)&net->csr_w, sell_w.data(), entries * 4));
    TRY(up((void **)&net->csr_row_len, row_len.data(), row_len.size() * 4));
    TRY(up((void **)&net->csr_edge_slot, edge_slot.data(), nnz * 4));
    TRY(up((void **)&net->csr_post, post.data(), nnz * 4));
    TRY(up((void **)&net->csr_t_ptr, t_ptr.data(), t_ptr.size() * 4));
    TRY(up((void **)&net->csr_t_edge, t_edge.data(), nnz * 4));
    if (net->trace) { (void)hipFree(net->trace); net->trace = nullptr; }      // traces belong to the replaced edges
    net->nnz = nnz;
    net->sell_entries = entries;
    net->edge_slot_host.swap(edge_slot);
    net->counts_dirty = true;
    return SNN_OK;
}

int snn_get_graph_csr(snn_network_t *net, float *weights, uint64_t nnz)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph");
    if (nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    if (nnz == 0) return SNN_OK;
    if (!weights) return fail(SNN_ERR_BAD_ARG, "weights is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    std::vector<float> sell((size_t)net->sell_entries);
    HIP_TRY(hipMemcpy(sell.data(), net->csr_w, sell.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    for (uint64_t e = 0; e < nnz; ++e) weights[e] = sell[net->edge_slot_host[e]];
    return SNN_OK;
}

int snn_set_synapses(snn_network_t *net, int electrical_synapse, int chemical_synapse)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->electrical = electrical_synapse ? 1 : 0;
    net->chemical = chemical_synapse ? 1 : 0;
    return SNN_OK;
}

int snn_set_plasticity(snn_network_t *net, uint32_t id, float a_plus, float a_minus, float tau_plus,
                       float tau_minus, float dt, int do_plasticity)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "plasticity belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    float *s = &net->stdp_host[(size_t)l->slot * 5];
    s[0] = a_plus; s[1] = a_minus; s[2] = tau_plus; s[3] = tau_minus; s[4] = dt;
    net->plast_host[l->slot] = do_plasticity ? 1u : 0u;
    net->any_plasticity = false;
    for (uint32_t p : net->plast_host) net->any_plasticity |= (p != 0);
    TRY(end_run(net));
    HIP_TRY(hipMemcpy(net->stdp_dev, net->stdp_host.data(), net->stdp_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpy(net->plast_dev, net->plast_host.data(), net->plast_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}

namespace {
size_t trace_elems(const snn_network *net) { return net->csr ? (size_t)net->sell_entries : (size_t)net->n_tot * net->ld; }

int ensure_traces(snn_network *net)
{
    if (net->trace) return SNN_OK;
    const size_t n = std::max<size_t>(trace_elems(net), 64);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&net->trace), n * 4), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(hipMemsetAsync(net->trace, 0, n * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}
} // namespace

int snn_set_reward_modulator(snn_network_t *net, uint32_t id, float dopamine, float tau_d, float tau_c, float a_plus,
                             float a_minus, float tau_plus, float tau_minus, float dt, int do_modulation)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reward modulation belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t nl = net->rm_on_host.size();
    // dopamine of the other lattices evolves on the device: refresh the host copy before rewriting the table
    HIP_TRY(hipMemcpy(net->rm_host.data(), net->rm_dev, nl * RM_STRIDE * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    float *m = &net->rm_host[(size_t)l->slot * RM_STRIDE];
    m[0] = dopamine; m[1] = tau_d; m[2] = tau_c; m[3] = a_plus; m[4] = a_minus; m[5] = tau_plus; m[6] = tau_minus; m[7] = dt;
    net->rm_on_host[l->slot] = do_modulation ? 1u : 0u;
    net->any_modulation = false;
    for (uint32_t v : net->rm_on_host) net->any_modulation |= (v != 0);
    if (do_modulation) {
        // a RewardModulatedLattice has no STDP rule of its own
        net->plast_host[l->slot] = 0;
        net->any_plasticity = false;
        for (uint32_t p : net->plast_host) net->any_plasticity |= (p != 0);
        HIP_TRY(hipMemcpy(net->plast_dev, net->plast_host.data(), net->plast_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        if (net->csr && !net->csr_ptr) return fail(SNN_ERR_BAD_STATE, "set the sparse graph before enabling reward modulation");
        TRY(ensure_traces(net));
    }
    HIP_TRY(hipMemcpy(net->rm_dev, net->rm_host.data(), nl * RM_STRIDE * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpy(net->rm_on_dev, net->rm_on_host.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    hipLaunchKernelGGL(k_modulator_update, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, net->stream, net->rm_dev,
                       net->rm_on_dev, (uint32_t)nl, 0.0f, 1);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

int snn_get_dopamine(snn_network_t *net, uint32_t id, float *dopamine)
{
    if (!net || !dopamine) return fail(SNN_ERR_BAD_ARG, "null pointer");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reward modulation belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    HIP_TRY(hipMemcpy(dopamine, net->rm_dev + (size_t)l->slot * RM_STRIDE, 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int snn_apply_reward(snn_network_t *net, float reward)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->any_modulation) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    const size_t nl = net->rm_on_host.size();
    hipLaunchKernelGGL(k_modulator_update, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, net->stream, net->rm_dev,
                       net->rm_on_dev, (uint32_t)nl, reward, 0);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

int snn_run_with_reward(snn_network_t *net, float reward)
{
    int rc = snn_apply_reward(net, reward);
    return rc ? rc : snn_run(net, 1);
}

// TraceRSTDP::c of the edges in presynaptic rows [pre_begin, pre_begin + pre_count), row-major [pre_count][n_neurons];
// a shard handle reads / writes its own columns only.
static int trace_rows_io(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *traces, bool set)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a CSR graph: use snn_set_traces_csr / snn_get_traces_csr");
    if ((uint64_t)pre_begin + pre_count > net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "row range exceeds n_tot");
    if (pre_count == 0 || net->nn == 0 || net->n_loc == 0) return SNN_OK;
    if (!traces) return fail(SNN_ERR_BAD_ARG, "traces is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_traces(net));
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    float *dev = net->trace + (size_t)pre_begin * net->ld;
    float *host = traces + net->q0;
    if (set)
        HIP_TRY(hipMemcpy2D(dev, (size_t)net->ld * 4, host, (size_t)net->nn * 4, (size_t)net->n_loc * 4, pre_count,
                            hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    else
        HIP_TRY(hipMemcpy2D(host, (size_t)net->nn * 4, dev, (size_t)net->ld * 4, (size_t)net->n_loc * 4, pre_count,
                            hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}
int snn_set_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *traces)
{ return trace_rows_io(net, pre_begin, pre_count, const_cast<float *>(traces), true); }
int snn_get_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *traces)
{ return trace_rows_io(net, pre_begin, pre_count, traces, false); }

static int traces_csr_io(snn_network_t *net, float *traces, uint64_t nnz, bool set)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph");
    if (nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    if (nnz == 0) return SNN_OK;
    if (!traces) return fail(SNN_ERR_BAD_ARG, "traces is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_traces(net));
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    std::vector<float> sell((size_t)net->sell_entries);
    HIP_TRY(hipMemcpy(sell.data(), net->trace, sell.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    if (set) {
        for (uint64_t e = 0; e < nnz; ++e) sell[net->edge_slot_host[e]] = traces[e];
        HIP_TRY(hipMemcpy(net->trace, sell.data(), sell.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    } else {
        for (uint64_t e = 0; e < nnz; ++e) traces[e] = sell[net->edge_slot_host[e]];
    }
    return SNN_OK;
}
int snn_set_traces_csr(snn_network_t *net, const float *traces, uint64_t nnz)
{ return traces_csr_io(net, const_cast<float *>(traces), nnz, true); }
int snn_get_traces_csr(snn_network_t *net, float *traces, uint64_t nnz)
{ return traces_csr_io(net, traces, nnz, false); }

int snn_set_history(snn_network_t *net, int voltage_history, int spike_history)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if ((voltage_history != 0) != (net->want_vhist != 0) || (spike_history != 0) != (net->want_raster != 0)) {
        // switching what is recorded restarts the record so that all rows cover the same steps
        net->hist_steps = 0; net->hist_tick = 0;
    }
    net->want_vhist = voltage_history ? 1 : 0;
    net->want_raster = spike_history ? 1 : 0;
    return SNN_OK;
}

int snn_reset_history(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->hist_steps = 0; net->hist_tick = 0;
    if (net->finalized && net->spike_counts) {
        HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
        TRY(end_run(net));
        HIP_TRY(hipMemset(net->spike_counts, 0, (size_t)net->n_pad * 4), SNN_ERR_BUFFER_WRITE);
    }
    return SNN_OK;
}

int snn_set_firing_times(snn_network_t *net, uint32_t id, const uint32_t *cell_ptr, const float *times, size_t n_times)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->st_kind != SNN_ST_PRESET) return fail(SNN_ERR_BAD_STATE, "the spike-train model is not SNN_ST_PRESET");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || !l->spike_train) return fail(SNN_ERR_BAD_ARG, "no such spike-train lattice");
    if (l->count == 0) return SNN_OK;
    if (!cell_ptr || (n_times && !times)) return fail(SNN_ERR_BAD_ARG, "null pointer");
    if (cell_ptr[0] != 0 || cell_ptr[l->count] != n_times) return fail(SNN_ERR_DIM_MISMATCH, "cell_ptr must run from 0 to n_times");
    for (uint32_t i = 0; i < l->count; ++i)
        if (cell_ptr[i] > cell_ptr[i + 1]) return fail(SNN_ERR_BAD_ARG, "cell_ptr must be non-decreasing");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const uint32_t c0 = l->first - net->nn;
    for (uint32_t i = 0; i < l->count; ++i) net->preset_host[c0 + i].assign(times + cell_ptr[i], times + cell_ptr[i + 1]);
    std::vector<uint32_t> ptr((size_t)net->c_pad + 1, 0);
    std::vector<float> flat;
    for (uint32_t s = 0; s < net->nc; ++s) {
        ptr[s] = (uint32_t)flat.size();
        flat.insert(flat.end(), net->preset_host[s].begin(), net->preset_host[s].end());
    }
    for (size_t s = net->nc; s <= net->c_pad; ++s) ptr[s] = (uint32_t)flat.size();
    float *nt = nullptr;
    HIP_TRY(hipMalloc(&nt, std::max<size_t>(256, flat.size() * 4)), SNN_ERR_BUFFER_CREATE);
    if (!flat.empty()) HIP_TRY(hipMemcpy(nt, flat.data(), flat.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpy(const_cast<uint32_t *>(net->ca.preset_ptr), ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice),
            SNN_ERR_BUFFER_WRITE);
    if (net->preset_times_dev) (void)hipFree(net->preset_times_dev);
    net->preset_times_dev = nt;
    net->ca.preset_times = nt;
    return SNN_OK;
}

int snn_set_history_stride(snn_network_t *net, uint32_t every)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (every == 0) return fail(SNN_ERR_BAD_ARG, "stride must be at least 1");
    if (every != net->hist_every) { net->hist_steps = 0; net->hist_tick = 0; }
    net->hist_every = every;
    return SNN_OK;
}

int snn_set_reduced_history(snn_network_t *net, int average_voltage, int eeg, int spike_counts,
                            float reference_voltage, float distance, float conductivity)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if ((average_voltage != 0) != (net->want_avg != 0) || (eeg != 0) != (net->want_eeg != 0))
        net->hist_steps = 0, net->hist_tick = 0;      // all recorded rows must cover the same steps
    net->want_avg = average_voltage ? 1 : 0;
    net->want_eeg = eeg ? 1 : 0;
    net->want_counts = spike_counts ? 1 : 0;
    net->eeg_ref = reference_voltage; net->eeg_dist = distance; net->eeg_cond = conductivity;
    return SNN_OK;
}

static int get_summary(snn_network_t *net, uint32_t id, float *dst, size_t steps, bool eeg)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reduced histories exist for neuron lattices");
    if (!(eeg ? net->want_eeg : net->want_avg)) return fail(SNN_ERR_BAD_STATE, "this reduced history is off");
    if (steps != net->hist_steps) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (steps == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t nl = net->lattices.size();
    const float *src = (eeg ? net->summ_eeg : net->summ_avg) + l->slot;
    HIP_TRY(hipMemcpy2D(dst, 4, src, nl * 4, 4, steps, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int snn_get_average_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t steps)
{ return get_summary(net, id, dst, steps, false); }
int snn_get_eeg_history(snn_network_t *net, uint32_t id, float *dst, size_t steps)
{ return get_summary(net, id, dst, steps, true); }

int snn_get_spike_counts(snn_network_t *net, uint32_t id, uint32_t *dst, size_t count)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "spike counts exist for neuron lattices");
    if (count != l->count) return fail(SNN_ERR_DIM_MISMATCH, "count must equal rows*cols");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    HIP_TRY(hipMemcpy(dst, net->spike_counts + l->first, count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int snn_get_clock(const snn_network_t *net, uint64_t *clock)
{
    if (!net || !clock) return fail(SNN_ERR_BAD_ARG, "null argument");
    *clock = (uint64_t)net->clock;
    return SNN_OK;
}

int snn_reset_timing(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    net->clock = 0;
    net->view_dirty = true;
    for (auto &c : net->st_clock) c = 0;
    HIP_TRY(hipMemsetAsync(net->na.last_firing_time, 0xFF, (size_t)net->n_pad * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->ca.last_firing_time, 0xFF, (size_t)net->c_pad * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}

int snn_run(snn_network_t *net, uint64_t iterations)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->xl.n_shards > 1) return fail(SNN_ERR_BAD_STATE, "sharded handles are stepped with snn_step_begin/end");
    if (iterations == 0 || net->n_tot == 0) return SNN_OK;          // gpu_lattices/mod.rs:1089-1091, 3196-3203
    if (!net->electrical && !net->chemical) return SNN_OK;           // neuron/mod.rs:1217, 2672
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, iterations));
    for (uint64_t it = 0; it < iterations; ++it) {
        if (net->nn) TRY(step_begin(net));
        TRY(step_end(net));
        if (net->profile && net->ev_used >= 8192) TRY(collect_profile(net));
    }
    return end_run(net);
}

int snn_step_begin_local(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, 1));
    // Only when nothing of the previous step is still pending for these chunks: STDP rewrites W in
    // snn_step_end and needs the gathered spikes first, so with plasticity on the split is not taken.
    if (net->nn && !net->csr && !net->any_plasticity && !net->any_modulation && !net->local_inputs_done) {
        TRY(launch_inputs(net, INPUTS_LOCAL));
        net->local_inputs_done = true;
    }
    return SNN_OK;
}

int snn_step_begin(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;           // neuron/mod.rs:1217, 2672
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, 1));
    if (net->nn) TRY(step_begin(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}

int snn_step_end(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    if (!net->run_active) return fail(SNN_ERR_BAD_STATE, "snn_step_end without snn_step_begin");
    TRY(step_end(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}

int snn_exchange_buffer(snn_network_t *net, void **device_ptr, uint32_t *words_per_neuron, uint32_t *n_padded)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (device_ptr) *device_ptr = net->xbuf;
    if (words_per_neuron) *words_per_neuron = NUM_PLANES;
    if (n_padded) *n_padded = net->xl.stride * net->xl.n_shards;
    return SNN_OK;
}

int snn_set_stream(snn_network_t *net, void *hip_stream)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (hip_stream) {
        net->stream = static_cast<hipStream_t>(hip_stream);
        net->external_stream = true;
    } else {
        net->stream = net->own_stream;
        net->external_stream = false;
    }
    return SNN_OK;
}

int snn_synchronize(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    return end_run(net);
}

int snn_stream(snn_network_t *net, void **hip_stream)
{
    if (!net || !hip_stream) return fail(SNN_ERR_BAD_ARG, "null argument");
    *hip_stream = net->stream;
    return SNN_OK;
}

int snn_history_steps(const snn_network_t *net, uint64_t *steps)
{
    if (!net || !steps) return fail(SNN_ERR_BAD_ARG, "null argument");
    *steps = net->hist_steps;
    return SNN_OK;
}

int snn_get_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t count)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id");
    if (!net->want_vhist) return fail(SNN_ERR_BAD_STATE, "voltage history is off");
    if (count != net->hist_steps * l->count) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const float *src = l->spike_train ? net->st_vhist + (l->first - net->nn) : net->vhist + l->first;
    const size_t pitch = (size_t)(l->spike_train ? net->c_pad : net->n_pad) * 4;
    HIP_TRY(hipMemcpy2D(dst, (size_t)l->count * 4, src, pitch, (size_t)l->count * 4, net->hist_steps, hipMemcpyDeviceToHost),
            SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int snn_get_spike_history(snn_network_t *net, uint32_t id, uint8_t *dst, size_t count)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "spike history exists for neuron lattices");
    if (!net->want_raster) return fail(SNN_ERR_BAD_STATE, "spike history is off");
    if (count != net->hist_steps * l->count) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t words = net->n_pad / 64;
    std::vector<unsigned long long> host(net->hist_steps * words);
    HIP_TRY(hipMemcpy(host.data(), net->raster, host.size() * 8, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    for (uint64_t s = 0; s < net->hist_steps; ++s)
        for (uint32_t i = 0; i < l->count; ++i) {
            const uint32_t q = l->first + i;
            dst[s * l->count + i] = (uint8_t)((host[s * words + (q >> 6)] >> (q & 63)) & 1ull);
        }
    return SNN_OK;
}

int snn_profile_enable(snn_network_t *net, int enable)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->profile = enable ? 1 : 0;
    return SNN_OK;
}
int snn_profile_reset(snn_network_t *net)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    net->ev_used = 0; net->prof_launches = 0; net->prof_ms = 0.0;
    return SNN_OK;
}
int snn_profile_read(snn_network_t *net, uint64_t *launches, double *total_ms)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(collect_profile(net));
    if (launches) *launches = net->prof_launches;
    if (total_ms) *total_ms = net->prof_ms;
    return SNN_OK;
}
int snn_input_kernel_bytes(const snn_network_t *net, uint64_t *bytes)
{
    if (!net || !bytes) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    *bytes = net->csr ? (uint64_t)8 * net->nnz                 // sparse: index + weight of every stored synapse
                      : (uint64_t)4 * net->n_tot * net->n_loc; // dense: every weight of the shard, read once
    return SNN_OK;
}

int snn_probe_bandwidth(int device, uint64_t bytes, int repeats, double *read_gbps, double *copy_gbps)
{
    if (!read_gbps || !copy_gbps) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (bytes < (1u << 20) || repeats <= 0) return fail(SNN_ERR_BAD_ARG, "need >= 1 MiB and >= 1 repeat");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    const size_t n4 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    float *sink = nullptr;
    HIP_TRY(hipMalloc(&a, n4 * 16), SNN_ERR_BUFFER_CREATE);
    if (hipMalloc(&b, n4 * 16) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&sink), 256) != hipSuccess) {
        (void)hipFree(a);
        if (b) (void)hipFree(b);
        return fail(SNN_ERR_BUFFER_CREATE, "probe allocation failed");
    }
    int rc = SNN_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemset(a, 0, n4 * 16) != hipSuccess || hipMemset(b, 0, n4 * 16) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        rc = fail(SNN_ERR_QUEUE, "probe setup failed");
    auto timed = [&](bool copy, double *out) {
        const unsigned blocks = 256 * 8;     // 8 workgroups per CU, grid-stride
        for (int warm = 0; warm < 2; ++warm) {
            if (copy) hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, (probe_v4f *)b, n4);
            else hipLaunchKernelGGL(k_probe_read, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, n4, sink);
        }
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < repeats; ++r) {
            if (copy) hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, (probe_v4f *)b, n4);
            else hipLaunchKernelGGL(k_probe_read, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, n4, sink);
        }
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(SNN_ERR_WAIT, "probe kernel failed"); return; }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *out = (double)(copy ? 2 : 1) * (double)(n4 * 16) * repeats / (ms * 1e-3) / 1e9;
    };
    if (rc == SNN_OK) timed(false, read_gbps);
    if (rc == SNN_OK) timed(true, copy_gbps);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return rc;
}

int snn_probe_math(int device, int which, const float *in, float *out, size_t count)
{
    if (!in || !out) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (which < 0 || which > 2) return fail(SNN_ERR_BAD_ARG, "unknown function selector");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    if (count == 0) return SNN_OK;
    float *di = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&di), count * 4), SNN_ERR_BUFFER_CREATE);
    if (hipMalloc(reinterpret_cast<void **>(&dout), count * 4) != hipSuccess) { (void)hipFree(di); return fail(SNN_ERR_BUFFER_CREATE, "hipMalloc failed"); }
    int rc = SNN_OK;
    if (hipMemcpy(di, in, count * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SNN_ERR_BUFFER_WRITE, "upload failed");
    if (rc == SNN_OK) {
        hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, which, di, dout, count);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail(SNN_ERR_WAIT, "probe kernel failed");
    }
    if (rc == SNN_OK && hipMemcpy(out, dout, count * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SNN_ERR_BUFFER_READ, "download failed");
    (void)hipFree(di);
    (void)hipFree(dout);
    return rc;
}

} // extern "C"

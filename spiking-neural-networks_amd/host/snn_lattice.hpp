// snn_lattice.hpp -- C++17 host-side mirror of the reference's lattice containers over the C ABI
// (include/snn_amd.h).  The reference's host language is Rust, which this image lacks; this header
// keeps the reference's operator interface for the hot path -- same type / method / field names,
// argument meaning and error behaviour -- so that code written against
//   Lattice / LatticeGPU            backend/src/neuron/mod.rs:556-1220, neuron/gpu_lattices/mod.rs:327-1118
//   SpikeTrainLattice                backend/src/neuron/mod.rs:1292-1436
//   LatticeNetwork / LatticeNetworkGPU   neuron/mod.rs:1538-2675, neuron/gpu_lattices/mod.rs:1517-3212
//   AdjacencyMatrix                  backend/src/graph/mod.rs:139-297
//   GPUError                         backend/src/error/mod.rs:221-238
// reads the same.  Header-only; link with libsnn_amd.so.  There is no CPU stepping here: `run_lattice`
// exists on the *GPU containers only, exactly where the drop-in boundary sits.
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/snn_amd.h"

namespace snn_host {

using Position = std::pair<size_t, size_t>;

// ---- errors ------------------------------------------------------------------------------------
// GPUError's variants in declaration order (error/mod.rs:221-238) are the ABI's codes 1..8.
struct GPUError : std::runtime_error {
    int code;
    GPUError(int c, const std::string &m) : std::runtime_error("GPUError(" + std::to_string(c) + "): " + m), code(c) {}
};
struct GraphError : std::runtime_error { using std::runtime_error::runtime_error; };   // error/mod.rs:14-33
struct LatticeNetworkError : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(int rc)
{
    if (rc != SNN_OK) throw GPUError(rc, snn_last_error());
}

// ---- neurotransmitters / receptors (iterate_and_spike/mod.rs:122-205, 394-446, 1078-1305) ------
enum IonotropicNeurotransmitterType { AMPA = 0, NMDA = 1, GABA = 2 };

struct Neurotransmitter {          // Approximate: t, t_max, clearance_constant; Destexhe: t, t_max, v_p, k_p
    float t = 0.0f, t_max = 1.0f, clearance_constant = 0.01f, v_p = 2.0f, k_p = 5.0f;
};
struct Receptor {                  // AMPAReceptor / NMDAReceptor / GABAReceptor + their kinetics
    float g = 1.0f, e = 0.0f, mg = 0.0f, current = 0.0f, r = 0.0f, alpha = 1.0f, beta = 1.0f;
    static Receptor default_for(int type)
    {
        Receptor x;
        if (type == NMDA) { x.g = 0.6f; x.mg = 0.3f; }
        if (type == GABA) { x.g = 1.2f; x.e = -80.0f; }
        return x;
    }
};
using Neurotransmitters = std::map<int, Neurotransmitter>;   // Neurotransmitters<N, T>::insert/remove
using Ionotropic = std::map<int, Receptor>;                   // Ionotropic<R>::insert/remove

// ---- neuron models: fields = the reference's pub fields, defaults = its Default impls -----------
struct NeuronBase {
    float current_voltage, gap_conductance = 7.0f, dt, c_m, v_th;
    bool is_spiking = false;
    std::optional<size_t> last_firing_time;
    Neurotransmitters synaptic_neurotransmitters;
    Ionotropic receptors;
};


struct IzhikevichNeuron : NeuronBase {          // integrate_and_fire/mod.rs:1159-1220
    float v_init = -65.0f, a = 0.02f, b = 0.2f, c = -55.0f, d = 8.0f, w_value = 30.0f, w_init = 30.0f, tau_m = 1.0f;
    IzhikevichNeuron() { current_voltage = -65.0f; dt = 0.1f; c_m = 100.0f; v_th = 30.0f; }
    static constexpr int MODEL = SNN_MODEL_IZHIKEVICH;
    static std::vector<std::pair<const char *, float IzhikevichNeuron::*>> fields()
    {
        return {{"w_value", &IzhikevichNeuron::w_value}, {"a", &IzhikevichNeuron::a}, {"b", &IzhikevichNeuron::b},
                {"c", &IzhikevichNeuron::c}, {"d", &IzhikevichNeuron::d}, {"tau_m", &IzhikevichNeuron::tau_m}};
    }
};

struct LeakyIntegrateAndFireNeuron : NeuronBase {   // integrate_and_fire/mod.rs:108-171
    float v_reset = -75.0f, v_init = -75.0f, refractory_count = 0.0f, tref = 10.0f, leak_constant = -1.0f,
          integration_constant = 1.0f, e_l = -75.0f, g_l = 10.0f, tau_m = 10.0f;
    LeakyIntegrateAndFireNeuron() { current_voltage = -75.0f; dt = 0.1f; c_m = 100.0f; v_th = -55.0f; }
    static constexpr int MODEL = SNN_MODEL_LIF;
    static std::vector<std::pair<const char *, float LeakyIntegrateAndFireNeuron::*>> fields()
    {
        using T = LeakyIntegrateAndFireNeuron;
        return {{"v_reset", &T::v_reset}, {"refractory_count", &T::refractory_count}, {"tref", &T::tref},
                {"leak_constant", &T::leak_constant}, {"integration_constant", &T::integration_constant},
                {"e_l", &T::e_l}, {"g_l", &T::g_l}, {"tau_m", &T::tau_m}};
    }
};

struct QuadraticIntegrateAndFireNeuron : NeuronBase {   // integrate_and_fire/mod.rs:259-322 (has a reference GPU impl)
    float v_reset = -75.0f, v_init = -75.0f, refractory_count = 0.0f, tref = 10.0f, alpha = 1.0f, v_c = -60.0f,
          integration_constant = 1.0f, tau_m = 100.0f;
    QuadraticIntegrateAndFireNeuron() { current_voltage = -75.0f; dt = 0.1f; c_m = 100.0f; v_th = -55.0f; }
    static constexpr int MODEL = SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE;
    static std::vector<std::pair<const char *, float QuadraticIntegrateAndFireNeuron::*>> fields()
    {
        using T = QuadraticIntegrateAndFireNeuron;
        return {{"v_reset", &T::v_reset}, {"refractory_count", &T::refractory_count}, {"tref", &T::tref},
                {"alpha", &T::alpha}, {"v_c", &T::v_c}, {"integration_constant", &T::integration_constant},
                {"tau_m", &T::tau_m}};
    }
};

struct SimpleLeakyIntegrateAndFire : NeuronBase {       // integrate_and_fire/mod.rs:1523-1570 (has a reference GPU impl)
    float g = -0.1f, e = 0.0f, v_reset = -75.0f, v_init = -75.0f;
    SimpleLeakyIntegrateAndFire() { current_voltage = -75.0f; gap_conductance = 10.0f; dt = 0.1f; c_m = 100.0f; v_th = -55.0f; }
    static constexpr int MODEL = SNN_MODEL_SIMPLE_LIF;
    static std::vector<std::pair<const char *, float SimpleLeakyIntegrateAndFire::*>> fields()
    {
        using T = SimpleLeakyIntegrateAndFire;
        return {{"g", &T::g}, {"e", &T::e}, {"v_reset", &T::v_reset}};
    }
};

struct AdaptiveLeakyIntegrateAndFireNeuron : NeuronBase {   // integrate_and_fire/mod.rs:918-996
    float v_reset = -75.0f, v_init = -75.0f, refractory_count = 0.0f, tref = 10.0f, alpha = 6.0f, beta = 10.0f,
          w_value = 0.0f, w_init = 0.0f, leak_constant = -1.0f, integration_constant = 1.0f, e_l = -75.0f, g_l = 10.0f,
          tau_m = 10.0f;
    AdaptiveLeakyIntegrateAndFireNeuron() { current_voltage = -75.0f; dt = 0.1f; c_m = 100.0f; v_th = -55.0f; }
    static constexpr int MODEL = SNN_MODEL_ADAPTIVE_LIF;
    static std::vector<std::pair<const char *, float AdaptiveLeakyIntegrateAndFireNeuron::*>> fields()
    {
        using T = AdaptiveLeakyIntegrateAndFireNeuron;
        return {{"v_reset", &T::v_reset}, {"refractory_count", &T::refractory_count}, {"tref", &T::tref},
                {"alpha", &T::alpha}, {"beta", &T::beta}, {"w_value", &T::w_value}, {"leak_constant", &T::leak_constant},
                {"integration_constant", &T::integration_constant}, {"e_l", &T::e_l}, {"g_l", &T::g_l},
                {"tau_m", &T::tau_m}};
    }
};

struct AdaptiveExpLeakyIntegrateAndFireNeuron : NeuronBase {   // integrate_and_fire/mod.rs:1051-1130
    float v_reset = -75.0f, v_init = -75.0f, refractory_count = 0.0f, tref = 10.0f, alpha = 6.0f, beta = 10.0f,
          slope_factor = 1.0f, w_value = 0.0f, w_init = 0.0f, leak_constant = -1.0f, integration_constant = 1.0f,
          e_l = -75.0f, g_l = 10.0f, tau_m = 10.0f;
    AdaptiveExpLeakyIntegrateAndFireNeuron() { current_voltage = -75.0f; dt = 0.1f; c_m = 100.0f; v_th = -55.0f; }
    static constexpr int MODEL = SNN_MODEL_ADAPTIVE_EXP_LIF;
    static std::vector<std::pair<const char *, float AdaptiveExpLeakyIntegrateAndFireNeuron::*>> fields()
    {
        using T = AdaptiveExpLeakyIntegrateAndFireNeuron;
        return {{"v_reset", &T::v_reset}, {"refractory_count", &T::refractory_count}, {"tref", &T::tref},
                {"alpha", &T::alpha}, {"beta", &T::beta}, {"slope_factor", &T::slope_factor}, {"w_value", &T::w_value},
                {"leak_constant", &T::leak_constant}, {"integration_constant", &T::integration_constant},
                {"e_l", &T::e_l}, {"g_l", &T::g_l}, {"tau_m", &T::tau_m}};
    }
};

struct LeakyIzhikevichNeuron : NeuronBase {     // integrate_and_fire/mod.rs:1270-1331
    float v_init = -65.0f, a = 0.02f, b = 0.2f, c = -55.0f, d = 8.0f, w_value = 30.0f, w_init = 30.0f, e_l = -65.0f,
          tau_m = 10.0f;
    LeakyIzhikevichNeuron() { current_voltage = -65.0f; dt = 0.1f; c_m = 100.0f; v_th = 30.0f; }
    static constexpr int MODEL = SNN_MODEL_LEAKY_IZHIKEVICH;
    static std::vector<std::pair<const char *, float LeakyIzhikevichNeuron::*>> fields()
    {
        using T = LeakyIzhikevichNeuron;
        return {{"w_value", &T::w_value}, {"a", &T::a}, {"b", &T::b}, {"c", &T::c}, {"d", &T::d}, {"tau_m", &T::tau_m},
                {"e_l", &T::e_l}};
    }
};

struct HodgkinHuxleyNeuron : NeuronBase {      // hodgkin_huxley/mod.rs:49-98, ion_channels/mod.rs:192-317
    float m_state = 0.0f, h_state = 0.0f, n_state = 0.0f;
    float g_na = 120.0f, e_na = 50.0f, g_k = 36.0f, e_k = -77.0f, g_k_leak = 0.3f, e_k_leak = -55.0f;
    float na_current = 0.0f, k_current = 0.0f, k_leak_current = 0.0f;
    bool was_increasing = false;
    HodgkinHuxleyNeuron() { current_voltage = -65.0f; dt = 0.01f; c_m = 1.0f; v_th = 0.0f; }
    static constexpr int MODEL = SNN_MODEL_HODGKIN_HUXLEY;
    static std::vector<std::pair<const char *, float HodgkinHuxleyNeuron::*>> fields()
    {
        using T = HodgkinHuxleyNeuron;
        return {{"na_channel$m$state", &T::m_state}, {"na_channel$h$state", &T::h_state}, {"k_channel$n$state", &T::n_state},
                {"na_channel$g_na", &T::g_na}, {"na_channel$e_na", &T::e_na}, {"k_channel$g_k", &T::g_k},
                {"k_channel$e_k", &T::e_k}, {"k_leak_channel$g_k_leak", &T::g_k_leak},
                {"k_leak_channel$e_k_leak", &T::e_k_leak}, {"na_channel$current", &T::na_current},
                {"k_channel$current", &T::k_current}, {"k_leak_channel$current", &T::k_leak_current}};
    }
};

// ---- spike trains (spike_train/mod.rs:259-371, 975-1031) ---------------------------------------
struct SpikeTrainBase {
    float current_voltage = 0.0f, v_th = 30.0f, v_resting = 0.0f, dt = 0.1f;
    float neural_refractoriness_k = 10000.0f;     // DeltaDiracRefractoriness::k
    bool is_spiking = false;
    std::optional<size_t> last_firing_time;
    Neurotransmitters synaptic_neurotransmitters;
};
struct PoissonNeuron : SpikeTrainBase {
    float chance_of_firing = 0.0f;
    uint32_t seed = 1;                            // xorshift32 state (the reference's GPU generator)
    static constexpr int MODEL = SNN_ST_POISSON;
    static PoissonNeuron from_firing_rate(float hertz, float dt)   // spike_train/mod.rs:327-334
    {
        PoissonNeuron p;
        p.dt = dt;
        p.chance_of_firing = 1.0f / ((1000.0f / p.dt) / hertz);
        return p;
    }
};
struct RateSpikeTrain : SpikeTrainBase {
    float rate = 0.0f, step = 0.0f;
    static constexpr int MODEL = SNN_ST_RATE;
};

// ---- plasticity (plasticity/mod.rs:16-39) -------------------------------------------------------
struct STDP {
    float a_plus = 2.0f, a_minus = 2.0f, tau_plus = 4.5f, tau_minus = 4.5f, dt = 0.1f;
};
struct BCM {                                   // plasticity/mod.rs:80-95
    float decay = 0.1f, average_scalar = 0.1f, dt = 0.1f;
};
struct RewardModulatedSTDP {                   // plasticity/mod.rs:158-190
    float dopamine = 0.0f, tau_d = 20.0f, tau_c = 0.0001f, a_plus = 2.0f, a_minus = 2.0f, tau_plus = 4.5f,
          tau_minus = 4.5f, dt = 0.1f;
};

// ---- AdjacencyMatrix (graph/mod.rs:139-297): dense Option<f32>, index = insertion order --------
struct AdjacencyMatrix {
    std::map<Position, size_t> position_to_index;
    std::vector<Position> index_to_position;
    std::vector<std::vector<std::optional<float>>> matrix;
    size_t id = 0;

    void add_node(Position p)
    {
        if (position_to_index.count(p)) return;
        position_to_index[p] = index_to_position.size();
        index_to_position.push_back(p);
        for (auto &row : matrix) row.push_back(std::nullopt);
        matrix.emplace_back(index_to_position.size(), std::nullopt);
    }
    std::optional<float> lookup_weight(Position pre, Position post) const
    {
        if (!position_to_index.count(post)) throw GraphError("PostsynapticNotFound");
        if (!position_to_index.count(pre)) throw GraphError("PresynapticNotFound");
        return matrix[position_to_index.at(pre)][position_to_index.at(post)];
    }
    void edit_weight(Position pre, Position post, std::optional<float> w)
    {
        if (!position_to_index.count(post)) throw GraphError("PostsynapticNotFound");
        if (!position_to_index.count(pre)) throw GraphError("PresynapticNotFound");
        matrix[position_to_index.at(pre)][position_to_index.at(post)] = w;
    }
    std::vector<Position> get_incoming_connections(Position pos) const
    {
        if (!position_to_index.count(pos)) throw GraphError("PositionNotFound");
        std::vector<Position> out;
        const size_t j = position_to_index.at(pos);
        for (size_t i = 0; i < matrix.size(); ++i) if (matrix[i][j]) out.push_back(index_to_position[i]);
        return out;
    }
    std::vector<Position> get_outgoing_connections(Position pos) const
    {
        if (!position_to_index.count(pos)) throw GraphError("PositionNotFound");
        std::vector<Position> out;
        const auto &row = matrix[position_to_index.at(pos)];
        for (size_t j = 0; j < row.size(); ++j) if (row[j]) out.push_back(index_to_position[j]);
        return out;
    }
};

// ---- Lattice (neuron/mod.rs:556-1157): the CPU-side container the GPU one is built from ---------
template <class T>
struct Lattice {
    std::vector<std::vector<T>> cell_grid;
    AdjacencyMatrix graph;
    bool electrical_synapse = true, chemical_synapse = false, do_plasticity = false;
    bool update_grid_history = false;
    STDP plasticity;
    size_t internal_clock = 0;

    size_t get_id() const { return graph.id; }
    void set_id(size_t id) { graph.id = id; }

    // populate (neuron/mod.rs:1105-1126): row-major insertion => index = row*cols + col
    void populate(const T &base_neuron, size_t num_rows, size_t num_cols)
    {
        const size_t id = graph.id;
        graph = AdjacencyMatrix();
        graph.id = id;
        cell_grid.assign(num_rows, std::vector<T>(num_cols, base_neuron));
        for (size_t i = 0; i < num_rows; ++i)
            for (size_t j = 0; j < num_cols; ++j) graph.add_node({i, j});
    }
    // connect (neuron/mod.rs:1134-1157): weight 1.0 when weight_logic is absent, None when not connected
    void connect(const std::function<bool(Position, Position)> &connecting_conditional,
                 const std::function<float(Position, Position)> &weight_logic = nullptr)
    {
        for (const auto &i : graph.index_to_position)
            for (const auto &j : graph.index_to_position) {
                if (connecting_conditional(i, j)) graph.edit_weight(i, j, weight_logic ? weight_logic(i, j) : 1.0f);
                else graph.edit_weight(i, j, std::nullopt);
            }
    }
    void apply(const std::function<void(T &)> &f)
    {
        for (auto &row : cell_grid) for (auto &c : row) f(c);
    }
    void apply_given_position(const std::function<void(Position, T &)> &f)
    {
        for (size_t i = 0; i < cell_grid.size(); ++i)
            for (size_t j = 0; j < cell_grid[i].size(); ++j) f({i, j}, cell_grid[i][j]);
    }
    void set_dt(float dt)          // neuron/mod.rs:649-652: neurons AND the lattice's plasticity
    {
        apply([dt](T &n) { n.dt = dt; });
        plasticity.dt = dt;
    }
    void reset_timing()            // neuron/mod.rs:405-420
    {
        internal_clock = 0;
        apply([](T &n) { n.last_firing_time.reset(); });
    }
    size_t rows() const { return cell_grid.size(); }
    size_t cols() const { return cell_grid.empty() ? 0 : cell_grid[0].size(); }
};

template <class S>
struct SpikeTrainLattice {          // neuron/mod.rs:1292-1436
    std::vector<std::vector<S>> cell_grid;
    size_t id = 0;
    bool update_grid_history = false;
    size_t internal_clock = 0;
    void set_id(size_t i) { id = i; }
    void populate(const S &base, size_t num_rows, size_t num_cols)
    {
        cell_grid.assign(num_rows, std::vector<S>(num_cols, base));
    }
    void apply(const std::function<void(S &)> &f)
    {
        for (auto &row : cell_grid) for (auto &c : row) f(c);
    }
    size_t rows() const { return cell_grid.size(); }
    size_t cols() const { return cell_grid.empty() ? 0 : cell_grid[0].size(); }
};

struct GraphPosition { size_t id; Position pos; };    // graph/mod.rs:24-30

// ---- AoS <-> named SoA buffers (IterateAndSpikeGPU::convert_to_gpu / convert_to_cpu) ------------
namespace detail {

template <class T, class Grid>
void upload_common(snn_network_t *h, uint32_t id, const Grid &g)
{
    const size_t n = g.size() * (g.empty() ? 0 : g[0].size());
    if (n == 0) return;
    std::vector<float> buf(n);
    std::vector<uint32_t> ubuf(n);
    std::vector<int32_t> ibuf(n);
    auto flat = [&](auto get, auto &dst) {
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) dst[k++] = get(c);
    };
    auto putf = [&](const char *name, auto get) { flat(get, buf); check(snn_set_attr_f32(h, id, name, buf.data(), n)); };
    putf("current_voltage", [](const T &c) { return c.current_voltage; });
    putf("dt", [](const T &c) { return c.dt; });
    flat([](const T &c) { return (uint32_t)c.is_spiking; }, ubuf);
    check(snn_set_attr_u32(h, id, "is_spiking", ubuf.data(), n));
    flat([](const T &c) { return c.last_firing_time ? (int32_t)*c.last_firing_time : -1; }, ibuf);
    check(snn_set_attr_i32(h, id, "last_firing_time", ibuf.data(), n));
    // neurotransmitters: [cell*3 + type]
    std::vector<float> t3(3 * n), tmax(3 * n, 1.0f), clr(3 * n, 0.01f), vp(3 * n, 2.0f), kp(3 * n, 5.0f);
    std::vector<uint32_t> fl(3 * n, 0);
    size_t k = 0;
    for (const auto &row : g)
        for (const auto &c : row) {
            for (const auto &[type, nt] : c.synaptic_neurotransmitters) {
                const size_t i = 3 * k + (size_t)type;
                fl[i] = 1; t3[i] = nt.t; tmax[i] = nt.t_max; clr[i] = nt.clearance_constant; vp[i] = nt.v_p; kp[i] = nt.k_p;
            }
            ++k;
        }
    check(snn_set_attr_u32(h, id, "neurotransmitters$flags", fl.data(), 3 * n));
    check(snn_set_attr_f32(h, id, "neurotransmitters$t", t3.data(), 3 * n));
    check(snn_set_attr_f32(h, id, "neurotransmitters$t_max", tmax.data(), 3 * n));
    check(snn_set_attr_f32(h, id, "neurotransmitters$clearance_constant", clr.data(), 3 * n));
    check(snn_set_attr_f32(h, id, "neurotransmitters$v_p", vp.data(), 3 * n));
    check(snn_set_attr_f32(h, id, "neurotransmitters$k_p", kp.data(), 3 * n));
}

template <class T>
void upload_neurons(snn_network_t *h, uint32_t id, const std::vector<std::vector<T>> &g)
{
    const size_t n = g.size() * (g.empty() ? 0 : g[0].size());
    if (n == 0) return;
    upload_common<T>(h, id, g);
    std::vector<float> buf(n);
    auto putf = [&](const char *name, auto get) {
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) buf[k++] = get(c);
        check(snn_set_attr_f32(h, id, name, buf.data(), n));
    };
    putf("gap_conductance", [](const T &c) { return c.gap_conductance; });
    putf("c_m", [](const T &c) { return c.c_m; });
    putf("v_th", [](const T &c) { return c.v_th; });
    for (const auto &[name, member] : T::fields()) putf(name, [member = member](const T &c) { return c.*member; });
    // receptors
    static const char *TN[3] = {"AMPA", "NMDA", "GABA"};
    std::vector<uint32_t> fl(3 * n, 0);
    for (int type = 0; type < 3; ++type) {
        const std::string p = std::string("receptors$") + TN[type];
        auto putr = [&](const std::string &name, float Receptor::*m) {
            size_t k = 0;
            for (const auto &row : g)
                for (const auto &c : row) {
                    auto it = c.receptors.find(type);
                    buf[k++] = it == c.receptors.end() ? Receptor::default_for(type).*m : it->second.*m;
                }
            check(snn_set_attr_f32(h, id, name.c_str(), buf.data(), n));
        };
        putr(p + "_g", &Receptor::g); putr(p + "_e", &Receptor::e); putr(p + "_current", &Receptor::current);
        putr(p + "$r$kinetics$r", &Receptor::r); putr(p + "$r$kinetics$alpha", &Receptor::alpha);
        putr(p + "$r$kinetics$beta", &Receptor::beta);
        if (type == NMDA) putr(p + "_mg", &Receptor::mg);
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) { fl[3 * k + type] = c.receptors.count(type) ? 1u : 0u; ++k; }
    }
    check(snn_set_attr_u32(h, id, "receptors$flags", fl.data(), 3 * n));
    if constexpr (T::MODEL == SNN_MODEL_HODGKIN_HUXLEY) {
        std::vector<uint32_t> wi(n);
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) wi[k++] = c.was_increasing;
        check(snn_set_attr_u32(h, id, "was_increasing", wi.data(), n));
    }
}

template <class T>
void download_neurons(snn_network_t *h, uint32_t id, std::vector<std::vector<T>> &g)
{
    const size_t n = g.size() * (g.empty() ? 0 : g[0].size());
    if (n == 0) return;
    std::vector<float> buf(n);
    std::vector<uint32_t> ubuf(n);
    std::vector<int32_t> ibuf(n);
    auto getf = [&](const char *name, auto set) {
        check(snn_get_attr_f32(h, id, name, buf.data(), n));
        size_t k = 0;
        for (auto &row : g) for (auto &c : row) set(c, buf[k++]);
    };
    getf("current_voltage", [](T &c, float v) { c.current_voltage = v; });
    for (const auto &[name, member] : T::fields()) getf(name, [member = member](T &c, float v) { c.*member = v; });
    check(snn_get_attr_u32(h, id, "is_spiking", ubuf.data(), n));
    check(snn_get_attr_i32(h, id, "last_firing_time", ibuf.data(), n));
    size_t k = 0;
    for (auto &row : g)
        for (auto &c : row) {
            c.is_spiking = ubuf[k] != 0;
            if (ibuf[k] < 0) c.last_firing_time.reset(); else c.last_firing_time = (size_t)ibuf[k];
            ++k;
        }
    std::vector<float> t3(3 * n);
    check(snn_get_attr_f32(h, id, "neurotransmitters$t", t3.data(), 3 * n));
    static const char *TN[3] = {"AMPA", "NMDA", "GABA"};
    for (int type = 0; type < 3; ++type) {
        const std::string p = std::string("receptors$") + TN[type];
        std::vector<float> r(n), cur(n);
        check(snn_get_attr_f32(h, id, (p + "$r$kinetics$r").c_str(), r.data(), n));
        check(snn_get_attr_f32(h, id, (p + "_current").c_str(), cur.data(), n));
        k = 0;
        for (auto &row : g)
            for (auto &c : row) {
                auto it = c.receptors.find(type);
                if (it != c.receptors.end()) { it->second.r = r[k]; it->second.current = cur[k]; }
                auto nt = c.synaptic_neurotransmitters.find(type);
                if (nt != c.synaptic_neurotransmitters.end()) nt->second.t = t3[3 * k + type];
                ++k;
            }
    }
    if constexpr (T::MODEL == SNN_MODEL_HODGKIN_HUXLEY) {
        std::vector<uint32_t> wi(n);
        check(snn_get_attr_u32(h, id, "was_increasing", wi.data(), n));
        k = 0;
        for (auto &row : g) for (auto &c : row) c.was_increasing = wi[k++] != 0;
    }
}

template <class S>
void upload_cells(snn_network_t *h, uint32_t id, const std::vector<std::vector<S>> &g)
{
    const size_t n = g.size() * (g.empty() ? 0 : g[0].size());
    if (n == 0) return;
    upload_common<S>(h, id, g);
    std::vector<float> buf(n);
    auto putf = [&](const char *name, auto get) {
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) buf[k++] = get(c);
        check(snn_set_attr_f32(h, id, name, buf.data(), n));
    };
    putf("v_th", [](const S &c) { return c.v_th; });
    putf("v_resting", [](const S &c) { return c.v_resting; });
    putf("neural_refractoriness$k", [](const S &c) { return c.neural_refractoriness_k; });
    if constexpr (S::MODEL == SNN_ST_POISSON) {
        putf("chance_of_firing", [](const S &c) { return c.chance_of_firing; });
        std::vector<uint32_t> seeds(n);
        size_t k = 0;
        for (const auto &row : g) for (const auto &c : row) seeds[k++] = c.seed;
        check(snn_set_attr_u32(h, id, "seed", seeds.data(), n));
    } else {
        putf("rate", [](const S &c) { return c.rate; });
        putf("step", [](const S &c) { return c.step; });
    }
}

} // namespace detail

// ---- LatticeNetwork + LatticeNetworkGPU ---------------------------------------------------------
template <class T, class S = RateSpikeTrain>
struct LatticeNetwork {                     // neuron/mod.rs:1538-1588
    std::map<size_t, Lattice<T>> lattices;
    std::map<size_t, SpikeTrainLattice<S>> spike_train_lattices;
    // connecting graph between lattices (GraphPosition keys), stored as edge list
    std::vector<std::tuple<GraphPosition, GraphPosition, float>> connecting_edges;
    bool electrical_synapse = true, chemical_synapse = false;
    size_t internal_clock = 0;

    void add_lattice(const Lattice<T> &l)   // neuron/mod.rs:1663-1679
    {
        if (lattices.count(l.get_id()) || spike_train_lattices.count(l.get_id()))
            throw LatticeNetworkError("GraphIDAlreadyPresent(" + std::to_string(l.get_id()) + ")");
        lattices[l.get_id()] = l;
    }
    void add_spike_train_lattice(const SpikeTrainLattice<S> &l)
    {
        if (lattices.count(l.id) || spike_train_lattices.count(l.id))
            throw LatticeNetworkError("GraphIDAlreadyPresent(" + std::to_string(l.id) + ")");
        spike_train_lattices[l.id] = l;
    }
    // connect (neuron/mod.rs:1845-1935): presynaptic may be a spike-train lattice, postsynaptic never
    void connect(size_t presynaptic_id, size_t postsynaptic_id,
                 const std::function<bool(Position, Position)> &connecting_conditional,
                 const std::function<float(Position, Position)> &weight_logic = nullptr)
    {
        if (spike_train_lattices.count(postsynaptic_id))
            throw LatticeNetworkError("PostsynapticLatticeCannotBeSpikeTrain");
        if (!lattices.count(postsynaptic_id)) throw LatticeNetworkError("PostsynapticIDNotFound");
        size_t pr = 0, pc = 0;
        if (lattices.count(presynaptic_id)) { pr = lattices[presynaptic_id].rows(); pc = lattices[presynaptic_id].cols(); }
        else if (spike_train_lattices.count(presynaptic_id)) {
            pr = spike_train_lattices[presynaptic_id].rows(); pc = spike_train_lattices[presynaptic_id].cols();
        } else throw LatticeNetworkError("PresynapticIDNotFound");
        const auto &post = lattices[postsynaptic_id];
        for (size_t a = 0; a < pr; ++a) for (size_t b = 0; b < pc; ++b)
            for (size_t c = 0; c < post.rows(); ++c) for (size_t d = 0; d < post.cols(); ++d)
                if (connecting_conditional({a, b}, {c, d}))
                    connecting_edges.emplace_back(GraphPosition{presynaptic_id, {a, b}}, GraphPosition{postsynaptic_id, {c, d}},
                                                  weight_logic ? weight_logic({a, b}, {c, d}) : 1.0f);
    }
    void set_dt(float dt)
    {
        for (auto &[id, l] : lattices) l.set_dt(dt);
        for (auto &[id, l] : spike_train_lattices) l.apply([dt](S &s) { s.dt = dt; });
    }
};

template <class T, class S = RateSpikeTrain>
class LatticeNetworkGPU {                    // neuron/gpu_lattices/mod.rs:1517-3212
public:
    LatticeNetwork<T, S> network;            // host copy, refreshed after every run_lattices
    std::map<size_t, std::vector<std::vector<std::vector<float>>>> grid_history;   // id -> [T][rows][cols]

    // from_network (gpu_lattices/mod.rs:1636-1651); kinetics are per network, as in the reference's type params
    static LatticeNetworkGPU from_network(const LatticeNetwork<T, S> &net, int nt_kinetics = SNN_NT_APPROXIMATE,
                                          int receptor_kinetics = SNN_RC_APPROXIMATE, int device = 0)
    {
        LatticeNetworkGPU g;
        g.network = net;
        const int st = net.spike_train_lattices.empty() ? SNN_ST_NONE : S::MODEL;
        check(snn_network_create(device, T::MODEL, nt_kinetics, receptor_kinetics, st, &g.h_));
        for (const auto &[id, l] : net.lattices) check(snn_network_add_lattice(g.h_, (uint32_t)id, (uint32_t)l.rows(), (uint32_t)l.cols()));
        for (const auto &[id, l] : net.spike_train_lattices)
            check(snn_network_add_spike_train_lattice(g.h_, (uint32_t)id, (uint32_t)l.rows(), (uint32_t)l.cols()));
        check(snn_network_finalize(g.h_));
        g.upload();
        // internal_clock: lattice_network.internal_clock (gpu_lattices/mod.rs:1630): a network that has run on the host goes on
        // at its clock -- the firing times uploaded above are absolute step numbers; each spike-train lattice keeps its own
        check(snn_set_clock(g.h_, (uint64_t)net.internal_clock));
        for (const auto &[id, l] : net.spike_train_lattices) check(snn_set_spike_train_clock(g.h_, (uint32_t)id, (uint64_t)l.internal_clock));
        return g;
    }
    LatticeNetworkGPU() = default;
    LatticeNetworkGPU(LatticeNetworkGPU &&o) noexcept { *this = std::move(o); }
    LatticeNetworkGPU &operator=(LatticeNetworkGPU &&o) noexcept
    {
        if (this != &o) { release(); h_ = o.h_; o.h_ = nullptr; network = std::move(o.network); grid_history = std::move(o.grid_history); }
        return *this;
    }
    LatticeNetworkGPU(const LatticeNetworkGPU &) = delete;
    LatticeNetworkGPU &operator=(const LatticeNetworkGPU &) = delete;
    ~LatticeNetworkGPU() { release(); }

    // RunNetwork::run_lattices (gpu_lattices/mod.rs:3183-3212): zero iterations / empty network => Ok(())
    void run_lattices(size_t iterations)
    {
        check(snn_set_synapses(h_, network.electrical_synapse, network.chemical_synapse));
        bool hist = false;
        for (const auto &[id, l] : network.lattices) hist |= l.update_grid_history;
        for (const auto &[id, l] : network.spike_train_lattices) hist |= l.update_grid_history;
        check(snn_set_history(h_, hist, 0));
        check(snn_run(h_, iterations));
        download();
    }
    snn_network_t *handle() const { return h_; }

  private:
    template <class V>
    std::vector<std::vector<V>> cross_rows(size_t pre, size_t post, bool counter)
    {
        uint32_t pf = 0, pc = 0, qf = 0, qc = 0, nn = 0;
        check(snn_network_lattice_range(h_, (uint32_t)pre, &pf, &pc));
        check(snn_network_lattice_range(h_, (uint32_t)post, &qf, &qc));
        check(snn_network_sizes(h_, &nn, nullptr, nullptr, nullptr));
        std::vector<std::vector<V>> out(pc, std::vector<V>(qc));
        if (counter) {
            std::vector<uint8_t> rows((size_t)pc * nn);
            check(snn_get_counter_rows(h_, pf, pc, rows.data()));
            for (uint32_t p = 0; p < pc; ++p)
                for (uint32_t q = 0; q < qc; ++q) out[p][q] = (V)rows[(size_t)p * nn + qf + q];
        } else {
            std::vector<float> rows((size_t)pc * nn);
            check(snn_get_pending_rows(h_, pf, pc, rows.data()));
            for (uint32_t p = 0; p < pc; ++p)
                for (uint32_t q = 0; q < qc; ++q) out[p][q] = (V)rows[(size_t)p * nn + qf + q];
        }
        return out;
    }

  public:

    // Reward modulation of lattice `id`'s internal edges (RewardModulatedLattice, neuron/mod.rs:2719-3417): the
    // lattice's weights become TraceRSTDP::weight, traces start at 0.  update_and_apply_reward = one step preceded by
    // RewardModulator::update(reward) (Agent, :3402-3407); results stay on the device until sync().
    void set_reward_modulator(size_t id, const RewardModulatedSTDP &m, bool do_modulation = true)
    {
        check(snn_set_reward_modulator(h_, (uint32_t)id, m.dopamine, m.tau_d, m.tau_c, m.a_plus, m.a_minus, m.tau_plus,
                                       m.tau_minus, m.dt, do_modulation));
    }
    void update_and_apply_reward(float reward)
    {
        check(snn_set_synapses(h_, network.electrical_synapse, network.chemical_synapse));
        check(snn_run_with_reward(h_, reward));
    }
    void sync() { download(); }
    float dopamine(size_t id)
    {
        float d = 0.0f;
        check(snn_get_dopamine(h_, (uint32_t)id, &d));
        return d;
    }
    // TraceRSTDP::c of lattice `id`'s internal edges, [n][n] presynaptic index first
    std::vector<std::vector<float>> traces(size_t id)
    {
        uint32_t first = 0, count = 0, nn = 0;
        check(snn_network_lattice_range(h_, (uint32_t)id, &first, &count));
        check(snn_network_sizes(h_, &nn, nullptr, nullptr, nullptr));
        std::vector<float> rows((size_t)count * nn);
        check(snn_get_trace_rows(h_, first, count, rows.data()));
        std::vector<std::vector<float>> out(count, std::vector<float>(count));
        for (uint32_t p = 0; p < count; ++p)
            for (uint32_t q = 0; q < count; ++q) out[p][q] = rows[(size_t)p * nn + first + q];
        return out;
    }
    // RewardModulatedLatticeNetwork (neuron/mod.rs:3419-3453): every connection from lattice (or spike-train lattice) `pre` into
    // lattice `post` is a RewardModulatedConnection -- reward_modulated = true: ::RewardModulatedWeight (the TraceRSTDP's c, dw and
    // counter live in the trace / pending / counter rows of the C ABI), false: ::Weight.  A configuration in which the
    // reference's visits would unwrap None (update_weights_from_neurons_across_lattices :4707-4802, _across_reward_lattices
    // :4855-4977) makes the next run throw with the reference line in the message.
    void set_connection(size_t pre, size_t post, bool reward_modulated)
    {
        check(snn_set_connection_kind(h_, (uint32_t)pre, (uint32_t)post, reward_modulated ? 1 : 2));
    }
    // TraceRSTDP::dw / ::counter of the connections from lattice `pre` into lattice `post`, [n_pre][n_post]
    std::vector<std::vector<float>> pending(size_t pre, size_t post) { return cross_rows<float>(pre, post, false); }
    std::vector<std::vector<float>> counters(size_t pre, size_t post) { return cross_rows<float>(pre, post, true); }

    // BCM rule for lattice `id` (needs BCMActivity neurons: SNN_MODEL_BCM_IZHIKEVICH)
    void set_bcm(size_t id, const BCM &b, bool do_plasticity = true)
    {
        check(snn_set_bcm(h_, (uint32_t)id, b.decay, b.average_scalar, b.dt, do_plasticity));
    }

    // Reduced histories kept on the device (the CPU lattices' AverageVoltageHistory / EEGHistory /
    // SpikeHistory::aggregate, neuron/mod.rs:233-360) and strided capture of every history row.
    void set_reduced_history(bool average_voltage, bool eeg, bool spike_counts, float reference_voltage = 0.007f,
                             float distance = 0.8f, float conductivity = 251.0f)
    {
        check(snn_set_reduced_history(h_, average_voltage, eeg, spike_counts, reference_voltage, distance, conductivity));
    }
    void set_history_stride(uint32_t every) { check(snn_set_history_stride(h_, every)); }
    std::vector<float> average_voltage_history(size_t id) { return reduced(id, false); }
    std::vector<float> eeg_history(size_t id) { return reduced(id, true); }
    std::vector<std::vector<uint32_t>> spike_counts(size_t id)
    {
        const auto &l = network.lattices.at(id);
        std::vector<uint32_t> flat(l.rows() * l.cols());
        check(snn_get_spike_counts(h_, (uint32_t)id, flat.data(), flat.size()));
        std::vector<std::vector<uint32_t>> out(l.rows(), std::vector<uint32_t>(l.cols()));
        for (size_t r = 0; r < l.rows(); ++r)
            for (size_t c = 0; c < l.cols(); ++c) out[r][c] = flat[r * l.cols() + c];
        return out;
    }

private:
    snn_network_t *h_ = nullptr;
    void release() { if (h_) { snn_network_destroy(h_); h_ = nullptr; } }

    std::vector<float> reduced(size_t id, bool eeg)
    {
        uint64_t steps = 0;
        check(snn_history_steps(h_, &steps));
        std::vector<float> out(steps);
        check((eeg ? snn_get_eeg_history : snn_get_average_voltage_history)(h_, (uint32_t)id, out.data(), out.size()));
        return out;
    }

    void upload()
    {
        uint32_t nn = 0, nc = 0;
        check(snn_network_sizes(h_, &nn, &nc, nullptr, nullptr));
        const size_t nt = (size_t)nn + nc;
        for (const auto &[id, l] : network.lattices) {
            detail::upload_neurons<T>(h_, (uint32_t)id, l.cell_grid);
            check(snn_set_plasticity(h_, (uint32_t)id, l.plasticity.a_plus, l.plasticity.a_minus, l.plasticity.tau_plus,
                                     l.plasticity.tau_minus, l.plasticity.dt, l.do_plasticity));
        }
        for (const auto &[id, l] : network.spike_train_lattices) detail::upload_cells<S>(h_, (uint32_t)id, l.cell_grid);
        if (nt == 0 || nn == 0) return;
        // InterleavingGraphGPU::convert_to_gpu (graph/mod.rs:644-807): internal graphs + connecting graph
        std::vector<float> w(nt * nn, 0.0f);
        std::vector<uint32_t> c(nt * nn, 0);
        for (const auto &[id, l] : network.lattices) {
            uint32_t first = 0, count = 0;
            check(snn_network_lattice_range(h_, (uint32_t)id, &first, &count));
            const size_t cols = l.cols();
            for (size_t i = 0; i < l.graph.matrix.size(); ++i)
                for (size_t j = 0; j < l.graph.matrix[i].size(); ++j)
                    if (l.graph.matrix[i][j]) {
                        const Position pi = l.graph.index_to_position[i], pj = l.graph.index_to_position[j];
                        const size_t gi = first + pi.first * cols + pi.second, gj = first + pj.first * cols + pj.second;
                        w[gi * nn + gj] = *l.graph.matrix[i][j];
                        c[gi * nn + gj] = 1;
                    }
        }
        for (const auto &[pre, post, weight] : network.connecting_edges) {
            uint32_t f0 = 0, c0 = 0, f1 = 0, c1 = 0;
            check(snn_network_lattice_range(h_, (uint32_t)pre.id, &f0, &c0));
            check(snn_network_lattice_range(h_, (uint32_t)post.id, &f1, &c1));
            const size_t pc = network.lattices.count(pre.id) ? network.lattices.at(pre.id).cols()
                                                             : network.spike_train_lattices.at(pre.id).cols();
            const size_t qc = network.lattices.at(post.id).cols();
            const size_t gi = f0 + pre.pos.first * pc + pre.pos.second, gj = f1 + post.pos.first * qc + post.pos.second;
            w[gi * nn + gj] = weight;
            c[gi * nn + gj] = 1;
        }
        check(snn_set_graph_rows(h_, 0, (uint32_t)nt, w.data(), c.data()));
    }

    void download()
    {
        uint64_t clock = 0, steps = 0;
        check(snn_get_clock(h_, &clock));
        check(snn_history_steps(h_, &steps));
        network.internal_clock = (size_t)clock;
        uint32_t nn = 0, nc = 0;
        check(snn_network_sizes(h_, &nn, &nc, nullptr, nullptr));
        const size_t nt = (size_t)nn + nc;
        std::vector<float> w;
        std::vector<uint32_t> c;
        if (nt && nn) {
            w.resize(nt * nn); c.resize(nt * nn);
            check(snn_get_graph_rows(h_, 0, (uint32_t)nt, w.data(), c.data()));
        }
        for (auto &[id, l] : network.lattices) {
            detail::download_neurons<T>(h_, (uint32_t)id, l.cell_grid);
            l.internal_clock = (size_t)clock;
            uint32_t first = 0, count = 0;
            check(snn_network_lattice_range(h_, (uint32_t)id, &first, &count));
            const size_t cols = l.cols();
            for (size_t i = 0; i < l.graph.matrix.size(); ++i)          // weights back into the AdjacencyMatrix
                for (size_t j = 0; j < l.graph.matrix[i].size(); ++j) {
                    const Position pi = l.graph.index_to_position[i], pj = l.graph.index_to_position[j];
                    const size_t gi = first + pi.first * cols + pi.second, gj = first + pj.first * cols + pj.second;
                    if (c[gi * nn + gj]) l.graph.matrix[i][j] = w[gi * nn + gj]; else l.graph.matrix[i][j].reset();
                }
            if (l.update_grid_history && steps && count) {
                std::vector<float> hist(steps * count);
                check(snn_get_voltage_history(h_, (uint32_t)id, hist.data(), hist.size()));
                auto &out = grid_history[id];
                out.assign(steps, std::vector<std::vector<float>>(l.rows(), std::vector<float>(cols)));
                for (size_t s = 0; s < steps; ++s)
                    for (size_t r = 0; r < l.rows(); ++r)
                        for (size_t q = 0; q < cols; ++q) out[s][r][q] = hist[s * count + r * cols + q];
            }
        }
        for (auto &e : network.connecting_edges) {
            auto &[pre, post, weight] = e;
            uint32_t f0 = 0, c0 = 0, f1 = 0, c1 = 0;
            check(snn_network_lattice_range(h_, (uint32_t)pre.id, &f0, &c0));
            check(snn_network_lattice_range(h_, (uint32_t)post.id, &f1, &c1));
            const size_t pc = network.lattices.count(pre.id) ? network.lattices.at(pre.id).cols()
                                                             : network.spike_train_lattices.at(pre.id).cols();
            const size_t qc = network.lattices.at(post.id).cols();
            weight = w[(f0 + pre.pos.first * pc + pre.pos.second) * nn + f1 + post.pos.first * qc + post.pos.second];
        }
    }
};

// ---- LatticeGPU: a single lattice is a network of one (gpu_lattices/mod.rs:327-1118) -----------
template <class T>
class LatticeGPU {
public:
    static LatticeGPU from_lattice(const Lattice<T> &lattice, int nt_kinetics = SNN_NT_APPROXIMATE,
                                   int receptor_kinetics = SNN_RC_APPROXIMATE, int device = 0)
    {
        LatticeNetwork<T> net;
        net.add_lattice(lattice);
        net.electrical_synapse = lattice.electrical_synapse;
        net.chemical_synapse = lattice.chemical_synapse;
        LatticeGPU g;
        g.id_ = lattice.get_id();
        g.net_ = LatticeNetworkGPU<T>::from_network(net, nt_kinetics, receptor_kinetics, device);
        return g;
    }
    // RunLattice::run_lattice (gpu_lattices/mod.rs:1081-1100)
    void run_lattice(size_t iterations) { net_.run_lattices(iterations); }
    Lattice<T> &lattice() { return net_.network.lattices.at(id_); }
    std::vector<std::vector<T>> &cell_grid() { return lattice().cell_grid; }
    const std::vector<std::vector<std::vector<float>>> &grid_history() { return net_.grid_history[id_]; }
    bool &electrical_synapse() { return net_.network.electrical_synapse; }
    bool &chemical_synapse() { return net_.network.chemical_synapse; }
    size_t internal_clock() const { return net_.network.internal_clock; }
    snn_network_t *handle() const { return net_.handle(); }

private:
    size_t id_ = 0;
    LatticeNetworkGPU<T> net_;
};

} // namespace snn_host

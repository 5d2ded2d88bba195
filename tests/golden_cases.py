"""Golden cases shared by tests/golden/make_golden.py (writes the fixtures with the numpy restatement
tests/numpy_net.py), the CPU tests that pin the C oracle and that restatement to them, and the GPU test that holds the
HIP stepper to the SAME committed vectors.  The reference holds no golden vectors for this path (every test draws from
thread_rng), so these are produced here -- by the numpy restatement with the host libm, as SURVEY
section 8c specifies -- and each vector therefore pins two independently written CPU implementations.  The builders below
only lay out INPUT arrays (in an oracle-binding container, which also serves as the oracle's input)."""
import numpy as np

import oracle_binding as ob
import parity


def izh_4x4_ones():
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 16, -65.0, 30.0)
    net.connect_all_to_all(1.0)
    return net, 600


def izh_4x4_random():
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 16, -65.0, 30.0)
    net.fill_graph(2, 0.5, 1.5)
    return net, 600


def izh_32x32_random():
    net = parity.make_oracle(parity.Layout([(0, 32, 32)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 1024, -65.0, 30.0)
    net.fill_graph(2, 0.5, 1.5)
    return net, 400


def stdp_3_neurons():
    net = parity.make_oracle(parity.Layout([(0, 1, 3)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = np.array([25.0, -40.0, 5.0], np.float32)
    net["a"] = np.array([0.02, 0.1, 0.05], np.float32)
    net.connect_all_to_all(1.0)
    net["do_plasticity"] = 1
    return net, 1500


def hh_pair():
    net = parity.make_oracle(parity.Layout([(0, 1, 2)]), model=ob.HH)
    net["current_voltage"] = np.array([-65.0, -20.0], np.float32)
    net["gap_conductance"] = np.array([0.5, 0.1], np.float32)
    net.connect_all_to_all(1.0)
    return net, 2000


def ampa_pair():
    net = parity.make_oracle(parity.Layout([(0, 1, 2)]), chemical=True)
    net["current_voltage"] = np.array([29.5, -65.0], np.float32)
    net["gap_conductance"] = 4.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 2.0
    net.connect_all_to_all(1.0)
    return net, 1200


def spike_trains_poisson():
    net = parity.make_oracle(parity.Layout([(1, 1, 1)], [(0, 4, 4)]), st_kind=ob.ST_POISSON)
    net["st_seed"] = np.arange(1, 17, dtype=np.uint32)                 # seeds 1..16
    net["st_chance_of_firing"] = 0.02
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = 1.0
    return net, 1000


def spike_trains_rate():
    net = parity.make_oracle(parity.Layout([(1, 1, 1)], [(0, 2, 3)]), st_kind=ob.ST_RATE)
    net["st_rate"] = np.array([0.0, 1.0, 2.5, 3.0, 7.7, 10.0], np.float32)
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = 1.0
    return net, 1000


def adaptive_exp_lif_3x3():
    """AdaptiveExpLeakyIntegrateAndFireNeuron (integrate_and_fire/mod.rs:1051-1155), heterogeneous slope / beta"""
    net = parity.make_oracle(parity.Layout([(0, 3, 3)]), model=ob.ADAPTIVE_EXP_LIF)
    net["current_voltage"] = ob.uniform_array(3, 9, -75.0, -56.0)
    net["gap_conductance"] = 3.0
    net["leak_constant"] = 1.0
    net["c_m"] = 1.0
    net["v_reset"] = -73.0
    net["tref"] = 0.7
    net["adp_beta"] = ob.uniform_array(4, 9, 1.0, 4.0)
    net["slope_factor"] = ob.uniform_array(5, 9, 0.5, 3.0)
    net.fill_graph(6, 0.5, 1.5)
    return net, 600


def leaky_izhikevich_3x3():
    """LeakyIzhikevichNeuron (integrate_and_fire/mod.rs:1270-1356)"""
    net = parity.make_oracle(parity.Layout([(0, 3, 3)]), model=ob.LEAKY_IZHIKEVICH)
    net["current_voltage"] = ob.uniform_array(7, 9, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["w_value"] = 0.5
    net.fill_graph(8, 0.5, 1.5)
    return net, 600


def preset_exponential_decay_kinetics():
    """PresetSpikeTrain cells (spike_train/mod.rs:753-833, one with ExponentialDecayRefractoriness :164-178) into two
    neurons with ExponentialDecay neurotransmitter / receptor kinetics (iterate_and_spike/mod.rs:323-366, 497-533)"""
    net = parity.make_oracle(parity.Layout([(1, 1, 2)], [(0, 1, 3)]), st_kind=ob.ST_PRESET,
                             nt_kind=ob.NT_EXPONENTIAL_DECAY, rc_kind=ob.RC_EXPONENTIAL_DECAY, chemical=True)
    net["gap_conductance"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["nt_clearance"][:, 0] = 3.0
    net["st_nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 3.0
    net["rc_beta"][:, 0] = 1.5
    net["st_refractoriness"][1] = 1
    net["st_k"][1] = 200.0
    net.set_firing_times([[5.0], [6.0, 2.5], [1.5, 2.0, 4.0]])
    net["connections"][0, 1] = net["connections"][1, 0] = 1
    net["weights"][0, 1] = net["weights"][1, 0] = 1.0
    net["connections"][2:, :] = 1
    net["weights"][2:, :] = 1.5
    return net, 1200


def reward_modulated_4x4():
    """RewardModulatedLattice + RewardModulatedSTDP / TraceRSTDP (neuron/mod.rs:2719-3417, plasticity/mod.rs:126-242),
    constant dopamine (run without reward), deferred form"""
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]))
    net["current_voltage"] = ob.uniform_array(1, 16, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net.fill_graph(2, 0.5, 1.5)
    net["rm_do_modulation"] = 1
    net["rm_dopamine"] = 0.02
    net["rm_tau_c"] = 0.05
    net["rm_a_plus"] = 0.01
    net["rm_a_minus"] = 0.01
    net["traces"][...] = 0.001 * net["connections"]
    return net, 900


def reward_modulated_network():
    """RewardModulatedLatticeNetwork (neuron/mod.rs:3419-3453), every connection between lattices a RewardModulatedConnection: a
    plastic plain lattice (id 0), two reward-modulated lattices (ids 1, 3), a plain lattice without plasticity (id 2) and Poisson
    cells.  RewardModulatedWeight between 0, 1 and 3 (and from the cells into 1 and 3), Weight everywhere else; all connections
    exist in both directions, which the outgoing halves of update_weights_from_neurons_across_lattices (:4707-4802) and
    _across_reward_lattices (:4855-4977) need (they look up the reverse connection).  Constant dopamine."""
    lay = parity.Layout([(0, 3, 3), (1, 3, 4), (2, 2, 3), (3, 2, 2)], [(5, 2, 3)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON)
    nn = net.n_neurons
    net["current_voltage"] = ob.uniform_array(21, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net.fill_graph(22, 0.5, 1.5)
    net["st_chance_of_firing"] = 0.15
    net["st_seed"] = np.arange(31, 37, dtype=np.uint32)
    for l, dopamine in ((1, 0.015), (3, -0.01)):
        net["rm_do_modulation"][l] = 1
        net["rm_dopamine"][l] = dopamine
        net["rm_tau_c"][l] = 0.05
        net["rm_a_plus"][l] = 0.01
        net["rm_a_minus"][l] = 0.012
    net["do_plasticity"][0] = 1
    net["stdp_a_plus"][0] = 0.8
    net["stdp_a_minus"][2] = 0.5
    net["conn_kind"][...] = 2           # RewardModulatedConnection::Weight ...
    for a, b in ((0, 1), (0, 3), (1, 3)):
        net["conn_kind"][a, b] = net["conn_kind"][b, a] = 1          # ... RewardModulatedWeight among 0, 1 and 3
    net["conn_kind"][4, 1] = net["conn_kind"][4, 3] = 1              # and from the cells (source slot n_lattices + 0) into 1 and 3
    net["conn_kind"][np.arange(4), np.arange(4)] = 0                 # (a lattice's own edges follow its own rule)
    net["traces"][...] = ob.uniform_array(23, net["traces"].size, -0.01, 0.01).reshape(net["traces"].shape) * net["connections"]
    assert net.reward_cross_check() == 0
    return net, 701                     # (an odd count: the two-visit cycle of the traces ends half way)


def generated_morris_lecar_3x3():
    """a generated model (SNN_MODEL_CUSTOM): the three-channel Morris-Lecar description of test_modelgen_channels.py in a
    3x3 lattice driven by two Poisson cells; the oracle steps it as a stack program, the device as generated HIP"""
    import modelgen_ref
    from snn_amd import modelgen
    from test_modelgen_channels import MORRIS_LECAR
    model = modelgen.parse(MORRIS_LECAR)
    net = parity.make_oracle(parity.Layout([(0, 3, 3)], [(1, 1, 2)]), model=ob.CUSTOM, st_kind=ob.ST_POISSON)
    modelgen_ref.attach(net, model)
    names = [n for n, _ in model.variables]
    net["current_voltage"] = ob.uniform_array(11, 9, -70.0, -20.0)
    net["gap_conductance"] = 1.0
    net["custom_vars"][names.index("k_channel$phi")] = ob.uniform_array(12, 9, 0.04, 0.09)
    net["st_chance_of_firing"] = np.array([0.02, 0.05], np.float32)
    net["st_seed"] = np.array([77, 78], np.uint32)
    net.fill_graph(13, 0.5, 1.5)
    net["do_plasticity"] = 1
    return net, 2400


CASES = {f.__name__: f for f in (izh_4x4_ones, izh_4x4_random, izh_32x32_random, stdp_3_neurons, hh_pair, ampa_pair,
                                 spike_trains_poisson, spike_trains_rate, adaptive_exp_lif_3x3, leaky_izhikevich_3x3,
                                 preset_exponential_decay_kinetics, reward_modulated_4x4, reward_modulated_network,
                                 generated_morris_lecar_3x3)}

EXTRA = {"hh_pair": ("m_state", "h_state", "n_state"), "stdp_3_neurons": ("weights",),
         "ampa_pair": ("nt_t", "rc_r", "rc_current"), "spike_trains_poisson": ("st_seed", "st_last_firing_time"),
         "spike_trains_rate": ("st_step", "st_last_firing_time"),
         "adaptive_exp_lif_3x3": ("w_value", "refractory_count"), "leaky_izhikevich_3x3": ("w_value",),
         "preset_exponential_decay_kinetics": ("nt_t", "rc_r", "st_step", "st_counter", "st_last_firing_time"),
         "reward_modulated_4x4": ("weights", "traces"), "reward_modulated_network": ("weights", "traces", "pending", "edge_counter"), "generated_morris_lecar_3x3": ("custom_vars", "weights")}


def outputs(name, net, steps):
    """What a fixture stores: raster (bit-packed), final state, and the voltage trace (full for small cases,
    every 8th step for 32x32)."""
    stride = 8 if net.n_neurons > 64 else 1
    out = {"steps": np.int64(steps), "trace_stride": np.int64(stride),
           "raster": np.packbits(net.spike_history, axis=1),
           "voltage_trace": net.voltage_history[::stride].copy(),
           "final_voltage": net["current_voltage"].copy(),
           "last_firing_time": net["last_firing_time"].copy()}
    if net.n_cells:
        out["st_voltage_spikes"] = np.packbits(net.st_voltage_history > 0, axis=1)
    for k in EXTRA.get(name, ()):
        out[k] = net[k].copy()
    return out

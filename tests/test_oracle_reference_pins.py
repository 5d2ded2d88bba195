"""Pin the oracle to every known-answer / behavioural test the reference holds for the hot path
 and to hand-derived values of each formula.  The reference has no golden vectors and its
own CPU sums run in HashSet order, so bit-level parity with the Rust crate is unpinned; these are the
anchors that exist.  Citations relative to /root/reference/backend/."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

f32 = np.float32


# ---- tests/rate_spike_train.rs:28-72 -------------------------------------------------------------
@pytest.mark.parametrize("rate", [0, 100, 200, 300, 400, 500])
def test_rate_spike_train_expected_rate(rate):
    """test_expected_rate: 10 000 iterations at the default dt = 0.1 fire ITER/(rate/0.1) +- 1 times."""
    net = ob.Net(0, n_cells=1, st_kind=ob.ST_RATE)
    net["st_rate"] = rate
    spikes = 0
    for _ in range(10_000):
        net.spike_trains()
        spikes += int(net["st_is_spiking"][0])
    if rate == 0:
        assert spikes == 0
    else:
        assert abs(spikes - 10_000 / (rate / 0.1)) <= 1.0


def test_rate_spike_train_spacing():
    """test_spacing: rate = 100, dt = 1 fires exactly when (i + 1) % 100 == 0 (and never at i == 0)."""
    net = ob.Net(0, n_cells=1, st_kind=ob.ST_RATE)
    net["st_rate"] = 100.0
    net["st_dt"] = 1.0
    for i in range(1001):
        net.spike_trains()
        expect = not (i == 0 or (i + 1) % 100 != 0)
        assert bool(net["st_is_spiking"][0]) == expect, i
    # last_firing_time is stamped with the spike-train lattice's own clock (src/neuron/mod.rs:1377-1393)
    assert net["st_last_firing_time"][0] == 999
    assert net["st_clock"][0] == 1001


# ---- src/graph/mod.rs:113-137 (doc-test) ---------------------------------------------------------
def test_none_is_not_some_zero_in_the_averager():
    """lookup_weight distinguishes None from Some(w); an edge Some(0.0) still counts as an input
    (src/neuron/mod.rs:722-727 divides by input_positions.len())."""
    net = ob.Net(3)
    net["current_voltage"] = np.array([-60.0, -50.0, -40.0], f32)
    net["gap_conductance"] = 2.0
    # edges into neuron 1: from 0 with w = 0.5, from 2 with w = Some(0.0)
    net["connections"][0, 1] = 1
    net["weights"][0, 1] = 0.5
    net["connections"][2, 1] = 1
    net["weights"][2, 1] = 0.0
    net.inputs()
    expect = (f32(2.0) * (f32(-60.0) - f32(-50.0))) * f32(0.5) + (f32(2.0) * (f32(-40.0) - f32(-50.0))) * f32(0.0)
    assert net["input_current"][1] == f32(expect) / f32(2.0)          # two inputs, not one
    # removing the zero-weight edge (edit_weight(.., None)) changes the averager
    net["connections"][2, 1] = 0
    net.inputs()
    assert net["input_current"][1] == f32(-10.0)
    assert net["input_current"][0] == 0.0 and net["input_current"][2] == 0.0   # no inputs -> 0 / 1


# ---- tests/size_zero_cases.rs ----------------------------------------------------------------------
def test_zero_size_lattice_is_a_noop():
    net = ob.Net(0)
    net.run(25)
    assert net.clock == 25          # CPU run_lattice still counts steps on an empty grid (mod.rs:979)
    net = ob.Net(4)
    net.electrical = net.chemical = False
    before = net["current_voltage"].copy()
    net.run(10)
    assert net.clock == 0 and np.array_equal(before, net["current_voltage"])   # (false,false) => Ok(())


# ---- tests/spike_train_neuron_interaction.rs:91-203 ------------------------------------------------
def _poisson_to_neuron(model, synapses, dt, iterations, hh=False):
    lay = parity.Layout([(1, 1, 1)], [(0, 1, 1)])
    net = parity.make_oracle(lay, model=model, st_kind=ob.ST_POISSON, electrical=synapses[0], chemical=synapses[1])
    net["gap_conductance"] = 7.0 if hh else 10.0
    net["nt_flags"][:, 0] = 1
    net["st_nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["dt"] = dt
    net["st_dt"] = dt
    net["stdp_dt"] = dt
    net["connections"][1, 0] = 1        # spike-train cell (index 1) -> neuron 0, x == y, weight 1
    net["weights"][1, 0] = 1.0
    net["st_seed"] = 12345
    net.run(iterations, spike_history=True)
    first = int(net.spike_history.sum())
    net["st_chance_of_firing"] = (dt / 1.0) * 0.01
    net.run(iterations, spike_history=True)
    return first, int(net.spike_history.sum())


def test_poisson_drives_izhikevich_through_gap_junction():
    first, second = _poisson_to_neuron(ob.IZHIKEVICH, (True, False), 1.0, 2500)
    assert first <= 1 and second > 2


def test_poisson_drives_izhikevich_through_ampa():
    first, second = _poisson_to_neuron(ob.IZHIKEVICH, (False, True), 1.0, 2500)
    assert first <= 1 and second > 2


def test_poisson_drives_hodgkin_huxley_through_ampa():
    first, second = _poisson_to_neuron(ob.HH, (False, True), 0.01, 100_000, hh=True)
    assert first <= 1 and second > 2


# ---- hand-derived single-step values -----------------------------------------------------------------
def test_izhikevich_default_first_step():
    """default neuron, I = 0: dv = (((0.04*65*65 - 325) + 140) - 30 + 0) * (0.1/100)."""
    net = ob.Net(1)
    net.run(1)
    v, w = f32(-65.0), f32(30.0)
    dv = ((((f32(0.04) * (v * v)) + f32(5.0) * v) + f32(140.0)) - w + f32(0.0)) * (f32(0.1) / f32(100.0))
    dw = (f32(0.02) * (f32(0.2) * v - w)) * (f32(0.1) / f32(1.0))
    assert net["current_voltage"][0] == v + dv
    assert net["w_value"][0] == w + dw
    assert net["is_spiking"][0] == 0 and net["last_firing_time"][0] == -1


def test_izhikevich_spike_reset_and_stamp():
    net = ob.Net(1)
    net["current_voltage"] = 29.99
    net.clock = 41
    net.run(1)
    assert net["is_spiking"][0] == 1 and net["current_voltage"][0] == f32(-55.0)
    assert net["last_firing_time"][0] == 41 and net.clock == 42


def test_lif_step_and_refractory_period():
    """integrate_and_fire/mod.rs:87-102, 173-179: default LIF with input 300 crosses threshold, then
    stays at v_reset for tref/dt = 100 steps."""
    net = ob.Net(1, model=ob.LIF)
    v = f32(-75.0)
    net["connections"][0, 0] = 1          # self-edge contributes g*(V-V)*w = 0 but exercises the averager
    net["weights"][0, 0] = 1.0
    net.run(1)
    dv = ((f32(-1.0) * (v - f32(-75.0))) + (f32(1.0) * (f32(0.0) / f32(10.0)))) * (f32(0.1) / f32(10.0))
    assert net["current_voltage"][0] == v + dv
    net["current_voltage"] = -54.0
    net.run(1)
    assert net["is_spiking"][0] == 1 and net["refractory_count"][0] == f32(10.0) / f32(0.1)
    net.run(100)
    assert net["refractory_count"][0] == 0.0 and net["current_voltage"][0] == f32(-75.0)


def test_stdp_delta_known_answers():
    """plasticity/mod.rs:45-66 with defaults (2, 2, 4.5, 4.5, 0.1): (tp,tq) = (10,15) -> 2*exp(-|-5*0.1|/4.5)."""
    L = ob.lib()
    d = L.snn_o_stdp_delta(10, 15, 2.0, 2.0, 4.5, 4.5, 0.1)
    assert d == f32(2.0) * f32(ob.expf(f32(-1.0) * abs((f32(10.0) - f32(15.0)) * f32(0.1)) / f32(4.5)))
    assert abs(d - 2.0 * np.exp(-0.5 / 4.5)) < 1e-6
    d = L.snn_o_stdp_delta(15, 10, 2.0, 2.0, 4.5, 4.5, 0.1)
    assert abs(d + 2.0 * np.exp(-0.5 / 4.5)) < 1e-6
    assert L.snn_o_stdp_delta(7, 7, 2.0, 2.0, 4.5, 4.5, 0.1) == 0.0
    assert L.snn_o_stdp_delta(-1, 7, 2.0, 2.0, 4.5, 4.5, 0.1) == 0.0      # pre never fired
    assert L.snn_o_stdp_delta(7, -1, 2.0, 2.0, 4.5, 4.5, 0.1) == 0.0


def test_delta_dirac_and_spike_train_gap_junction():
    """spike_train/mod.rs:84-86, neuron/mod.rs:119-137."""
    L = ob.lib()
    e = L.snn_o_delta_dirac_effect(12, 10, 30.0, 0.0, 10000.0, 0.1)
    assert abs(e - 30.0 * np.exp(-(1.0 / (10000.0 / 0.1)) * 4.0)) < 1e-5
    lay = parity.Layout([(1, 1, 1)], [(0, 1, 1)])
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE)
    net["connections"][1, 0] = 1
    net["weights"][1, 0] = 0.5
    net["st_v_resting"] = -3.0
    net.inputs()
    assert net["input_current"][0] == f32(-3.0) * f32(0.5)          # never fired: v_resting, no conductance
    net["st_last_firing_time"] = 10
    net.clock = 12
    net.inputs()
    eff = L.snn_o_delta_dirac_effect(12, 10, 30.0, -3.0, 10000.0, 0.1)
    assert net["input_current"][0] == (f32(7.0) * f32(eff)) * f32(0.5)


def test_xorshift32_and_poisson_threshold():
    """spike_train/mod.rs:380-388, 419-426."""
    L = ob.lib()
    x = 1
    for _ in range(3):
        x = L.snn_o_xorshift32(x)
    ref = 1
    for _ in range(3):
        ref ^= (ref << 13) & 0xFFFFFFFF
        ref ^= ref >> 17
        ref ^= (ref << 5) & 0xFFFFFFFF
    assert x == ref
    net = ob.Net(0, n_cells=4, st_kind=ob.ST_POISSON)
    net["st_chance_of_firing"] = 0.25
    seeds = net["st_seed"].copy()
    net.spike_trains()
    for s in range(4):
        ns = L.snn_o_xorshift32(int(seeds[s]))
        assert net["st_seed"][s] == ns
        assert bool(net["st_is_spiking"][s]) == (f32(ns) / f32(4294967296.0) < f32(0.25))
        assert net["st_current_voltage"][s] == (30.0 if net["st_is_spiking"][s] else 0.0)


def test_approximate_neurotransmitter_and_receptor_currents():
    """iterate_and_spike/mod.rs:193-196 (t uses the PREVIOUS step's is_spiking), :1103-1105, :1132-1137,
    :1286-1304."""
    net = ob.Net(1, chemical=True, electrical=False)
    net["nt_flags"][0, :] = 1
    net["nt_t"][0, :] = 0.5
    net["is_spiking"] = 1
    net.run(1)
    t = f32(0.5) + (f32(0.1) * f32(-0.01) * f32(0.5) + f32(1.0) * f32(1.0))
    assert np.all(net["nt_t"][0] == min(f32(1.0), max(t, f32(0.0))))
    # receptor currents from a hand-set gating value
    net = ob.Net(1, chemical=True, electrical=False)
    net["rc_flags"][0, :] = 1
    net["rc_r"][0, :] = 0.25
    v = f32(-65.0)
    net.run(1)
    i_ampa = (f32(1.0) * f32(0.25)) * (v - f32(0.0))
    i_nmda = ((f32(1.0) / (f32(1.0) + ((f32(ob.expf(f32(-0.062) * v)) * f32(0.3)) / f32(3.75))) * f32(0.6)) * f32(0.25)) * (v - f32(0.0))
    i_gaba = (f32(1.2) * f32(0.25)) * (v - f32(-80.0))
    assert net["rc_current"][0, 0] == i_ampa and net["rc_current"][0, 1] == i_nmda and net["rc_current"][0, 2] == i_gaba
    total = ((f32(0.0) + i_ampa) + i_nmda) + i_gaba
    dv = (f32(0.04) * (v * v) + f32(5.0) * v + f32(140.0) - f32(30.0) + f32(0.0)) * (f32(0.1) / f32(100.0))
    assert net["current_voltage"][0] == v + (dv + -(total * (f32(0.1) / f32(100.0))))


def test_hodgkin_huxley_first_step():
    """hodgkin_huxley/mod.rs:156-201, ion_channels/mod.rs:40-44, 219-236, 267-282, 309-312 at V = -65,
    gates starting from 0 (the reference never calls init_state)."""
    net = ob.Net(1, model=ob.HH)
    net["connections"][0, 0] = 0
    net.run(1)
    e = lambda x: f32(ob.expf(f32(x)))
    v, dt = f32(-65.0), f32(0.01)
    m_a = f32(0.1) * ((v + f32(40.0)) / (f32(1.0) - e(-(v + f32(40.0)) / f32(10.0))))
    m_b = f32(4.0) * e(-(v + f32(65.0)) / f32(18.0))
    h_a = f32(0.07) * e(-(v + f32(65.0)) / f32(20.0))
    h_b = f32(1.0) / (e(-(v + f32(35.0)) / f32(10.0)) + f32(1.0))
    n_a = f32(0.01) * (v + f32(55.0)) / (f32(1.0) - e(-(v + f32(55.0)) / f32(10.0)))
    n_b = f32(0.125) * e(-(v + f32(65.0)) / f32(80.0))
    m = f32(0.0) + dt * (m_a * (f32(1.0) - f32(0.0)) - m_b * f32(0.0))
    h = f32(0.0) + dt * (h_a * f32(1.0) - h_b * f32(0.0))
    n = f32(0.0) + dt * (n_a * f32(1.0) - n_b * f32(0.0))
    assert net["m_state"][0] == m and net["h_state"][0] == h and net["n_state"][0] == n
    i_na = f32(np.float64(m) ** 3) * h * f32(120.0) * (v - f32(50.0))
    i_k = f32(np.float64(n) ** 4) * f32(36.0) * (v - f32(-77.0))
    i_l = f32(0.3) * (v - f32(-55.0))
    assert net["na_current"][0] == i_na and net["k_current"][0] == i_k and net["k_leak_current"][0] == i_l
    i_sum = f32(0.0) - (i_na + i_k + i_l)
    assert net["current_voltage"][0] == v + (dt * i_sum / f32(1.0) - f32(0.0))


def test_hodgkin_huxley_fires_under_constant_drive():
    """Behavioural: a constant suprathreshold input makes the default HH neuron spike repeatedly
    (peak detection, hodgkin_huxley/mod.rs:207-220)."""
    lay = parity.Layout([(0, 1, 2)])
    net = parity.make_oracle(lay, model=ob.HH)
    # neuron 1 is clamped far above neuron 0 through a strong gap junction -> sustained drive
    net["connections"][1, 0] = 1
    net["weights"][1, 0] = 1.0
    net["gap_conductance"][0] = 0.5
    net["current_voltage"][1] = 0.0
    net["g_na"][1] = 0.0
    net["g_k"][1] = 0.0
    net["g_k_leak"][1] = 0.0            # neuron 1 has no currents: its voltage stays at 0 mV
    net.run(20_000, spike_history=True)
    assert net.spike_history[:, 0].sum() >= 3
    assert net["current_voltage"][1] == 0.0


def test_deferred_plasticity_is_applied_per_incident_edge():
    """neuron/mod.rs:2308-2417, 2573-2576: when neuron j spikes, every incoming edge (p,j) gets
    stdp(lft[p], t) and every outgoing edge (j,r) gets stdp(t, lft[r]); both-spiking pairs get +0."""
    net = ob.Net(3)
    net.connect_all_to_all(1.0)
    net["do_plasticity"] = 1
    net["last_firing_time"] = np.array([5, -1, 8], np.int32)
    net["current_voltage"] = np.array([-65.0, 29.999, -65.0], f32)   # only neuron 1 will spike
    net["gap_conductance"] = 0.0
    net.clock = 10
    net.run(1)
    assert list(net["is_spiking"]) == [0, 1, 0] and net["last_firing_time"][1] == 10
    L = ob.lib()
    w = net["weights"]
    assert w[0, 1] == f32(1.0) + L.snn_o_stdp_delta(5, 10, 2, 2, 4.5, 4.5, 0.1)     # incoming, pre before post: +
    assert w[2, 1] == f32(1.0) + L.snn_o_stdp_delta(8, 10, 2, 2, 4.5, 4.5, 0.1)
    assert w[1, 0] == f32(1.0) + L.snn_o_stdp_delta(10, 5, 2, 2, 4.5, 4.5, 0.1)     # outgoing, post before pre: -
    assert w[1, 2] == f32(1.0) + L.snn_o_stdp_delta(10, 8, 2, 2, 4.5, 4.5, 0.1)
    assert w[0, 2] == 1.0 and w[2, 0] == 1.0                                         # not incident to the spike
    assert w[0, 1] > 1.0 > w[1, 0]


# ---- tests/interleaving_graph_conversion.rs (index placement) -----------------------------------------
def test_interleaved_index_space():
    """Lattices in ascending id, row-major; spike-train lattices after all neurons (graph/mod.rs:668-727)."""
    lay = parity.Layout([(3, 2, 2), (1, 1, 3)], [(2, 1, 2), (0, 2, 1)])
    r = lay.ranges()
    assert r[1] == (0, 3, False) and r[3] == (3, 4, False)
    assert r[0] == (0, 2, True) and r[2] == (2, 2, True)        # offsets inside the cell block
    assert lay.n_neurons == 7 and lay.n_cells == 4


# ---- further kinetics and the preset spike train (hand-derived single steps) -----------------------
def _one_cell(nt_kind, st_kind=ob.ST_RATE):
    net = ob.Net(0, n_cells=1, st_kind=st_kind, nt_kind=nt_kind)
    net["st_nt_flags"][0, 0] = 1
    return net


def test_exponential_decay_neurotransmitter_single_steps():
    """ExponentialDecayNeurotransmitter::apply_t_change (iterate_and_spike/mod.rs:345-354):
    t += (-t * exp(dt / -decay)) + spike * t_max, clamped to [0, t_max]; default decay_constant 2."""
    net = _one_cell(ob.NT_EXPONENTIAL_DECAY)
    assert net["st_nt_clearance"][0, 0] == f32(2.0)
    net["st_nt_t"][0, 0] = 0.5
    net.spike_trains()                                      # rate 0: no spike
    e = f32(ob.expf(f32(0.1) / f32(-2.0)))
    expect = f32(f32(0.5) + f32(f32(f32(-0.5) * e) + f32(0.0)))
    assert net["st_nt_t"][0, 0] == expect
    assert abs(float(expect) - (0.5 - 0.5 * np.exp(-0.05))) < 1e-7
    # a spiking cell adds t_max and the clamp holds it at t_max
    net["st_rate"] = 0.1
    net["st_step"] = 0.0
    net.spike_trains()
    assert net["st_is_spiking"][0] == 1 and net["st_nt_t"][0, 0] == f32(1.0)


def test_discrete_spike_neurotransmitter():
    """DiscreteSpikeNeurotransmitter (iterate_and_spike/mod.rs:300-302): t = t_max * is_spiking."""
    net = _one_cell(ob.NT_DISCRETE_SPIKE)
    net["st_nt_t_max"][0, 0] = 0.7
    net["st_nt_t"][0, 0] = 0.3
    net["st_rate"] = 0.2
    seen = []
    for _ in range(6):
        net.spike_trains()
        seen.append((int(net["st_is_spiking"][0]), float(net["st_nt_t"][0, 0])))
    assert seen == [(0, 0.0), (1, float(f32(0.7))), (0, 0.0), (1, float(f32(0.7))), (0, 0.0), (1, float(f32(0.7)))]


def test_exponential_decay_receptor_single_step():
    """ExponentialDecayReceptor::apply_r_change (iterate_and_spike/mod.rs:510-513): r += (-r * exp(dt / -decay)) + t,
    clamped to [0, r_max]; defaults r_max 1, decay_constant 2."""
    net = ob.Net(2, rc_kind=ob.RC_EXPONENTIAL_DECAY, chemical=True, electrical=False)
    assert net["rc_beta"][0, 0] == f32(2.0) and net["rc_alpha"][0, 0] == f32(1.0)
    net["nt_flags"][0, 0] = 1
    net["nt_t"][0, 0] = 0.25
    net["rc_flags"][1, 0] = 1
    net["rc_r"][1, 0] = 0.5
    net["connections"][0, 1] = 1
    net["weights"][0, 1] = 1.0
    net.inputs()
    net.update_neurons()
    e = f32(ob.expf(f32(0.1) / f32(-2.0)))
    expect = f32(f32(0.5) + f32(f32(f32(-0.5) * e) + f32(0.25)))
    assert net["rc_r"][1, 0] == expect
    # r_max clamps
    net["rc_alpha"][1, 0] = 0.2
    net.inputs()
    net.update_neurons()
    assert net["rc_r"][1, 0] == f32(0.2)


def test_preset_spike_train_cycles_through_its_firing_times():
    """PresetSpikeTrain::iterate (spike_train/mod.rs:803-827): the clock advances by dt, a spike happens when it
    EXCEEDS firing_times[counter], which resets the clock and advances the counter cyclically (the reference's
    backend/examples/stdp/main.rs:59-64 drives STDP with such cells)."""
    net = ob.Net(0, n_cells=2, st_kind=ob.ST_PRESET)
    net["st_dt"] = 1.0
    net.set_firing_times([[3.0, 1.0], []])
    spikes = []
    for i in range(14):
        net.spike_trains()
        spikes.append(int(net["st_is_spiking"][0]))
        assert net["st_is_spiking"][1] == 0                 # no firing times: never fires
    # clock 1,2,3,4(>3: spike, counter 1) | 1, 2(>1: spike, counter 0) | 1,2,3,4(spike) | 1,2(spike) | 1, 2
    assert spikes == [0, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0]
    assert net["st_counter"][0] == 0 and net["st_last_firing_time"][0] == 11
    assert net["st_current_voltage"][0] == f32(0.0) and net["st_step"][0] == f32(2.0)


# ---- reward modulation (plasticity/mod.rs:126-242) ---------------------------------------------------
def test_reward_modulated_stdp_hand_derived():
    """RewardModulatedSTDP::update (dopamine = dopamine * exp(-dt / tau_d) + tau_d * reward) and the two
    update_weight visits every edge receives per step (do_update is always true): dw accumulates the STDP delta
    twice, the trace becomes c * exp(-dt / tau_c) + tau_c * dw on the second visit and the weight moves by
    c * dopamine on both."""
    net = ob.Net(2)
    net["rm_do_modulation"] = 1
    net["rm_tau_c"] = 0.5
    net["rm_dopamine"] = 0.25
    net["connections"][0, 1] = 1
    net["weights"][0, 1] = 1.0
    net["traces"][0, 1] = 0.125
    net["last_firing_time"][...] = [10, 15]
    net.apply_reward(0.5)
    dop = f32(f32(f32(0.25) * f32(ob.expf(f32(-0.1) / f32(20.0)))) + f32(f32(20.0) * f32(0.5)))
    assert net["rm_dopamine"][0] == dop
    net.reward_modulation()
    delta = f32(f32(2.0) * f32(ob.expf(f32(f32(-1.0) * abs(f32(f32(10.0) - f32(15.0)) * f32(0.1))) / f32(4.5))))
    assert delta == f32(lib_stdp(10, 15))
    w1 = f32(f32(1.0) + f32(f32(0.125) * dop))
    dw = f32(delta + delta)
    c = f32(f32(f32(0.125) * f32(ob.expf(f32(-0.1) / f32(0.5)))) + f32(f32(0.5) * dw))
    w2 = f32(w1 + f32(c * dop))
    assert net["traces"][0, 1] == c and net["weights"][0, 1] == w2
    # an absent edge and an edge of a lattice without modulation stay untouched
    assert net["weights"][1, 0] == 0 and net["traces"][1, 0] == 0


def lib_stdp(tp, tq):
    return ob.lib().snn_o_stdp_delta(tp, tq, 2.0, 2.0, 4.5, 4.5, 0.1)


def test_exponential_decay_refractoriness_hand_derived():
    """ExponentialDecayRefractoriness::get_effect (spike_train/mod.rs:164-178): a * exp((-1 / (k / dt)) * Δ) + v_resting
    -- the delta-dirac form without the square; chosen per cell, used as the presynaptic value of a fired cell."""
    L = ob.lib()
    e = L.snn_o_exponential_decay_effect(14, 10, 30.0, -3.0, 50.0, 0.1)
    expect = f32(f32(f32(33.0) * f32(ob.expf(f32(f32(-1.0) / f32(f32(50.0) / f32(0.1))) * f32(4.0)))) + f32(-3.0))
    assert f32(e) == expect
    assert abs(float(e) - (33.0 * np.exp(-4.0 / 500.0) - 3.0)) < 1e-5
    net = ob.Net(1, n_cells=2, st_kind=ob.ST_RATE)
    net["st_k"] = 50.0
    net["st_v_resting"] = -3.0
    net["st_last_firing_time"][...] = 10
    net["st_refractoriness"][1] = 1
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = 1.0
    net["gap_conductance"] = 2.0
    net.clock = 14
    net.inputs()
    dd = f32(L.snn_o_delta_dirac_effect(14, 10, 30.0, -3.0, 50.0, 0.1))
    total = f32(f32(f32(f32(2.0) * dd) * f32(1.0)) + f32(f32(f32(2.0) * f32(e)) * f32(1.0)))
    assert net["input_current"][0] == f32(total / f32(2.0))


# ---- BCM (plasticity/mod.rs:72-116, integrate_and_fire/mod.rs:1358-1518, spike_train/mod.rs:835-970) -------------
def test_bcm_rule_and_activity_bookkeeping_hand_derived():
    """BCM::update_weight: w += (post.act * (post.act - post.avg / average_scalar) * pre.act - decay * w) * dt on the
    edges of a spiking neuron; BCMIzhikevichNeuron counts the PREVIOUS step's spike, closes a window when the clock
    reaches firing_rate_window (rate = num_spikes / (window * dt) without neurotransmission, / window with it) and
    moves the average by 1/period; num_spikes is never reset."""
    net = ob.Net(2, model=ob.BCM_IZHIKEVICH)
    net["plasticity_kind"] = 1
    net["do_plasticity"] = 1
    net["bcm_current_activity"][...] = [0.5, 2.0]
    net["bcm_average_activity"][...] = [0.1, 0.3]
    net["connections"][0, 1] = 1
    net["weights"][0, 1] = 1.5
    net["is_spiking"][1] = 1                       # the postsynaptic neuron spiked: its incoming edge is visited once
    net.plasticity()
    sliding = f32(f32(0.3) / f32(0.1))
    term = f32(f32(2.0) * f32(f32(2.0) - sliding))
    expect = f32(f32(1.5) + f32(f32(f32(term * f32(0.5)) - f32(f32(0.1) * f32(1.5))) * f32(0.1)))
    assert net["weights"][0, 1] == expect
    # both ends spiking: the edge is visited twice (incoming of 1, outgoing of 0), each visit from the current weight
    net["is_spiking"][0] = 1
    net.plasticity()
    w1 = f32(expect + f32(f32(f32(term * f32(0.5)) - f32(f32(0.1) * expect)) * f32(0.1)))
    w2 = f32(w1 + f32(f32(f32(term * f32(0.5)) - f32(f32(0.1) * w1)) * f32(0.1)))
    assert net["weights"][0, 1] == w2

    net = ob.Net(1, model=ob.BCM_IZHIKEVICH)
    net["bcm_window"] = 0.3                        # three steps of dt = 0.1 (clock: 0.1, 0.2, 0.3 in float32)
    net["is_spiking"][0] = 1
    net["current_voltage"] = -70.0
    clock = f32(0)
    for step in range(3):
        net.inputs()
        net.update_neurons()
        clock = f32(clock + f32(0.1))
    closes = clock >= f32(0.3)
    assert net["bcm_num_spikes"][0] == 1           # only the first step saw a previous spike
    if closes:
        rate = f32(f32(1.0) / f32(f32(0.3) * f32(0.1)))
        assert net["bcm_current_activity"][0] == rate and net["bcm_clock"][0] == 0
        assert net["bcm_average_activity"][0] == f32(f32(0) - f32(0) / f32(3)) + f32(rate / f32(3))
    else:
        assert net["bcm_current_activity"][0] == 0 and net["bcm_clock"][0] == clock


def test_bcm_poisson_cell_activity():
    """BCMPoissonNeuron::iterate: activity = voltage change of the step (v_th - V or v_resting - V), then the window."""
    net = ob.Net(0, n_cells=1, st_kind=ob.ST_BCM_POISSON)
    net["st_chance_of_firing"] = 1.0               # xorshift32 draw / 2^32 < 1: fires every step
    net.spike_trains()
    assert net["st_is_spiking"][0] == 1 and net["st_bcm_current_activity"][0] == f32(30.0) and net["st_bcm_num_spikes"][0] == 1
    net.spike_trains()
    assert net["st_bcm_current_activity"][0] == f32(0.0)       # 30 - 30
    net["st_chance_of_firing"] = 0.0
    net.spike_trains()
    assert net["st_bcm_current_activity"][0] == f32(-30.0) and net["st_bcm_num_spikes"][0] == 2

"""GPU parity for the other models and synapse kinds of the hot path: LIF, Hodgkin-Huxley (Na/K/K-leak
gating), Approximate and Destexhe neurotransmitter/receptor kinetics with AMPA/NMDA/GABA, electrical +
chemical in one pass, chemical only.  Bar: every state array and history bit-identical to the oracle."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def compare(snn, net, steps, chunks=1):
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    for _ in range(chunks):
        dn.run(steps // chunks)
    net.run(steps, voltage_history=True, spike_history=True)
    assert np.array_equal(dn.spike_history(net.layout.lattices[0][0]),
                          net.spike_history[:, :net.layout.ranges()[net.layout.lattices[0][0]][1]])
    vh = np.concatenate([dn.voltage_history(i) for i, _, _ in net.layout.lattices], axis=1)
    assert np.array_equal(parity.bits(vh), parity.bits(net.voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()
    return net


def test_lif_lattice_electrical(snn):
    lay = parity.Layout([(0, 9, 11)])
    net = parity.make_oracle(lay, model=ob.LIF)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(5, n, -80.0, -50.0)
    net["gap_conductance"] = 10.0
    net["tref"] = ob.uniform_array(6, n, 0.5, 2.0)
    net.fill_graph(7, 0.5, 1.5)
    compare(snn, net, 600, chunks=2)
    assert net.spike_history.sum() > 0


def test_hodgkin_huxley_lattice_electrical(snn):
    """HH with Na (m^3 h), K (n^4) and K-leak channels, gap junctions only, dt = 0.01."""
    lay = parity.Layout([(0, 6, 6)])
    net = parity.make_oracle(lay, model=ob.HH)
    net["current_voltage"] = ob.uniform_array(3, net.n_neurons, -70.0, -60.0)
    net.fill_graph(4, 0.5, 1.5)
    compare(snn, net, 3000)
    assert net.spike_history.sum() > 0


def test_c3_small_hodgkin_huxley_destexhe_ampa(snn):
    """BASELINE configs[2] at test size: HH + Na/K channels + Destexhe neurotransmitter (AMPA) + Destexhe
    receptor (HodgkinHuxleyNeuron::default_impl, hodgkin_huxley/mod.rs:100-105), electrical + chemical."""
    lay = parity.Layout([(0, 8, 8)])
    net = parity.make_oracle(lay, model=ob.HH, nt_kind=ob.NT_DESTEXHE, rc_kind=ob.RC_DESTEXHE, chemical=True)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(3, n, -70.0, -60.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(4, 0.5, 1.5)
    compare(snn, net, 2500)
    assert net.spike_history.sum() > 0 and net["rc_r"][:, 0].max() > 0


@pytest.mark.parametrize("electrical", [True, False])
def test_izhikevich_approximate_kinetics_all_receptor_types(snn, electrical):
    """Approximate neurotransmitter + Approximate receptor, AMPA/NMDA/GABA present on random subsets
    (per-type averaging counts only presynaptic cells that carry the type, iterate_and_spike/mod.rs:2847-2853)."""
    lay = parity.Layout([(0, 10, 10)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH, chemical=True, electrical=electrical)
    n = net.n_neurons
    rng = np.random.default_rng(9)
    net["current_voltage"] = ob.uniform_array(8, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][...] = rng.random((n, 3)) < 0.6
    net["rc_flags"][...] = rng.random((n, 3)) < 0.7
    net["nt_t"][...] = rng.random((n, 3)).astype(np.float32) * net["nt_flags"]
    net["nt_clearance"][...] = ob.uniform_array(10, 3 * n, 0.005, 0.05).reshape(n, 3)
    net["rc_g"][...] *= ob.uniform_array(11, 3 * n, 0.5, 1.5).reshape(n, 3)
    net.fill_graph(12, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    compare(snn, net, 800, chunks=4)
    assert net.spike_history.sum() > 0


def test_izhikevich_destexhe_kinetics(snn):
    lay = parity.Layout([(0, 7, 9)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH, nt_kind=ob.NT_DESTEXHE, rc_kind=ob.RC_DESTEXHE, chemical=True)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(13, n, -65.0, 30.0)
    net["nt_flags"][...] = 1
    net["rc_flags"][...] = 1
    net["rc_alpha"][...] = ob.uniform_array(14, 3 * n, 0.5, 2.0).reshape(n, 3)
    net["rc_beta"][...] = ob.uniform_array(15, 3 * n, 0.5, 2.0).reshape(n, 3)
    net.fill_graph(16, 0.2, 1.0)
    compare(snn, net, 600)
    assert net.spike_history.sum() > 0


def test_lif_chemical_only(snn):
    lay = parity.Layout([(0, 5, 5)])
    net = parity.make_oracle(lay, model=ob.LIF, chemical=True, electrical=False)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(17, n, -70.0, -50.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_e"][:, 0] = 0.0
    net["nt_t"][:, 0] = 0.8
    net["tref"] = 1.0
    net.fill_graph(18, 0.5, 1.5)
    compare(snn, net, 500)


@pytest.mark.parametrize("chemical", [False, True])
def test_quadratic_integrate_and_fire(snn, chemical):
    """QuadraticIntegrateAndFireNeuron -- one of the two models the reference's own GPU path implements
    (integrate_and_fire/mod.rs:368-917); buffer names alpha, v_c, v_reset, integration_constant, tau_m, ..."""
    lay = parity.Layout([(0, 6, 7)])
    net = parity.make_oracle(lay, model=ob.QIF, chemical=chemical)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(7, n, -75.0, -56.0)
    net["gap_conductance"] = 3.0
    net["tref"] = ob.uniform_array(8, n, 0.3, 1.5)
    net["tau_m"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(9, 0.5, 1.5)
    compare(snn, net, 600, chunks=2)
    assert net.spike_history.sum() > 0


@pytest.mark.parametrize("chemical", [False, True])
def test_simple_leaky_integrate_and_fire(snn, chemical):
    """SimpleLeakyIntegrateAndFire (integrate_and_fire/mod.rs:1523-1801), buffer names g, e, v_reset."""
    lay = parity.Layout([(0, 5, 8)])
    net = parity.make_oracle(lay, model=ob.SIMPLE_LIF, chemical=chemical)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(9, n, -75.0, -56.0)
    net["slif_g"] = 0.5
    net["slif_e"] = -76.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(10, 0.5, 1.5)
    compare(snn, net, 600, chunks=3)
    assert net.spike_history.sum() > 0


@pytest.mark.parametrize("chemical", [False, True])
@pytest.mark.parametrize("exponential", [False, True])
def test_adaptive_leaky_integrate_and_fire(snn, exponential, chemical):
    """AdaptiveLeakyIntegrateAndFireNeuron (integrate_and_fire/mod.rs:918-1049) and the exponential variant
    (:1051-1155): attributes alpha, beta, slope_factor, w_value next to the LIF set; heterogeneous parameters."""
    lay = parity.Layout([(0, 6, 7)])
    net = parity.make_oracle(lay, model=ob.ADAPTIVE_EXP_LIF if exponential else ob.ADAPTIVE_LIF, chemical=chemical)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(21, n, -75.0, -56.0)
    net["gap_conductance"] = 3.0
    net["tref"] = ob.uniform_array(22, n, 0.3, 1.5)
    net["adp_beta"] = ob.uniform_array(23, n, 1.0, 4.0)
    net["adp_alpha"] = ob.uniform_array(24, n, 3.0, 8.0)
    net["leak_constant"] = 1.0
    net["c_m"] = 1.0
    net["v_reset"] = -73.0
    if exponential:
        net["slope_factor"] = ob.uniform_array(25, n, 0.5, 3.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(26, 0.5, 1.5)
    compare(snn, net, 600, chunks=2)
    assert net.spike_history.sum() > 0


@pytest.mark.parametrize("chemical", [False, True])
def test_leaky_izhikevich(snn, chemical):
    """LeakyIzhikevichNeuron (integrate_and_fire/mod.rs:1270-1356): the Izhikevich attributes + e_l."""
    lay = parity.Layout([(0, 7, 6)])
    net = parity.make_oracle(lay, model=ob.LEAKY_IZHIKEVICH, chemical=chemical)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(27, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["w_value"] = ob.uniform_array(28, n, 0.0, 1.0)
    net["e_l"] = ob.uniform_array(29, n, -70.0, -60.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(30, 0.5, 1.5)
    compare(snn, net, 600, chunks=3)
    assert net.spike_history.sum() > 0


@pytest.mark.parametrize("nt_kind,rc_kind", [(ob.NT_EXPONENTIAL_DECAY, ob.RC_EXPONENTIAL_DECAY),
                                             (ob.NT_DISCRETE_SPIKE, ob.RC_APPROX),
                                             (ob.NT_EXPONENTIAL_DECAY, ob.RC_DESTEXHE),
                                             (ob.NT_APPROX, ob.RC_EXPONENTIAL_DECAY)])
def test_exponential_decay_and_discrete_spike_kinetics(snn, nt_kind, rc_kind):
    """ExponentialDecayNeurotransmitter / DiscreteSpikeNeurotransmitter (iterate_and_spike/mod.rs:287-366) and
    ExponentialDecayReceptor (:497-533) with heterogeneous decay constants and r_max, all three receptor types."""
    lay = parity.Layout([(0, 8, 9)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH, nt_kind=nt_kind, rc_kind=rc_kind, chemical=True)
    n = net.n_neurons
    rng = np.random.default_rng(31)
    net["current_voltage"] = ob.uniform_array(31, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][...] = rng.random((n, 3)) < 0.7
    net["rc_flags"][...] = rng.random((n, 3)) < 0.7
    if nt_kind == ob.NT_EXPONENTIAL_DECAY:
        net["nt_clearance"][...] = ob.uniform_array(32, 3 * n, 0.5, 4.0).reshape(n, 3)      # decay_constant
    if rc_kind == ob.RC_EXPONENTIAL_DECAY:
        net["rc_alpha"][...] = ob.uniform_array(33, 3 * n, 0.3, 1.0).reshape(n, 3)         # r_max
        net["rc_beta"][...] = ob.uniform_array(34, 3 * n, 0.5, 4.0).reshape(n, 3)          # decay_constant
    net.fill_graph(35, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    compare(snn, net, 800, chunks=2)
    assert net.spike_history.sum() > 0
    if nt_kind != ob.NT_DISCRETE_SPIKE:          # a discrete release is over one step after the last spike
        assert net["rc_r"].max() > 0

"""Host side of `snn_amd.lixirnet` (no GPU): the seventeen names of the reference's module, their constructors and
attributes, and the building / reading methods of `impl_network!` / `impl_network_gpu!`
(interface_gpu/lixirnet/src/lattices/mod.rs:697-2117) answered by the host container -- nothing is stepped here."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ln():
    import snn_amd                                   # noqa: F401  (publishes the package)
    from snn_amd import lixirnet
    return lixirnet


def test_the_seventeen_classes_of_the_reference_module(ln):
    names = ["IzhikevichNeuron", "BoundedNeurotransmitterKinetics", "BoundedReceptorKinetics", "DopaGluGABANeurotransmitterType",
             "GlutamateReceptor", "GABAReceptor", "DopamineReceptor", "DopaGluGABA", "STDP", "IzhikevichNeuronLattice",
             "IzhikevichNeuronLatticeGPU", "DeltaDiracRefractoriness", "RateSpikeTrain", "RateSpikeTrainLattice", "GraphPosition",
             "IzhikevichNeuronNetwork", "IzhikevichNeuronNetworkGPU"]            # lib.rs:463-482
    assert sorted(ln.__all__) == sorted(names)
    for n in names:
        assert getattr(ln, n).__name__ == n
    n = ln.IzhikevichNeuron()                                                       # lib.rs:68-78: the description's defaults
    assert (n.u, n.a, n.b, n.c, n.d, n.v_th, n.tau_m, n.c_m) == (30.0, 0.02, 0.2, -55.0, 8.0, 30.0, 1.0, 100.0)
    assert (n.current_voltage, n.dt, n.gap_conductance, n.is_spiking, n.last_firing_time) == (0.0, 0.1, 10.0, False, None)
    k = ln.BoundedNeurotransmitterKinetics()
    assert (k.t, k.t_max, k.clearance_constant, k.conc) == (0.0, 1.0, 0.001, 0.0)
    assert (ln.BoundedReceptorKinetics().r, ln.BoundedReceptorKinetics().r_max) == (0.0, 1.0)
    T = ln.DopaGluGABANeurotransmitterType
    assert [t.name for t in T] == ["Glutamate", "GABA", "Dopamine"]
    glu, gaba, dopa = ln.GlutamateReceptor(), ln.GABAReceptor(), ln.DopamineReceptor()
    assert (glu.g_ampa, glu.g_nmda, glu.mg, glu.ampa_r.r_max, glu.nmda_r.r) == (1.0, 0.6, 0.3, 1.0, 0.0)
    assert (gaba.g, gaba.e, gaba.r.r) == (1.2, -80.0, 0.0) and (dopa.s_d1, dopa.s_d2, dopa.r_d1.r, dopa.r_d2.r_max) == (0.0, 0.0, 0.0, 1.0)
    r = ln.DopaGluGABA()
    assert (r.inh_modifier, r.nmda_modifier) == (1.0, 1.0) and len(r) == 0
    r.insert(T.Glutamate, glu)
    with pytest.raises(ValueError, match="MismatchedTypes"):
        r.insert(T.GABA, ln.GlutamateReceptor())
    n.set_receptors(r)
    n.set_synaptic_neurotransmitters({T.Glutamate: k})
    with pytest.raises(TypeError):
        n.set_synaptic_neurotransmitters({T.Glutamate: ln.BoundedReceptorKinetics()})
    s = ln.STDP()
    assert (s.a_plus, s.a_minus, s.tau_plus, s.tau_minus, s.dt) == (2.0, 2.0, 4.5, 4.5, 0.1)


def test_rate_spike_train_and_refractoriness(ln):
    st = ln.RateSpikeTrain()
    st.rate = 0.35
    fired = [st.iterate() for _ in range(12)]                   # dt 0.1: step reaches the rate every fourth iteration
    assert fired == [False, False, False, True] * 3
    assert st.current_voltage == st.v_th and st.is_spiking
    ref = st.get_refractoriness()
    assert isinstance(ref, ln.DeltaDiracRefractoriness) and ref.k == 10000.0
    st.set_refractoriness(ln.DeltaDiracRefractoriness(500.0))
    assert st.get_refractoriness().k == 500.0
    # get_effect(timestep, last_firing_time, v_max, v_resting, dt): a * exp(-(dt / k) * time_difference^2) + v_resting
    got = ln.DeltaDiracRefractoriness(10000.0).get_effect(15, 10, 30.0, 0.0, 0.1)
    assert got == pytest.approx(30.0 * np.exp(-(0.1 / 10000.0) * 25.0), rel=1e-6)


def network(ln):
    neuron = ln.IzhikevichNeuron()
    a, b = ln.IzhikevichNeuronLattice(0), ln.IzhikevichNeuronLattice(1)
    a.populate(neuron, 3, 3)
    b.populate(neuron, 2, 2)
    a.connect(lambda x, y: x != y, lambda x, y: 5)
    b.connect(lambda x, y: x != y)
    st = ln.RateSpikeTrainLattice(4)
    st.populate(ln.RateSpikeTrain(), 3, 3)
    net = ln.IzhikevichNeuronNetwork.generate_network([a, b], [st])
    net.connect(0, 1, lambda x, y: x == y, lambda x, y: 2.5)
    net.connect(4, 0, lambda x, y: x == y)
    return net


def test_network_methods_of_the_reference_interface(ln):
    net = network(ln)
    G = ln.GraphPosition
    assert net.get_all_ids() == {0, 1, 4}
    assert net.get_weight(G(0, (0, 0)), G(0, (1, 1))) == 5.0 and net.get_weight(G(0, (0, 0)), G(0, (0, 0))) == 0.0
    assert net.get_weight(G(0, (1, 1)), G(1, (1, 1))) == 2.5 and net.get_weight(G(0, (2, 2)), G(1, (1, 1))) == 0.0
    assert net.get_weight(G(4, (2, 1)), G(0, (2, 1))) == 1.0
    assert net.get_weight(G(1, (0, 0)), G(4, (0, 0))) == 0.0           # both in the connecting graph, no edge: Ok(None) -> 0
    with pytest.raises(KeyError):
        net.get_weight(G(1, (5, 5)), G(4, (0, 0)))                     # GraphError::PositionNotFound
    assert net.get_incoming_connections_within_lattice(1, (0, 0)) == {(0, 1), (1, 0), (1, 1)}
    assert net.get_outgoing_connections_within_lattice(0, (1, 1)) == {(r, c) for r in range(3) for c in range(3)} - {(1, 1)}
    assert net.get_incoming_connectings_across_lattices(0, (2, 0)) == {G(4, (2, 0))}
    assert net.get_outgoing_connectings_across_lattices(0, (1, 0)) == {G(1, (1, 0))}
    assert net.get_outgoing_connectings_across_lattices(0, (2, 2)) == set()
    with pytest.raises(KeyError):
        net.get_incoming_connectings_across_lattices(7, (0, 0))
    index = net.connecting_position_to_index
    assert len(index) == 9 + 4 + 9 and net.get_connecting_position_to_index() == index
    m = net.connecting_weights
    assert m.shape == (22, 22) and m[index[G(0, (1, 0))], index[G(1, (1, 0))]] == 2.5 and m[G(4, (0, 2)), G(0, (0, 2))] == 1.0
    assert float(m.sum()) == 4 * 2.5 + 9 * 1.0
    # cells
    n = net.get_neuron(0, 2, 1)
    n.current_voltage = -33.0
    net.set_neuron(0, 2, 1, n)
    assert net.get_neuron(0, 2, 1).current_voltage == -33.0 and net.get_neuron(0, 0, 0).current_voltage == 0.0
    with pytest.raises(KeyError):
        net.get_neuron(0, 3, 0)
    s = net.get_spike_train(4, 0, 0)
    s.rate = 7.0
    net.set_spike_train(4, 0, 0, s)
    assert net.get_spike_train(4, 0, 0).rate == 7.0
    net.apply_lattice(1, lambda neuron: setattr(neuron, "u", 12.0))
    net.apply_lattice_given_position(1, lambda pos, neuron: setattr(neuron, "a", 0.01 * (1 + pos[0] + 2 * pos[1])))
    assert [c.u for row in net.get_lattice(1).cell_grid for c in row] == [12.0] * 4
    assert net.get_neuron(1, 1, 1).a == pytest.approx(0.04)
    net.apply_spike_train_lattice(4, lambda t: setattr(t, "dt", 0.05))
    net.apply_spike_train_lattice_given_position(4, lambda pos, t: setattr(t, "step", float(pos[0])))
    assert net.get_spike_train(4, 2, 0).dt == 0.05 and net.get_spike_train(4, 2, 0).step == 2.0
    # per-lattice switches
    net.set_do_plasticity(1, True)
    p = net.get_plasticity(1)
    p.a_plus = 3.0
    net.set_plasticity(1, p)
    assert net.get_do_plasticity(1) and net.get_plasticity(1).a_plus == 3.0 and net.get_plasticity(0).a_plus == 2.0
    net.set_update_grid_history(4, True)
    net.set_update_graph_history(0, True)
    assert net.get_update_grid_history(4) and not net.get_update_grid_history(0) and net.get_update_graph_history(0)
    net.set_dt(0.2)
    assert net.get_neuron(0, 0, 0).dt == 0.2 and net.get_plasticity(0).dt == 0.2 and net.get_spike_train(4, 1, 1).dt == 0.2
    # replacing lattices
    other = ln.IzhikevichNeuronLattice(9)
    other.populate(ln.IzhikevichNeuron(), 2, 2)
    net.set_lattice(1, other)
    assert net.get_lattice(1).id == 1 and net.get_neuron(1, 0, 0).u == 30.0
    with pytest.raises(KeyError):
        net.set_lattice(5, other)
    with pytest.raises(KeyError, match="GraphIDAlreadyPresent"):
        net.add_lattice(ln.IzhikevichNeuronLattice(4))
    with pytest.raises(KeyError, match="PostsynapticLatticeCannotBeSpikeTrain"):
        net.connect(0, 4, lambda x, y: True)
    net.clear()
    assert net.get_all_ids() == set() and net.connecting_weights.shape == (0, 0)


def test_lattice_interface(ln):
    lat = ln.IzhikevichNeuronLattice(3)
    lat.populate(ln.IzhikevichNeuron(), 2, 3)
    lat.connect(lambda x, y: x[0] == y[0] and x != y, lambda x, y: 0.5)
    assert lat.id == 3 and lat.get_every_node() == {(r, c) for r in range(2) for c in range(3)}
    assert lat.get_weight((0, 0), (0, 2)) == 0.5
    assert lat.get_weight((0, 0), (1, 0)) == 0.0                # lookup_weight -> None -> unwrap_or(0.)
    with pytest.raises(KeyError):
        lat.get_weight((0, 0), (2, 0))                          # outside the lattice
    assert lat.get_incoming_connections((1, 1)) == {(1, 0), (1, 2)} and lat.get_outgoing_connections((0, 2)) == {(0, 0), (0, 1)}
    w = lat.get_weights()
    assert w.shape == (6, 6) and w.sum() == 0.5 * 12 and lat.get_position_to_index_for_weights()[(1, 2)] == 5
    assert lat.history.shape[0] == 0 and not lat.update_grid_history and not lat.parallel
    lat.reset_timing()
    lat.reset_history()
    assert lat.history.shape == (0, 2, 3)
    with pytest.raises(NotImplementedError):
        lat.run_lattice(1)

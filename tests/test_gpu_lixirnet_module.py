"""`snn_amd.lixirnet` -- the drop-in for the reference's GPU Python module -- on the device.

(1) The description itself (interface_gpu/lixirnet/src/lib.rs:22-79) as generated HIP against the oracle's stack
    programs (which tests/test_lixirnet_description.py holds to a hand-written restatement): dense, sparse and sharded.
(2) The PROCEDURE of the reference's own Python tests (interface_gpu/lixirnet/tests/networks.py: electrical, chemical
    with glutamate, with Rate spike trains; tests/lattices.py: a lone lattice) written against the same seventeen names
    -- only the import differs -- with the CPU side of the comparison played by the oracle, bit for bit."""
import numpy as np
import pytest

import lixirnet_case as lc
import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ln(snn):
    from snn_amd import lixirnet
    lixirnet.IzhikevichNeuron            # compiles the description's library (cached)
    return lixirnet


def case(chemical, electrical, cells, seed=5):
    import test_lixirnet_description as t
    return t.build(chemical, electrical, cells, seed)


@pytest.mark.parametrize("chemical,electrical,cells,form", [(False, True, False, "dense"), (True, False, False, "dense"),
                                                            (True, True, True, "dense"), (True, True, True, "csr"),
                                                            (True, True, True, "shards")])
def test_generated_hip_equals_the_oracle_programs(snn, ln, chemical, electrical, cells, form):
    net = case(chemical, electrical, cells)
    net.custom_lib = ln.library
    steps = 700
    if form == "shards":
        from snn_amd import parallel
        import torch
        handles = [parity.device_from_oracle(snn, net, shard=(r, 2)) for r in range(2)]
        ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
        for _ in range(steps):
            ex.step()
        net.run(steps)
        for h in handles:
            st = parity.pull_state(h, net)
            own = h.owned
            for k in ("current_voltage", "last_firing_time"):
                assert np.array_equal(parity.bits(st[k][own]), parity.bits(net[k][own])), k
            assert np.array_equal(parity.bits(st["rx_vars"][:, own]), parity.bits(net["rx_vars"][:, own]))
            assert np.array_equal(parity.bits(st["custom_vars"][:, own]), parity.bits(net["custom_vars"][:, own]))
            h.close()
        return
    dn = parity.device_from_oracle(snn, net, csr=(form == "csr"))
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps)
    net.run(steps, voltage_history=True, spike_history=True)
    assert net.spike_history.sum() >= 10
    for i, (first, count, _) in ((i, r) for i, r in net.layout.ranges().items() if not r[2]):
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


# ---- the reference's Python procedure --------------------------------------------------------------------------------
e1, e2, i1, c1 = 0, 1, 2, 4
exc_n1, exc_n2 = 3, 2
iterations = 1000


def get_neuron_setup(init_state):
    def setup_neuron(pos, neuron):
        x, y = pos
        neuron.current_voltage = init_state[x][y]
        return neuron
    return setup_neuron


def get_spike_train_setup(init_state):
    def setup_spike_train(pos, neuron):
        x, y = pos
        neuron.step = init_state[x][y]
        return neuron
    return setup_spike_train


def oracle_of(ln, network):
    """the CPU side of the reference's comparison: the same network on the oracle (stack programs of the description)"""
    lattices = sorted(network.lattices.items())
    trains = sorted(network.spike_train_lattices.items())
    lay = parity.Layout([(i, l.rows, l.cols) for i, l in lattices], [(i, l.rows, l.cols) for i, l in trains])
    net = lc.oracle_net(lay, electrical=network.electrical_synapse, chemical=network.chemical_synapse,
                        st_kind=ob.ST_RATE if trains else ob.ST_NONE)
    rng = lay.ranges()
    d = net.description
    for i, l in lattices:
        first, count, _ = rng[i]
        cells = [c for row in l.cell_grid for c in row]
        for k in ("current_voltage", "dt", "c_m", "gap_conductance"):
            net[k][first:first + count] = [getattr(c, k) for c in cells]
        for name, _ in d.neuron.variables:
            lc.var(net, "custom_vars", name)[first:first + count] = [getattr(c, name) for c in cells]
        for j, c in enumerate(cells):
            q = first + j
            for ty, kin in c.synaptic_neurotransmitters.items():
                net["nt_flags"][q, int(ty)] = 1
                net["nt_t"][q, int(ty)] = kin.t
                for name, _ in d.nt_kinetics.variables:
                    lc.var(net, "nt_custom_vars", name)[q, int(ty)] = getattr(kin, name)
            for name in ("inh_modifier", "nmda_modifier"):
                lc.var(net, "rx_vars", name)[q] = getattr(c.receptors, name)
            for ty, rec in c.receptors.items():
                net["rc_flags"][q, int(ty)] = 1
                tname = ln.DopaGluGABANeurotransmitterType(int(ty)).name
                for name, _ in d.receptors.variables:
                    parts = name.split("$")
                    if parts[0] != tname:
                        continue
                    value = getattr(getattr(rec, parts[1]), parts[3]) if len(parts) == 4 else getattr(rec, parts[1])
                    lc.var(net, "rx_vars", name)[q] = value
        net["weights"][first:first + count, first:first + count] = l.weights
        net["connections"][first:first + count, first:first + count] = l.connections
        slot = [x[0] for x in lattices].index(i)
        p = l.plasticity
        for k, v in (("stdp_a_plus", p.a_plus), ("stdp_a_minus", p.a_minus), ("stdp_tau_plus", p.tau_plus),
                     ("stdp_tau_minus", p.tau_minus), ("stdp_dt", p.dt), ("do_plasticity", int(l.do_plasticity))):
            net[k][slot] = v
    nn = net.n_neurons
    for i, l in trains:
        first, count, _ = rng[i]
        cells = [c for row in l.cell_grid for c in row]
        for k, a in (("current_voltage", "st_current_voltage"), ("v_th", "st_v_th"), ("v_resting", "st_v_resting"),
                     ("dt", "st_dt"), ("k", "st_k"), ("rate", "st_rate"), ("step", "st_step")):
            net[a][first:first + count] = [getattr(c, k) for c in cells]
        for j, c in enumerate(cells):
            for ty, kin in c.synaptic_neurotransmitters.items():
                net["st_nt_flags"][first + j, int(ty)] = 1
                net["st_nt_t"][first + j, int(ty)] = kin.t
                for name, _ in d.nt_kinetics.variables:
                    lc.var(net, "st_nt_custom_vars", name)[first + j, int(ty)] = getattr(kin, name)

    def index(gp):
        first, _, is_st = rng[gp.id]
        l = network.spike_train_lattices[gp.id] if is_st else network.lattices[gp.id]
        return (nn if is_st else 0) + first + gp.pos[0] * l.cols + gp.pos[1]

    for (pre, post), w in network.connecting.items():
        net["weights"][index(pre), index(post)] = w
        net["connections"][index(pre), index(post)] = 1
    return net


def check_against_oracle(ln, network, gpu_network, ids, st_ids=()):
    net = oracle_of(ln, network)
    # what the reference's test asserts before the run: same weights, same voltages on both sides
    for i in ids:
        l, g = network.get_lattice(i), gpu_network.get_lattice(i)
        for n in range(l.rows):
            for m in range(l.cols):
                assert abs(network.get_lattice(i).get_neuron(n, m).current_voltage - g.get_neuron(n, m).current_voltage) < 0.1
                assert l.get_weight((n, m), ((n + 1) % l.rows, m)) == g.get_weight((n, m), ((n + 1) % l.rows, m))
    assert np.array_equal(network.connecting_weights, gpu_network.connecting_weights)
    gpu_network.run_lattices(iterations)
    net.run(iterations, voltage_history=True, st_voltage_history=bool(st_ids))
    rng = net.layout.ranges()
    for i in ids:
        first, count, _ = rng[i]
        hist = np.array(gpu_network.get_lattice(i).history, np.float32)
        want = net.voltage_history[:, first:first + count].reshape(hist.shape)
        assert hist.shape[0] == iterations
        assert np.array_equal(parity.bits(hist), parity.bits(want)), i          # the reference asserts |sum of differences| < 0.1
        l = gpu_network.get_lattice(i)
        assert l.get_neuron(0, 0).current_voltage == float(net["current_voltage"][first])
    for i in st_ids:
        first, count, _ = rng[i]
        hist = np.array(gpu_network.get_spike_train_lattice(i).history, np.float32)
        assert np.array_equal(parity.bits(hist.reshape(iterations, -1)), parity.bits(net.st_voltage_history[:, first:first + count]))
    return net


def build_network(ln, rng, chemical, spike_trains):
    neuron = ln.IzhikevichNeuron()
    neuron.gap_conductance = 5 if (chemical and spike_trains) else 10
    neuron.c_m = 25
    if chemical:
        glu_neuro = ln.BoundedNeurotransmitterKinetics()
        exc_neurotransmitters = {ln.DopaGluGABANeurotransmitterType.Glutamate: glu_neuro}
        glu = ln.GlutamateReceptor()
        receptors = ln.DopaGluGABA()
        receptors.insert(ln.DopaGluGABANeurotransmitterType.Glutamate, glu)
        neuron.set_synaptic_neurotransmitters(exc_neurotransmitters)
        neuron.set_receptors(receptors)
    init_state1 = rng.uniform(neuron.c, neuron.v_th, (exc_n1, exc_n1))
    init_state2 = rng.uniform(neuron.c, neuron.v_th, (exc_n2, exc_n2))
    second = i1 if spike_trains else e2
    lattice1 = ln.IzhikevichNeuronLattice(e1)
    lattice1.populate(neuron, exc_n1, exc_n1)
    lattice1.apply_given_position(get_neuron_setup(init_state1))
    lattice1.connect(lambda x, y: x != y, (lambda x, y: 2) if chemical else (lambda x, y: 5))
    lattice1.update_grid_history = True
    lattice2 = ln.IzhikevichNeuronLattice(second)
    lattice2.populate(neuron, exc_n2, exc_n2)
    lattice2.apply_given_position(get_neuron_setup(init_state2))
    lattice2.connect(lambda x, y: x != y, (lambda x, y: 0.5) if chemical else (lambda x, y: 3))
    lattice2.update_grid_history = True
    trains = []
    if spike_trains:
        spike_train = ln.RateSpikeTrain()
        spike_train.rate = 100
        if chemical:
            spike_train.set_synaptic_neurotransmitters({ln.DopaGluGABANeurotransmitterType.Glutamate: ln.BoundedNeurotransmitterKinetics()})
        stl = ln.RateSpikeTrainLattice(c1)
        stl.populate(spike_train, exc_n1, exc_n1)
        stl.apply_given_position(get_spike_train_setup(rng.uniform(0, 100, (exc_n1, exc_n1))))
        stl.update_grid_history = True
        trains.append(stl)
    network = ln.IzhikevichNeuronNetwork.generate_network([lattice1, lattice2], trains)
    network.connect(e1, second, lambda x, y: x == y, (lambda x, y: 1) if chemical else (lambda x, y: 5))
    network.connect(second, e1, lambda x, y: x == y, (lambda x, y: 1) if chemical else (lambda x, y: -3 if spike_trains else 3))
    if spike_trains:
        network.connect(c1, e1, lambda x, y: x == y, lambda x, y: 5)
    network.electrical_synapse = not chemical
    network.chemical_synapse = chemical
    return network, second


@pytest.mark.parametrize("chemical,spike_trains", [(False, False), (True, False), (False, True), (True, True)])
def test_networks_py_procedure(ln, chemical, spike_trains):
    """networks.py::test_network_{electrical,chemical}_using_from and ..._with_spike_trains"""
    rng = np.random.default_rng(10 + 2 * chemical + spike_trains)
    network, second = build_network(ln, rng, chemical, spike_trains)
    gpu_network = ln.IzhikevichNeuronNetworkGPU.from_network(network)
    net = check_against_oracle(ln, network, gpu_network, (e1, second), (c1,) if spike_trains else ())
    if chemical:
        assert np.abs(lc.var(net, "rx_vars", "Glutamate$current")).max() > 0
    # the generated library carries the one-launch step for its own neuron model (round 5): inputs + update of a step in one launch
    dn = gpu_network._dn
    assert dn.stat("steps_dense_one_launch") == iterations and dn.stat("steps_two_kernel") == 0
    gpu_network.close()


def test_network_built_on_the_gpu_class_itself(ln):
    """impl_network_gpu! carries the builders too (generate_network, add_lattice, connect, set_neuron, apply_lattice ...):
    a network assembled on the GPU class, edited between two runs, equals the host-built one"""
    rng = np.random.default_rng(3)
    network, second = build_network(ln, rng, chemical=False, spike_trains=False)
    g = ln.IzhikevichNeuronNetworkGPU.generate_network([network.get_lattice(e1), network.get_lattice(second)], [])
    g.connect(e1, second, lambda x, y: x == y, lambda x, y: 5)
    g.connect(second, e1, lambda x, y: x == y, lambda x, y: 3)
    g.electrical_synapse, g.chemical_synapse = True, False
    assert g.get_all_ids() == {e1, second}
    assert g.get_weight(ln.GraphPosition(e1, (0, 0)), ln.GraphPosition(second, (0, 0))) == 5.0
    assert g.get_weight(ln.GraphPosition(e1, (0, 1)), ln.GraphPosition(second, (0, 0))) == 0.0
    assert g.get_weight(ln.GraphPosition(e1, (0, 0)), ln.GraphPosition(e1, (1, 1))) == 5.0
    assert g.get_incoming_connectings_across_lattices(second, (1, 1)) == {ln.GraphPosition(e1, (1, 1))}
    assert g.get_outgoing_connectings_across_lattices(second, (0, 1)) == {ln.GraphPosition(e1, (0, 1))}
    assert g.get_incoming_connections_within_lattice(second, (0, 0)) == {(0, 1), (1, 0), (1, 1)}
    # connect() puts every position of both lattices into the connecting graph (neuron/mod.rs:1876-1880): 9 + 4 nodes
    assert len(g.connecting_position_to_index) == 13 and g.connecting_weights.shape == (13, 13)
    g.run_lattices(200)
    # edit between runs: one neuron replaced, one lattice's voltages shifted
    n = g.get_neuron(e1, 1, 1)
    n.current_voltage = -40.0
    g.set_neuron(e1, 1, 1, n)
    g.apply_lattice(second, lambda neuron: setattr(neuron, "u", neuron.u + 1.0))
    g.run_lattices(300)
    # the same, host-built
    h = ln.IzhikevichNeuronNetworkGPU.from_network(network)
    h.run_lattices(200)
    host = h.network
    n = host.get_neuron(e1, 1, 1)
    n.current_voltage = -40.0
    host.set_neuron(e1, 1, 1, n)
    host.apply_lattice(second, lambda neuron: setattr(neuron, "u", neuron.u + 1.0))
    h2 = ln.IzhikevichNeuronNetworkGPU.from_network(host)
    h2.run_lattices(300)
    for i in (e1, second):
        a, b = g.get_lattice(i), h2.get_lattice(i)
        assert [c.current_voltage for row in a.cell_grid for c in row] == [c.current_voltage for row in b.cell_grid for c in row]
        assert [c.u for row in a.cell_grid for c in row] == [c.u for row in b.cell_grid for c in row]
    for x in (g, h, h2):
        x.close()


def test_lattices_py_procedure(ln):
    """tests/lattices.py: a lone IzhikevichNeuronLatticeGPU built from a lattice, electrical"""
    rng = np.random.default_rng(21)
    neuron = ln.IzhikevichNeuron()
    neuron.gap_conductance = 10
    neuron.c_m = 25
    lattice = ln.IzhikevichNeuronLattice(0)
    lattice.populate(neuron, 4, 4)
    lattice.apply_given_position(get_neuron_setup(rng.uniform(neuron.c, neuron.v_th, (4, 4))))
    lattice.connect(lambda x, y: x != y, lambda x, y: 5)
    lattice.update_grid_history = True
    gpu = ln.IzhikevichNeuronLatticeGPU.from_lattice(lattice)
    gpu.electrical_synapse, gpu.chemical_synapse = True, False
    gpu.run_lattice(iterations)
    host = ln.IzhikevichNeuronNetwork.generate_network([lattice], [])
    net = oracle_of(ln, host)
    net.run(iterations, voltage_history=True)
    assert np.array_equal(parity.bits(np.array(gpu.history, np.float32).reshape(iterations, -1)), parity.bits(net.voltage_history))
    assert gpu.get_neuron(2, 3).current_voltage == float(net["current_voltage"][11])
    gpu.close()


def test_dopa_testing_py_procedure(ln):
    """interface_gpu/lixirnet/tests/dopa_testing.py: glutamate and dopamine Rate spike trains (one lattice each) drive a
    lattice whose neurons carry a Glutamate and a Dopamine receptor (s_d1 = 1: D1 scales the NMDA exponent), chemical
    synapses only, dt = 1"""
    rng = np.random.default_rng(77)
    n1, c2 = 4, 2
    T = ln.DopaGluGABANeurotransmitterType
    exc_neuron = ln.IzhikevichNeuron()
    exc_neuron.gap_conductance = 10
    exc_neuron.c_m = 25
    exc_neurotransmitters = {T.Glutamate: ln.BoundedNeurotransmitterKinetics()}
    dopa_neurotransmitters = {T.Dopamine: ln.BoundedNeurotransmitterKinetics()}
    glu, dopa = ln.GlutamateReceptor(), ln.DopamineReceptor()
    dopa.s_d1 = 1
    dopa.s_d2 = 0
    receptors = ln.DopaGluGABA()
    receptors.insert(T.Glutamate, glu)
    receptors.insert(T.Dopamine, dopa)
    exc_neuron.set_synaptic_neurotransmitters(exc_neurotransmitters)
    exc_neuron.set_receptors(receptors)
    exc_spike_train = ln.RateSpikeTrain()
    exc_spike_train.rate = 100
    exc_spike_train.set_synaptic_neurotransmitters(exc_neurotransmitters)
    dopa_spike_train = ln.RateSpikeTrain()
    dopa_spike_train.rate = 100
    dopa_spike_train.set_synaptic_neurotransmitters(dopa_neurotransmitters)
    stl1 = ln.RateSpikeTrainLattice(1)
    stl1.populate(exc_spike_train, n1, n1)
    stl1.apply_given_position(get_spike_train_setup(rng.uniform(0, 100, (n1, n1))))
    stl1.update_grid_history = True
    stl2 = ln.RateSpikeTrainLattice(c2)
    stl2.populate(dopa_spike_train, n1, n1)
    stl2.apply_given_position(get_spike_train_setup(rng.uniform(0, 100, (n1, n1))))
    stl2.update_grid_history = True
    lattice1 = ln.IzhikevichNeuronLattice(e1)
    lattice1.populate(exc_neuron, n1, n1)
    lattice1.apply_given_position(get_neuron_setup(rng.uniform(exc_neuron.c, exc_neuron.v_th, (n1, n1))))
    lattice1.connect(lambda x, y: x != y, lambda x, y: 1)
    lattice1.update_grid_history = True
    network = ln.IzhikevichNeuronNetwork.generate_network([lattice1], [stl1, stl2])
    network.connect(1, e1, lambda x, y: x == y, lambda x, y: 1)
    network.connect(c2, e1, lambda x, y: x == y, lambda x, y: 1)
    network.electrical_synapse = False
    network.chemical_synapse = True
    network.parallel = True
    network.set_dt(1)
    gpu_network = ln.IzhikevichNeuronNetworkGPU.from_network(network)
    net = check_against_oracle(ln, network, gpu_network, (e1,), (1, c2))
    assert not np.all(lc.var(net, "rx_vars", "nmda_modifier") == 1.0)          # the dopamine receptor has acted
    got = gpu_network.get_lattice(e1).get_neuron(1, 2).receptors
    assert got.nmda_modifier == float(lc.var(net, "rx_vars", "nmda_modifier")[6])
    assert got[T.Glutamate].nmda_r.r == float(lc.var(net, "rx_vars", "Glutamate$nmda_r$kinetics$r")[6])
    gpu_network.close()

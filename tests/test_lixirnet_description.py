"""The description of the reference's own GPU Python module (interface_gpu/lixirnet/src/lib.rs:22-79): what the
generator makes of it -- the oracle's stack programs -- against a numpy restatement typed in from the DSL text."""
import numpy as np
import pytest

import lixirnet_case as lc
import oracle_binding as ob
import parity


def build(chemical, electrical, cells, seed=5):
    lay = parity.Layout([(0, 3, 3), (1, 2, 2)], [(4, 3, 3)] if cells else [])
    net = lc.oracle_net(lay, electrical=electrical, chemical=chemical, st_kind=ob.ST_RATE if cells else ob.ST_NONE)
    nn = net.n_neurons
    rng = np.random.default_rng(seed)
    net["current_voltage"] = rng.uniform(-55.0, 30.0, nn).astype(np.float32)
    net["c_m"] = 25.0
    net["gap_conductance"] = 5.0
    net.fill_graph(seed, 0.5, 2.0)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    # glutamate everywhere, GABA on the second lattice, dopamine released by two neurons and felt by the first lattice
    net["nt_flags"][:, lc.GLU] = 1
    net["nt_flags"][9:, lc.GABA] = 1
    net["nt_flags"][[0, 4], lc.DOPA] = 1
    net["rc_flags"][:, lc.GLU] = 1
    net["rc_flags"][:9, lc.DOPA] = 1
    net["rc_flags"][::2, lc.GABA] = 1
    lc.var(net, "rx_vars", "Dopamine$s_d1")[...] = 0.5
    lc.var(net, "rx_vars", "Dopamine$s_d2")[...] = 0.25
    lc.var(net, "nt_custom_vars", "clearance_constant")[...] = 0.05
    if cells:
        net["st_rate"] = rng.uniform(1.0, 6.0, net.n_cells).astype(np.float32)
        net["st_nt_flags"][:, lc.GLU] = 1
        lc.var(net, "st_nt_custom_vars", "clearance_constant")[...] = 0.05
    return net


@pytest.mark.parametrize("chemical,electrical,cells", [(False, True, False), (True, False, False), (True, True, True)])
def test_oracle_program_equals_the_hand_written_restatement(chemical, electrical, cells):
    net = build(chemical, electrical, cells)
    twin = lc.LixirnetTwin(build(chemical, electrical, cells))
    steps = 900
    net.run(steps, voltage_history=True, spike_history=True)
    twin.run(steps)
    assert net.spike_history.sum() >= 10
    assert np.array_equal(twin.spike_history, net.spike_history)
    assert np.array_equal(parity.bits(twin.voltage_history), parity.bits(net.voltage_history))
    for k in ("custom_vars", "rx_vars", "nt_t", "nt_custom_vars", "last_firing_time"):
        assert np.array_equal(parity.bits(twin[k]), parity.bits(net[k])), k
    if chemical:
        glu = lc.var(net, "rx_vars", "Glutamate$current")
        assert np.abs(glu).max() > 0
        assert not np.all(lc.var(net, "rx_vars", "nmda_modifier")[:9] == 1.0)       # dopamine has acted on the first lattice
        assert np.all(lc.var(net, "rx_vars", "nmda_modifier")[9:] == 1.0)           # ... and only there


def test_the_description_reads_as_the_reference_registers_it():
    d = lc.description()
    assert d.neuron.name == "IzhikevichNeuron" and d.receptors.name == "DopaGluGABA"
    assert [t[0] for t in d.receptors.types] == ["Glutamate", "GABA", "Dopamine"]
    assert d.receptors.states == [["ampa_r", "nmda_r"], ["r"], ["r_d1", "r_d2"]]
    assert d.nt_kinetics.name == "BoundedNeurotransmitterKinetics" and d.receptor_kinetics.name == "BoundedReceptorKinetics"
    assert dict(d.neuron.variables)["u"] == 30.0 and d.neuron.mandatory["c_m"] == 100.0

"""GPU parity, Izhikevich lattices with gap junctions only (BASELINE configs[0] and small/ragged cases).

Bit-exact bar: spike raster, voltage history and every state array equal the CPU oracle's bit for bit
(integer and f32 alike -- no tolerance)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(rows, cols, seed, weights="ones", g=10.0, with_diagonal=False):
    lay = parity.Layout([(0, rows, cols)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH)
    n = rows * cols
    net["gap_conductance"] = g
    if n:
        net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    if weights == "ones":
        net.connect_all_to_all(1.0, with_diagonal=with_diagonal)
    else:
        net.fill_graph(seed + 1, 0.5, 1.5, with_diagonal=with_diagonal)
    return net


def run_both(snn, net, steps):
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps)
    net.run(steps, voltage_history=True, spike_history=True)
    return dn


@pytest.mark.parametrize("weights", ["ones", "uniform"])
def test_c1_32x32_1000_steps(snn, weights):
    """BASELINE configs[0]: 32x32, electrical gap junctions only, 1000 steps, dt = 0.1."""
    net = build(32, 32, seed=1, weights=weights)
    dn = run_both(snn, net, 1000)
    raster = dn.spike_history(0)
    assert raster.sum() > 0, "the case must actually spike"
    assert np.array_equal(raster, net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(net.voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    assert dn.clock == net.clock == 1000
    dn.close()


@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 2), (3, 3), (5, 7), (1, 255), (1, 256), (1, 257), (16, 33),
                                       (40, 40)])
def test_ragged_sizes(snn, rows, cols):
    """Sizes around the 256-row chunk, the 64-lane wavefront and the 1024-column tile boundaries."""
    net = build(rows, cols, seed=7 + rows * cols, weights="uniform")
    dn = run_both(snn, net, 200)
    assert np.array_equal(dn.spike_history(0), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(net.voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


def test_sparse_random_connectivity_and_diagonal(snn):
    """80 %-random connectivity as the reference's gpu_accuracy tests draw it (backend/tests/gpu_accuracy.rs:28-32),
    self-connections allowed, None vs Some(0.0) distinguished in the averager."""
    net = build(9, 9, seed=3, weights="uniform", with_diagonal=True)
    rng = np.random.default_rng(5)
    mask = rng.random(net["connections"].shape) < 0.8
    net["connections"][...] = mask
    zero_edges = rng.random(net["connections"].shape) < 0.1
    net["weights"][zero_edges] = 0.0            # Some(0.0): still counted by the averager
    dn = run_both(snn, net, 300)
    assert np.array_equal(dn.spike_history(0), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(net.voltage_history))
    parity.assert_graph_equal(net, dn)
    dn.close()


def test_heterogeneous_parameters_and_resume(snn):
    """Per-neuron parameters (apply_given_position in the reference) and two successive run calls
    resuming from device state (internal_clock persists, gpu_lattices/mod.rs:880)."""
    net = build(12, 12, seed=11, weights="uniform")
    n = net.n_neurons
    net["a"] = ob.uniform_array(21, n, 0.01, 0.1)
    net["b"] = ob.uniform_array(22, n, 0.15, 0.3)
    net["c"] = ob.uniform_array(23, n, -65.0, -50.0)
    net["d"] = ob.uniform_array(24, n, 2.0, 8.0)
    net["gap_conductance"] = ob.uniform_array(25, n, 1.0, 12.0)
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(120)
    dn.run(80)
    net.run(200, voltage_history=True, spike_history=True)
    assert dn.history_steps() == 200
    assert np.array_equal(dn.spike_history(0), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(net.voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


def test_zero_size_and_zero_iterations_are_noops(snn):
    """backend/tests/size_zero_cases.rs: empty lattices and zero iterations run as no-ops."""
    lay = parity.Layout([(0, 0, 0)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH)
    dn = parity.device_from_oracle(snn, net)
    dn.run(10)
    assert dn.clock == 0
    dn.close()
    net = build(3, 3, seed=1)
    dn = parity.device_from_oracle(snn, net)
    dn.run(0)
    assert dn.clock == 0
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.set_synapses(False, False)
    dn.run(5)                      # (false, false) => Ok(()) without stepping, neuron/mod.rs:1217
    assert dn.clock == 0
    dn.close()

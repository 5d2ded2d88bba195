"""GPU parity for the BCM rule (plasticity/mod.rs:72-116) with BCMIzhikevichNeuron (integrate_and_fire/mod.rs:1358-1518)
and BCMPoissonNeuron presynaptic cells (spike_train/mod.rs:835-970): weights, activity bookkeeping and all neuron
state bit-identical to the oracle -- dense and sparse handles, electrical / neurotransmission variants of the rate
formula, STDP and BCM lattices side by side, and the procedure of backend/examples/bcm/main.rs."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(lattices, st, chemical, seed, window=2.0):
    lay = parity.Layout(list(lattices), list(st))
    net = parity.make_oracle(lay, model=ob.BCM_IZHIKEVICH, st_kind=ob.ST_BCM_POISSON if st else ob.ST_NONE,
                             chemical=chemical)
    n, nc = net.n_neurons, net.n_cells
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["bcm_window"] = ob.uniform_array(seed + 1, n, window, 2 * window)      # windows close at different steps
    net["bcm_period"] = np.random.default_rng(seed).integers(2, 6, n).astype(np.uint32)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 2.0
    if nc:
        net["st_chance_of_firing"] = ob.uniform_array(seed + 2, nc, 0.02, 0.2)
        net["st_seed"] = np.arange(50, 50 + nc, dtype=np.uint32)
        net["st_bcm_window"] = ob.uniform_array(seed + 3, nc, window, 2 * window)
        net["st_nt_flags"][:, 0] = 1
    net.fill_graph(seed + 4, 0.5, 1.5)
    rng = np.random.default_rng(seed + 5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    net["plasticity_kind"] = 1
    net["bcm_decay"] = 0.05
    net["bcm_average_scalar"] = 0.5
    return net


def run_both(snn, net, steps, csr=False):
    dn = parity.device_from_oracle(snn, net, csr=csr)
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    w0 = net["weights"].copy()
    net.run(steps, voltage_history=True, spike_history=True)
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert net.spike_history.sum() > 10 and not np.array_equal(w0, net["weights"])
    assert net["bcm_average_activity"].max() > 0
    dn.close()


@pytest.mark.parametrize("chemical", [False, True])
@pytest.mark.parametrize("csr", [False, True])
def test_bcm_lattice_with_bcm_poisson_inputs(snn, chemical, csr):
    net = build([(1, 6, 7)], [(0, 3, 4)], chemical, seed=3)
    run_both(snn, net, 700, csr=csr)
    assert net["st_bcm_average_activity"].max() > 0


def test_stdp_and_bcm_lattices_side_by_side(snn):
    """lattice 0 keeps STDP, lattice 2 follows BCM; edges between them take the rule of the postsynaptic lattice"""
    net = build([(0, 5, 5), (2, 6, 6)], [(7, 2, 3)], False, seed=9)
    net["plasticity_kind"][0] = 0
    net["stdp_a_plus"][0] = 0.5
    run_both(snn, net, 700)


def test_procedure_of_the_bcm_example(snn):
    """backend/examples/bcm/main.rs:61-86: BCMPoissonNeuron spike trains (one chance_of_firing each) all connected to
    ONE BCMIzhikevichNeuron (c_m 50, gap_conductance 5) whose lattice carries BCM::default()."""
    lay = parity.Layout([(1, 1, 1)], [(0, 2, 1)])
    net = parity.make_oracle(lay, model=ob.BCM_IZHIKEVICH, st_kind=ob.ST_BCM_POISSON)
    net["c_m"] = 50.0
    net["gap_conductance"] = 5.0
    net["st_chance_of_firing"] = np.array([0.25, 0.125], np.float32)       # denser than main.rs:125 to keep the run short
    net["st_seed"] = np.array([11, 12], np.uint32)
    net["bcm_window"] = 5.0
    net["st_bcm_window"] = 5.0
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = np.array([1.45, 1.62], np.float32)
    net["do_plasticity"] = 1
    net["plasticity_kind"] = 1
    dn = parity.device_from_oracle(snn, net)
    dn.run(3000)
    w0 = net["weights"].copy()
    net.run(3000)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert not np.array_equal(w0, net["weights"]) and net["bcm_num_spikes"][0] > 0
    dn.close()


def test_bcm_errors(snn):
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
    dn.add_lattice(0, 2, 2)
    dn.finalize()
    with pytest.raises(snn.SnnError):
        dn.set_bcm(0)                              # needs BCMActivity neurons
    dn.close()
    dn = snn.DeviceNetwork(model=snn.BCM_IZHIKEVICH)
    dn.add_lattice(0, 2, 2)
    dn.finalize(0, 2)
    with pytest.raises(snn.SnnError):
        dn.set_bcm(0)                              # shard handle
    dn.close()

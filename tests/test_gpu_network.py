"""GPU parity for LatticeNetwork semantics: several lattices in the interleaved index space, Poisson /
Rate spike-train lattices as presynaptic drivers (gap-junction effect + neurotransmitter release), and
deferred STDP within and across lattices (BASELINE configs[3] at test size)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def run_both(snn, net, steps, chunks=1):
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    for _ in range(chunks):
        dn.run(steps // chunks)
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=True)
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count]), f"raster of lattice {i}"
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    for i, _, _ in net.layout.st_lattices:
        first, count, _ = rng[i]
        assert np.array_equal(parity.bits(dn.voltage_history(i)),
                              parity.bits(net.st_voltage_history[:, first:first + count])), f"spike train {i}"
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert dn.clock == net.clock
    return dn


def test_single_lattice_stdp(snn):
    """STDP on one lattice (deferred form): weights after 600 steps bit-identical."""
    lay = parity.Layout([(0, 8, 8)])
    net = parity.make_oracle(lay)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(1, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net.fill_graph(2, 0.5, 1.5)
    net["do_plasticity"] = 1
    w0 = net["weights"].copy()
    dn = run_both(snn, net, 600, chunks=3)
    assert net.spike_history.sum() > 10 and not np.array_equal(w0, net["weights"])
    dn.close()


def test_c4_small_excitatory_inhibitory_network_with_stdp(snn):
    """configs[3] at test size (backend/examples/interacting_pools/main.rs:22-46 scaled down):
    exc 12x12 (id 1, w = +1), inh 6x6 (id 0, w = -1), connect(0 -> 1, w = -1), connect(1 -> 0, w = +1),
    STDP defaults on both, different plasticity parameters per lattice."""
    lay = parity.Layout([(0, 6, 6), (1, 12, 12)])
    net = parity.make_oracle(lay)
    r = lay.ranges()
    inh = slice(r[0][0], r[0][0] + r[0][1])
    exc = slice(r[1][0], r[1][0] + r[1][1])
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(4, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["connections"][...] = 1
    net["connections"][np.arange(n), np.arange(n)] = 0
    rng = np.random.default_rng(4)
    net["connections"][rng.random((n, n)) < 0.25] = 0
    mag = ob.uniform_array(5, n * n, 0.5, 1.5).reshape(n, n)
    net["weights"][...] = mag
    net["weights"][inh, :] *= -1.0          # everything leaving the inhibitory pool is negative
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    net["stdp_a_plus"][1] = 1.5
    net["stdp_tau_minus"][0] = 3.0
    dn = run_both(snn, net, 500, chunks=2)
    assert net.spike_history.sum() > 10
    dn.close()


@pytest.mark.parametrize("st_kind", [ob.ST_POISSON, ob.ST_RATE, ob.ST_PRESET])
@pytest.mark.parametrize("synapses", [(True, False), (True, True), (False, True)])
def test_spike_train_lattice_drives_neurons(snn, st_kind, synapses):
    """Spike-train lattice (id 0) -> neuron lattice (id 1), one-to-one plus some random extra edges;
    internal all-to-all gap junctions; STDP on (spike-train presynaptic cells allowed, mod.rs:2326-2331)."""
    lay = parity.Layout([(1, 5, 6)], [(0, 5, 6)])
    net = parity.make_oracle(lay, st_kind=st_kind, electrical=synapses[0], chemical=synapses[1])
    nn, nc = net.n_neurons, net.n_cells
    net["current_voltage"] = ob.uniform_array(6, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 3.0
    net["st_nt_flags"][:, 0] = 1
    net["st_nt_flags"][::2, 2] = 1          # some cells also release GABA nobody listens to
    net["st_refractoriness"][1::3] = 1      # every third cell: ExponentialDecayRefractoriness (spike_train/mod.rs:164-178)
    net["st_k"][1::3] = 200.0
    if st_kind == ob.ST_POISSON:
        net["st_chance_of_firing"] = ob.uniform_array(7, nc, 0.0, 0.05)
        net["st_seed"] = np.arange(100, 100 + nc, dtype=np.uint32)
    elif st_kind == ob.ST_PRESET:
        # 0..3 firing times per cell (PresetSpikeTrain, spike_train/mod.rs:753-833); cell 0 has none
        r = np.random.default_rng(7)
        net.set_firing_times([list(r.uniform(0.5, 6.0, int(k))) for k in [0] + list(r.integers(1, 4, nc - 1))])
    else:
        net["st_rate"] = ob.uniform_array(7, nc, 2.0, 9.0)
        net["st_rate"][0] = 0.0             # rate 0 never fires (rate_spike_train.rs:44)
    net.fill_graph(8, 0.5, 1.5)
    rng = np.random.default_rng(8)
    block = net["connections"][nn:, :]
    block[...] = rng.random(block.shape) < 0.1
    block[np.arange(nc), np.arange(nn)] = 1
    net["weights"][nn:, :] = block * 2.0
    net["do_plasticity"] = 1
    dn = run_both(snn, net, 1000, chunks=2)
    assert net["st_last_firing_time"].max() > 0, "spike trains must fire"
    assert net.spike_history.sum() > 0
    dn.close()


def test_two_spike_train_lattices_and_resume(snn):
    """Two spike-train lattices (each with its own internal clock, neuron/mod.rs:1391) feeding one neuron
    lattice; three successive run calls resume clocks and last_firing_time consistently."""
    lay = parity.Layout([(2, 4, 4)], [(0, 2, 2), (5, 1, 3)])
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE)
    nn, nc = net.n_neurons, net.n_cells
    net["st_rate"] = ob.uniform_array(3, nc, 1.0, 4.0)
    net["connections"][nn:, :] = 1
    net["weights"][nn:, :] = 1.0
    dn = run_both(snn, net, 300, chunks=3)
    assert net["st_last_firing_time"].min() > 0 and list(net["st_clock"]) == [300, 300]
    dn.close()


def test_many_spike_train_rows_few_neurons(snn):
    """More than 65 535 presynaptic rows into a handful of neurons (graph import in row hops; chunked sum over a
    long column): bit-identical inputs."""
    lay = parity.Layout([(0, 1, 3)], [(1, 300, 256)])          # 76 800 spike-train cells
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE)
    nn, nc = net.n_neurons, net.n_cells
    net["st_rate"] = ob.uniform_array(1, nc, 0.5, 30.0)
    rng = np.random.default_rng(2)
    net["connections"][nn:, :] = rng.random((nc, nn)) < 0.5
    net["weights"][nn:, :] = ob.uniform_array(3, nc * nn, 0.0, 1.0).reshape(nc, nn) * net["connections"][nn:, :]
    net["connections"][:nn, :] = 1
    net["weights"][:nn, :] = 1.0
    net.n_threads = 8
    dn = run_both(snn, net, 60)
    assert net["st_last_firing_time"].max() > 0
    dn.close()

"""Cross-check the C oracle against the independent numpy float32 restatement (tests/numpy_ref.py)."""
import numpy as np
import pytest

import numpy_ref as nr
import oracle_binding as ob


def _state(net, names):
    return {k: net[k].copy() for k in names}


@pytest.mark.parametrize("n,seed", [(4, 1), (16, 2), (37, 3), (300, 4)])
def test_izhikevich_lattice_matches_numpy(n, seed):
    net = ob.Net(n)
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net.fill_graph(seed + 10, 0.5, 1.5)
    s = _state(net, ["current_voltage", "w_value", "a", "b", "c", "d", "v_th", "tau_m", "c_m", "dt"])
    steps = 700 if n < 100 else 320       # long enough for the first spikes (and resets) to occur
    vh, sh, lft = nr.run_lattice(nr.izhikevich_step, s, net["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), steps)
    net.run(steps, voltage_history=True, spike_history=True)
    assert sh.sum() > 0
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))
    assert np.array_equal(lft, net["last_firing_time"])


def test_lif_lattice_matches_numpy():
    n = 25
    net = ob.Net(n, model=ob.LIF)
    net["current_voltage"] = ob.uniform_array(5, n, -80.0, -50.0)
    net["gap_conductance"] = 10.0
    net["tref"] = 1.0
    net.fill_graph(6, 0.5, 1.5)
    s = _state(net, ["current_voltage", "refractory_count", "leak_constant", "integration_constant", "e_l", "g_l",
                     "tau_m", "dt", "v_th", "v_reset", "tref"])
    vh, sh, lft = nr.run_lattice(nr.lif_step, s, net["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), 150)
    net.run(150, voltage_history=True, spike_history=True)
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))


def test_qif_lattice_matches_numpy():
    n = 20
    net = ob.Net(n, model=ob.QIF)
    net["current_voltage"] = ob.uniform_array(7, n, -75.0, -56.0)
    net["gap_conductance"] = 3.0
    net["tref"] = 0.7
    net["tau_m"] = 10.0
    net.fill_graph(8, 0.5, 1.5)
    s = _state(net, ["current_voltage", "refractory_count", "qif_alpha", "qif_v_c", "integration_constant", "tau_m",
                     "dt", "v_th", "v_reset", "tref"])
    vh, sh, lft = nr.run_lattice(nr.qif_step, s, net["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), 400)
    net.run(400, voltage_history=True, spike_history=True)
    assert sh.sum() > 0
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))


def test_simple_lif_lattice_matches_numpy():
    n = 20
    net = ob.Net(n, model=ob.SIMPLE_LIF)
    net["current_voltage"] = ob.uniform_array(9, n, -75.0, -56.0)
    net["slif_g"] = 0.5                       # positive feedback so that cells cross threshold
    net["slif_e"] = -76.0
    net.fill_graph(10, 0.5, 1.5)
    s = _state(net, ["current_voltage", "slif_g", "slif_e", "dt", "v_th", "v_reset"])
    vh, sh, lft = nr.run_lattice(nr.simple_lif_step, s, net["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), 400)
    net.run(400, voltage_history=True, spike_history=True)
    assert sh.sum() > 0
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))


@pytest.mark.parametrize("exponential", [False, True])
def test_adaptive_lif_lattice_matches_numpy(exponential):
    n = 20
    net = ob.Net(n, model=ob.ADAPTIVE_EXP_LIF if exponential else ob.ADAPTIVE_LIF)
    net["current_voltage"] = ob.uniform_array(11, n, -75.0, -56.0)
    net["gap_conductance"] = 3.0
    net["tref"] = 0.7
    net["adp_beta"] = 3.0
    net["leak_constant"] = 1.0                # positive feedback away from e_l so that cells cross threshold
    net["c_m"] = 1.0
    net["v_reset"] = -73.0
    names = ["current_voltage", "w_value", "refractory_count", "leak_constant", "integration_constant", "e_l", "g_l",
             "tau_m", "c_m", "dt", "v_th", "v_reset", "tref", "adp_alpha", "adp_beta"]
    if exponential:
        net["slope_factor"] = 2.0
        names.append("slope_factor")
    net.fill_graph(12, 0.5, 1.5)
    s = _state(net, names)
    expf = np.vectorize(ob.expf, otypes=[np.float32]) if exponential else None
    vh, sh, lft = nr.run_lattice(lambda st, i: nr.adaptive_step(st, i, expf), s, net["gap_conductance"].copy(),
                                 net["weights"].copy(), net["connections"].copy(), 600)
    net.run(600, voltage_history=True, spike_history=True)
    assert sh.sum() > 0
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))
    assert np.array_equal(s["w_value"].view(np.uint32), net["w_value"].view(np.uint32))


def test_leaky_izhikevich_lattice_matches_numpy():
    n = 20
    net = ob.Net(n, model=ob.LEAKY_IZHIKEVICH)
    net["current_voltage"] = ob.uniform_array(13, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["w_value"] = 0.5
    net.fill_graph(14, 0.5, 1.5)
    s = _state(net, ["current_voltage", "w_value", "a", "b", "c", "d", "e_l", "tau_m", "c_m", "dt", "v_th"])
    vh, sh, lft = nr.run_lattice(nr.leaky_izhikevich_step, s, net["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), 600)
    net.run(600, voltage_history=True, spike_history=True)
    assert sh.sum() > 0
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))
    assert np.array_equal(s["w_value"].view(np.uint32), net["w_value"].view(np.uint32))


def test_reduced_histories_against_numpy():
    """lattice_summaries (oracle) vs the voltage history reduced with numpy in the same chunked order, and the
    spike totals vs the column sums of the raster (SpikeHistory::aggregate, neuron/mod.rs:331-360)."""
    import parity
    lay = parity.Layout([(0, 5, 5), (4, 24, 24)])
    net = parity.make_oracle(lay)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(3, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net.fill_graph(4, 0.5, 1.5)
    net.run(300, voltage_history=True, spike_history=True, summaries=True, spike_counts=True)
    assert net.spike_history.sum() > 10
    assert np.array_equal(net.spike_counts, net.spike_history.sum(axis=0).astype(np.uint32))
    f = np.float32
    for slot, (i, r, c) in enumerate(lay.lattices):
        first, count, _ = lay.ranges()[i]
        v = net.voltage_history[:, first:first + count]
        avg = np.zeros(300, f)
        eeg = np.zeros(300, f)
        for t in range(300):
            tot, tot_e = f(0), f(0)
            for c0 in range(0, count, 256):
                p, pe = f(0), f(0)
                for x in v[t, c0:c0 + 256]:
                    p = f(p + x)
                    pe = f(pe + f(x - f(0.007)))
                tot, tot_e = f(tot + p), f(tot_e + pe)
            avg[t] = f(tot / f(count))
            k = f(f(1) / f(f(f(f(4) * f(np.pi)) * f(251.0)) * f(0.8)))
            eeg[t] = f(k * tot_e)
        assert np.array_equal(avg.view(np.uint32), net.avg_history[:, slot].view(np.uint32))
        assert np.array_equal(eeg.view(np.uint32), net.eeg_history[:, slot].view(np.uint32))


@pytest.mark.parametrize("n,block,threads", [(700, 64, 3), (1030, 1024, 4), (300, 1024, 1)])
def test_tiled_all_core_inputs_equal_the_general_routine(n, block, threads):
    """bench.py's cpu_baseline streams the matrix in (1024 columns x 256 rows) tiles over all cores: same sums bit for
    bit as the general routine (ragged last chunk / last block, masked edges, column windows)."""
    net = ob.Net(n, model=ob.IZHIKEVICH)
    net["gap_conductance"] = ob.uniform_array(2, n, 1.0, 12.0)
    net["current_voltage"] = ob.uniform_array(1, n, -65.0, 30.0)
    net.fill_graph(3, -0.5, 1.5)
    rng = np.random.default_rng(n)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["connections"][:, 5] = 0                      # a column without any edge: averager 1
    net.n_threads = threads
    net.inputs()
    want = net["input_current"].copy()
    net["input_current"][...] = np.float32(np.nan)
    q0, q1 = 17, n - 3
    net.inputs_tiled(q0, q1, block)
    got = net["input_current"]
    assert np.array_equal(got[q0:q1].view(np.uint32), want[q0:q1].view(np.uint32))
    assert np.isnan(got[:q0]).all() and np.isnan(got[q1:]).all()

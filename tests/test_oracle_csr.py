"""The oracle's sparse entry points (snn_o_inputs_csr / snn_o_run_csr, the form BASELINE configs[4] needs) against its
dense routines and the numpy restatement on the same graphs; and plasticity on a column WINDOW of the matrix (the form
the full-size teacher-forced checks use) against plasticity on the whole matrix."""
import numpy as np
import pytest

import numpy_net
import oracle_binding as ob
import parity


def sparse_case(seed, chemical):
    lay = parity.Layout([(0, 9, 31), (2, 5, 7)], [(1, 3, 11)])            # 314 neurons (two chunks), 33 Poisson cells
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON, chemical=chemical, nt_kind=ob.NT_APPROX, rc_kind=ob.RC_DESTEXHE)
    rng = np.random.default_rng(seed)
    nn, nt = net.n_neurons, net.n_tot
    net["current_voltage"] = ob.uniform_array(seed, nn, -65.0, 30.0)
    net["gap_conductance"] = ob.uniform_array(seed + 1, nn, 5.0, 12.0)
    conn = rng.random((nt, nn)) < 0.04
    conn[:, 5] = False                                                    # a neuron without any input
    conn[:256, 7] = False                                                 # a row whose first chunk is empty
    net["connections"][...] = conn
    net["weights"][...] = np.where(conn, rng.uniform(-1.0, 1.5, (nt, nn)), 0.0).astype(np.float32)
    net["st_chance_of_firing"] = 0.05
    if chemical:
        net["nt_flags"][...] = rng.random((nn, 3)) < 0.6
        net["st_nt_flags"][...] = rng.random((net.n_cells, 3)) < 0.6
        net["rc_flags"][...] = rng.random((nn, 3)) < 0.7
    return net


@pytest.mark.parametrize("chemical", [False, True])
def test_csr_inputs_and_run_equal_the_dense_oracle_and_numpy(chemical):
    dense = sparse_case(11, chemical)
    sparse = sparse_case(11, chemical)
    twin = numpy_net.NumpyNet(sparse_case(11, chemical))
    ptr, pre, w = parity.csr_from_dense(dense, 0, dense.n_neurons)
    sparse.n_threads = 3
    steps = 120
    dense.run(steps, voltage_history=True, spike_history=True)
    sparse.run_csr(ptr, pre, w, steps, voltage_history=True, spike_history=True)
    twin.run(steps)
    assert dense.spike_history.sum() > 10
    assert np.array_equal(sparse.spike_history, dense.spike_history)
    assert np.array_equal(parity.bits(sparse.voltage_history), parity.bits(dense.voltage_history))
    assert np.array_equal(twin.spike_history, dense.spike_history)
    assert np.array_equal(parity.bits(twin.voltage_history), parity.bits(dense.voltage_history))
    for k in ("current_voltage", "w_value", "last_firing_time", "st_seed", "st_last_firing_time", "nt_t", "rc_r",
              "rc_current", "st_nt_t"):
        assert np.array_equal(parity.bits(sparse[k]), parity.bits(dense[k])), k
        assert np.array_equal(parity.bits(twin[k]), parity.bits(dense[k])), k
    assert sparse.clock == dense.clock == steps


def test_csr_inputs_on_a_range_of_posts():
    net = sparse_case(12, True)
    ptr, pre, w = parity.csr_from_dense(net, 0, net.n_neurons)
    net.inputs()
    want = {k: net[k].copy() for k in ("input_current", "input_t", "input_count")}
    for k in want:
        net[k] = -7.0
    net.inputs_csr(ptr, pre, w, 100, 300)
    for k in want:
        assert np.array_equal(parity.bits(net[k][100:300]), parity.bits(want[k][100:300])), k
        assert (net[k][:100] == -7.0).all() and (net[k][300:] == -7.0).all()


def test_plasticity_on_a_column_window_equals_the_whole_matrix():
    lay = parity.Layout([(0, 6, 8), (1, 5, 5)], [(2, 2, 3)])
    nets = []
    for _ in range(2):
        net = parity.make_oracle(lay, st_kind=ob.ST_POISSON)
        nn = net.n_neurons
        net.fill_graph(3, 0.5, 1.5)
        net["connections"][::5, ::3] = 0
        net["do_plasticity"] = 1
        net["stdp_a_plus"][0] = 1.5
        net["stdp_tau_minus"][1] = 3.0
        rng = np.random.default_rng(4)
        net["last_firing_time"] = rng.integers(-1, 40, nn)
        net["st_last_firing_time"] = rng.integers(-1, 40, net.n_cells)
        net["is_spiking"] = rng.random(nn) < 0.2
        net["last_firing_time"][net["is_spiking"] != 0] = 40
        nets.append(net)
    full, win = nets
    full.plasticity()
    c0, c1 = 17, 60
    w_all, c_all = win["weights"].copy(), win["connections"].copy()
    win.arr["weights"] = np.ascontiguousarray(w_all[:, c0:c1])
    win.arr["connections"] = np.ascontiguousarray(c_all[:, c0:c1])
    win.w_col0, win.w_ld = c0, c1 - c0
    win.plasticity(c0, c1)
    assert not np.array_equal(full["weights"][:, c0:c1], w_all[:, c0:c1])          # something was updated
    assert np.array_equal(parity.bits(win["weights"]), parity.bits(full["weights"][:, c0:c1]))

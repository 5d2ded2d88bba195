"""The HIP stepper against the COMMITTED golden vectors (no oracle run involved): raster, voltage traces,
final state and the case-specific arrays (gates, weights, receptor state, RNG seeds) bit for bit."""
import os

import numpy as np
import pytest

import golden_cases
import parity

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_hip_matches_golden(snn, name):
    net, steps = golden_cases.CASES[name]()          # inputs only; the oracle is not stepped
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    if getattr(net, "custom_model", None) is not None:          # a generated model: its library (hipcc, cached)
        net.custom_lib = snn._lib.build_custom(net.custom_model)
    dn = parity.device_from_oracle(snn, net)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot]:
            dn.set_reward_modulator(i, *(float(net[k][slot]) for k in (
                "rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")))
            dn.set_trace_rows(0, net["traces"])
    parity.push_connection_kinds(dn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps)
    ids = [i for i, _, _ in net.layout.lattices]          # the fixtures hold all lattices side by side, in layout order
    raster = np.concatenate([dn.spike_history(i) for i in ids], axis=1)
    assert np.packbits(raster, axis=1).tobytes() == want["raster"].tobytes()
    stride = int(want["trace_stride"])
    trace = np.concatenate([dn.voltage_history(i) for i in ids], axis=1)
    assert np.ascontiguousarray(trace[::stride]).tobytes() == want["voltage_trace"].tobytes()
    st = parity.pull_state(dn, net)
    assert st["current_voltage"].tobytes() == want["final_voltage"].tobytes()
    assert st["last_firing_time"].tobytes() == want["last_firing_time"].tobytes()
    if net.n_cells:
        sid = net.layout.st_lattices[0][0]
        assert np.packbits(dn.voltage_history(sid) > 0, axis=1).tobytes() == want["st_voltage_spikes"].tobytes()
    for k in golden_cases.EXTRA.get(name, ()):
        if k == "weights":
            w, c = dn.get_graph_rows(0, net.n_tot)
            assert w.tobytes() == np.where(net["connections"] != 0, want[k], np.float32(0)).astype(np.float32).tobytes()
        elif k == "traces":
            assert dn.get_trace_rows(0, net.n_tot).tobytes() == want[k].tobytes()
        elif k == "pending":
            assert dn.get_pending_rows(0, net.n_tot).tobytes() == want[k].tobytes()
        elif k == "edge_counter":
            assert dn.get_counter_rows(0, net.n_tot).tobytes() == want[k].tobytes()
        else:
            assert st[k].tobytes() == want[k].tobytes(), k
    dn.close()


def test_sparse_handle_matches_the_golden_reward_network(snn):
    """the committed vectors of `reward_modulated_network` (connections of kind 1 / 2 between lattices, written by the numpy twin)
    against a SPARSE handle: k_reward_cross_csr, trace / dw / counter per stored edge"""
    name = "reward_modulated_network"
    net, steps = golden_cases.CASES[name]()
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    dn = parity.device_from_oracle(snn, net, csr=True)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot]:
            dn.set_reward_modulator(i, *(float(net[k][slot]) for k in (
                "rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")))
    dn.set_traces_csr(parity.csr_values(net, net["traces"], dn.owned))
    parity.push_connection_kinds(dn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps)
    ids = [i for i, _, _ in net.layout.lattices]
    raster = np.concatenate([dn.spike_history(i) for i in ids], axis=1)
    assert np.packbits(raster, axis=1).tobytes() == want["raster"].tobytes()
    st = parity.pull_state(dn, net)
    assert st["current_voltage"].tobytes() == want["final_voltage"].tobytes()
    assert st["last_firing_time"].tobytes() == want["last_firing_time"].tobytes()
    posts = dn.owned
    assert dn.get_graph_csr().tobytes() == parity.csr_values(net, want["weights"], posts).astype(np.float32).tobytes()
    assert dn.get_traces_csr().tobytes() == parity.csr_values(net, want["traces"], posts).tobytes()
    assert dn.get_pending_csr().tobytes() == parity.csr_values(net, want["pending"], posts).tobytes()
    assert dn.get_counters_csr().tobytes() == parity.csr_values(net, want["edge_counter"], posts).tobytes()
    dn.close()

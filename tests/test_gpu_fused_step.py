"""The one-launch steps (k_step_resident: small dense lattices, n_tot <= 1024 presynaptic rows; k_step_csr: sparse
handles; both unsharded) against the two-kernel step (k_inputs_dense + k_update, forced with SNN_AMD_FUSED_STEP=0) and the oracle: all three
bit-identical, for every model family, synapse kind, spike-train rows, ragged sizes and split runs.  Every other
small-lattice GPU test already runs through the fused step; this file keeps the two-kernel path covered at the
same sizes."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(model, electrical, chemical, lattices, st, seed):
    lay = parity.Layout(lattices, st)
    net = parity.make_oracle(lay, model=model, st_kind=ob.ST_POISSON if st else ob.ST_NONE, electrical=electrical,
                             chemical=chemical)
    n, nc = net.n_neurons, net.n_cells
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.QIF: (-75, -56)}[model]
    net["current_voltage"] = ob.uniform_array(seed, n, lo, hi)
    net["gap_conductance"] = 10.0 if model == ob.IZHIKEVICH else 3.0
    if model in (ob.LIF, ob.QIF):
        net["tref"] = ob.uniform_array(seed + 1, n, 0.3, 1.5)
        net["tau_m"] = 10.0
    rng = np.random.default_rng(seed)
    net["nt_flags"][...] = rng.random((n, 3)) < 0.6
    net["nt_flags"][:, 0] = 1                      # AMPA everywhere (the uniform fast path), the others mixed
    net["rc_flags"][...] = rng.random((n, 3)) < 0.7
    net["nt_t"][...] = rng.random((n, 3)).astype(np.float32) * net["nt_flags"]
    if nc:
        net["st_nt_flags"][...] = rng.random((nc, 3)) < 0.5
        net["st_chance_of_firing"] = ob.uniform_array(seed + 2, nc, 0.0, 0.05)
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
    net.fill_graph(seed + 3, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.25] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    return net


def run_device(snn, net, steps, fused, csr=False, quarters=True):
    """quarters: the one-launch step with a chunk's rows over four wavefronts (k_step_resident_q, the default up to two chunks)"""
    old = os.environ.get("SNN_AMD_FUSED_STEP")
    os.environ["SNN_AMD_FUSED_STEP"] = "1" if fused else "0"
    try:
        dn = parity.device_from_oracle(snn, net, csr=csr)
    finally:
        if old is None:
            del os.environ["SNN_AMD_FUSED_STEP"]
        else:
            os.environ["SNN_AMD_FUSED_STEP"] = old
    dn.set_option("resident_quarters", int(quarters))
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps // 3)
    dn.run(steps - steps // 3)
    out = {"state": parity.pull_state(dn, net), "one_launch_steps": dn.stat("steps_dense_one_launch"),
           "graph": (dn.get_graph_csr(),) if csr else dn.get_graph_rows(0, net.n_tot)}
    for i, _, _ in net.layout.lattices:
        out[("v", i)] = dn.voltage_history(i)
        out[("s", i)] = dn.spike_history(i)
    dn.close()
    return out


CASES = [
    (ob.IZHIKEVICH, True, False, [(0, 1, 1)], [], 1),                       # a single neuron
    (ob.IZHIKEVICH, True, False, [(0, 7, 9)], [], 2),                       # one ragged 63-row block
    (ob.IZHIKEVICH, True, False, [(0, 16, 17)], [], 3),                     # two chunks, ragged second
    (ob.IZHIKEVICH, True, False, [(0, 32, 32)], [], 4),                     # BASELINE configs[0]: four full chunks
    (ob.IZHIKEVICH, True, True, [(0, 9, 10), (2, 11, 12)], [(5, 6, 7)], 5),  # two lattices + Poisson rows, both synapse kinds
    (ob.LIF, False, True, [(0, 12, 13)], [(3, 4, 5)], 6),                   # chemical only
    (ob.HH, True, True, [(0, 10, 10)], [], 7),
    (ob.QIF, True, False, [(1, 20, 20)], [(0, 10, 10)], 8),                 # spike-train rows straddle a 64-row block
]


@pytest.mark.parametrize("model,electrical,chemical,lattices,st,seed", CASES)
def test_fused_step_equals_two_kernel_step_and_oracle(snn, model, electrical, chemical, lattices, st, seed):
    net = build(model, electrical, chemical, lattices, st, seed)
    assert net.n_tot <= 1024
    steps = 300 if (model != ob.HH and (chemical or model != ob.IZHIKEVICH)) else 900
    a = run_device(snn, net, steps, fused=True)
    b = run_device(snn, net, steps, fused=False)
    c = run_device(snn, net, steps, fused=True, quarters=False)          # one wavefront per chunk (k_step_resident)
    assert a["one_launch_steps"] == c["one_launch_steps"] and b["one_launch_steps"] == 0
    for other in (b, c):
        for key in a:
            if key in ("state", "graph", "one_launch_steps"):
                continue
            assert np.array_equal(parity.bits(a[key]), parity.bits(other[key])), key
        for name in a["state"]:
            assert np.array_equal(parity.bits(a["state"][name]), parity.bits(other["state"][name])), name
        assert np.array_equal(parity.bits(a["graph"][0]), parity.bits(other["graph"][0]))
    net.run(steps, voltage_history=True, spike_history=True)
    parity.assert_state_equal(net, a["state"])
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(a[("s", i)], net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(a[("v", i)]), parity.bits(net.voltage_history[:, first:first + count]))
    assert net.spike_history.sum() > 0


@pytest.mark.parametrize("model,electrical,chemical,lattices,st,seed", [CASES[4], CASES[5], CASES[7]])
def test_fused_sparse_step_equals_two_kernel_step_and_oracle(snn, model, electrical, chemical, lattices, st, seed):
    """Sparse (SELL-64) handles: k_step_csr (row sums + neuron update in one launch) against k_inputs_csr + k_update."""
    net = build(model, electrical, chemical, lattices, st, seed)
    steps = 300
    a = run_device(snn, net, steps, fused=True, csr=True)
    b = run_device(snn, net, steps, fused=False, csr=True)
    for key in a:
        if key in ("state", "graph", "one_launch_steps"):
            continue
        assert np.array_equal(parity.bits(a[key]), parity.bits(b[key])), key
    for name in a["state"]:
        assert np.array_equal(parity.bits(a["state"][name]), parity.bits(b["state"][name])), name
    assert np.array_equal(parity.bits(a["graph"][0]), parity.bits(b["graph"][0]))
    net.run(steps, voltage_history=True, spike_history=True)
    parity.assert_state_equal(net, a["state"])
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(a[("s", i)], net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(a[("v", i)]), parity.bits(net.voltage_history[:, first:first + count]))
    assert net.spike_history.sum() > 0


def test_attribute_writes_between_fused_steps(snn):
    """The fused step reads S(t) from a shadow of the exchange buffer: a host write between steps must reach it."""
    net = build(ob.IZHIKEVICH, True, True, [(0, 8, 8)], [], 9)
    dn = parity.device_from_oracle(snn, net)
    dn.run(40)
    net.run(40)
    v = ob.uniform_array(77, net.n_neurons, -65.0, 30.0)
    t = np.zeros((net.n_neurons, 3), np.float32)
    t[:, 0] = 0.5
    dn.set_attr(0, "current_voltage", v)
    dn.set_attr(0, "neurotransmitters$t", t)
    net["current_voltage"] = v
    net["nt_t"] = t
    dn.run(40)
    net.run(40)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()

"""k_inputs_dense_close (round 5): on streamed dense matrices the last workgroup of a column tile updates the tile's neurons inside
the input pass's launch.  Against the two-kernel step (option "dense_close" 0) and the oracle: bit-identical, for the two streamed
shapes' column tiles (ragged last tile included), electrical and electrical + chemical passes (one live transmitter type: the
specialised pass; three: the generic one), spike-train rows, STDP after the step, split runs with a host write in between (the
shadows of the exchanged state are rebuilt)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity
from test_gpu_fused_step import build

pytestmark = pytest.mark.gpu

# (model, chemical, lattices, cells, one transmitter type only, plastic, seed)
CASES = [
    (ob.IZHIKEVICH, False, [(0, 65, 65)], [], False, False, 11),                       # 4 225 neurons: 9 tiles of 512, the last ragged
    (ob.IZHIKEVICH, False, [(0, 1, 4099)], [(3, 2, 3)], False, True, 12),              # just past the cache-resident sizes; cells; STDP
    (ob.HH, True, [(0, 50, 50), (2, 41, 42)], [(5, 8, 8)], True, False, 13),           # configs[2]'s kind: HH + AMPA only
    (ob.LIF, True, [(1, 40, 60), (4, 43, 43)], [(0, 5, 7)], False, True, 14),          # three live types, two lattices, STDP
]


def run_device(snn, net, steps, close, one_type):
    dn = parity.device_from_oracle(snn, net)
    dn.set_option("dense_close", 1 if close else 0)
    dn.set_history(voltage=True, spikes=True)
    first = steps // 3
    dn.run(first)
    i0 = net.layout.lattices[0][0]
    g = dn.get_attr(i0, "gap_conductance")                     # a host write between two run calls: same values, new shadows
    dn.set_attr(i0, "gap_conductance", g)
    v = dn.get_attr(i0, "current_voltage")
    dn.set_attr(i0, "current_voltage", v)
    dn.run(steps - first)
    out = {"state": parity.pull_state(dn, net), "graph": dn.get_graph_rows(0, net.n_tot),
           "closed": dn.stat("steps_dense_close"), "two": dn.stat("steps_two_kernel")}
    for i, _, _ in net.layout.lattices:
        out[("v", i)] = dn.voltage_history(i)
        out[("s", i)] = dn.spike_history(i)
    dn.close()
    return out


@pytest.mark.parametrize("model,chemical,lattices,st,one_type,plastic,seed", CASES)
def test_the_closing_pass_equals_the_two_kernel_step_and_the_oracle(snn, model, chemical, lattices, st, one_type, plastic, seed):
    net = build(model, True, chemical, lattices, st, seed)
    assert net.n_tot > 4096
    if one_type:
        net["nt_flags"][:, 1:] = 0
        net["nt_t"][:, 1:] = 0
        if net.n_cells:
            net["st_nt_flags"][:, 1:] = 0
            net["st_nt_flags"][:, 0] = 1
    net["do_plasticity"] = 1 if plastic else 0
    steps = 36
    a = run_device(snn, net, steps, True, one_type)
    b = run_device(snn, net, steps, False, one_type)
    assert a["closed"] == steps and a["two"] == 0 and b["closed"] == 0 and b["two"] == steps
    for key in a:
        if key in ("state", "graph", "closed", "two"):
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
    want = np.where(net["connections"] != 0, net["weights"], np.float32(0))
    assert np.array_equal(parity.bits(a["graph"][0]), parity.bits(want))

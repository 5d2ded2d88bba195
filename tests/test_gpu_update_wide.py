"""The wide neuron update of dense handles with chemical synapses (k_update_wide, option "update_all_planes" 2 -- the default): four
wavefronts share a column's chunk partials, the running sums pass from one to the next in canonical order, the others warm the
cache.  Bit for bit the two older forms (0: plane after plane, 1: all planes in one thread) and the oracle: Hodgkin-Huxley + AMPA
(the model of BASELINE configs[2]), three live transmitter types, leaky integrate-and-fire; chunk counts that are not a multiple
of four, fewer chunks than wavefronts (falls back), more than 16 chunks per wavefront; shard handles (the update packs the wire)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def chem_net(model, rows, cols, types, seed, cells=0):
    lay = parity.Layout([(0, rows, cols)], [(1, 1, cells)] if cells else [])
    net = parity.make_oracle(lay, model=model, st_kind=ob.ST_RATE if cells else ob.ST_NONE, chemical=True,
                             nt_kind=ob.NT_DESTEXHE if model == ob.HH else ob.NT_APPROX, rc_kind=ob.RC_DESTEXHE if model == ob.HH else ob.RC_APPROX)
    nn = net.n_neurons
    lo, hi = (-70.0, -60.0) if model == ob.HH else (-65.0, 30.0) if model == ob.IZHIKEVICH else (-75.0, -55.0)
    net["current_voltage"] = ob.uniform_array(seed, nn, lo, hi)
    net["gap_conductance"] = 0.05 if model == ob.HH else 10.0
    for k in types:
        net["nt_flags"][:, k] = 1
        net["rc_flags"][:, k] = 1
    net["nt_flags"][::7, :] = 0                       # some neurons release nothing
    net.fill_graph(seed + 1, 0.5, 1.5)
    rng = np.random.default_rng(seed)
    net["connections"][rng.random(net["connections"].shape) < 0.4] = 0
    net["weights"][...] *= net["connections"]
    if cells:
        net["st_rate"] = 0.5
        net["st_nt_flags"][:, types[0]] = 1
    return net


@pytest.mark.parametrize("model,rows,cols,types,steps", [
    (ob.HH, 36, 36, (0,), 40),                 # 1296 neurons: 6 chunks (2 + 2 + 2 + 0 per wavefront)
    (ob.HH, 64, 66, (0,), 12),                 # 4224 neurons: 17 chunks (5 + 5 + 5 + 2)
    (ob.IZHIKEVICH, 33, 31, (0, 1, 2), 60),    # 1023 neurons + cells: 5 chunks, three live types
    (ob.LIF, 40, 40, (0, 2), 60),
    (ob.IZHIKEVICH, 20, 20, (0,), 60),         # 2 chunks: fewer than the wavefronts of the wide form -- the older kernel steps
    (ob.IZHIKEVICH, 96, 96, (0,), 6),          # 9216 neurons: 36 chunks
])
def test_the_three_forms_of_the_update_agree_with_each_other_and_the_oracle(snn, model, rows, cols, types, steps):
    net = chem_net(model, rows, cols, types, seed=3 + rows, cells=9 if rows == 33 else 0)
    states = []
    for form in (2, 1, 0):
        dn = parity.device_from_oracle(snn, net)
        dn.set_option("update_all_planes", form)
        dn.set_option("persistent_run", 0)            # (small networks would otherwise take the one-launch run: not this kernel)
        dn.set_option("fused_step", 0)
        dn.set_history(voltage=True, spikes=True)
        dn.run(steps)
        assert dn.stat("steps_two_kernel") == steps
        states.append((parity.pull_state(dn, net), parity.bits(dn.voltage_history(0)), dn.spike_history(0)))
        dn.close()
    for st, vh, sh in states[1:]:
        for k in st:
            assert np.array_equal(parity.bits(st[k]), parity.bits(states[0][0][k])), k
        assert np.array_equal(vh, states[0][1]) and np.array_equal(sh, states[0][2])
    net.run(steps, voltage_history=True, spike_history=True)
    parity.assert_state_equal(net, states[0][0])
    assert np.array_equal(states[0][1], parity.bits(net.voltage_history[:, :net.n_neurons]))
    assert np.array_equal(states[0][2], net.spike_history[:, :net.n_neurons])


def test_wide_update_on_shard_handles_packs_the_wire(snn):
    net = chem_net(ob.IZHIKEVICH, 48, 48, (0, 2), seed=9)        # 2304 neurons, 3 shards of 768 (3 chunks each... 9 chunks of rows)
    from snn_amd import parallel
    handles = [parity.device_from_oracle(snn, net, shard=(r, 3)) for r in range(3)]
    for dn in handles:
        dn.set_option("fused_step", 0)
    import torch
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    for _ in range(25):
        for dn in handles:
            dn.step_begin()
        ex.exchange()
        for dn in handles:
            dn.step_end()
    net.run(25)
    for dn in handles:
        st = parity.pull_state(dn, net)
        own = np.asarray(dn.owned)
        for k in ("current_voltage", "is_spiking", "last_firing_time", "nt_t"):
            assert np.array_equal(parity.bits(st[k][own]), parity.bits(net[k][own])), k
        dn.close()

"""Connections BETWEEN lattices of a reward-modulated network (RewardModulatedLatticeNetwork, neuron/mod.rs:3419-3453) that end
in a modulated lattice: the incoming half of update_weights_from_neurons_across_reward_lattices (:4859-4924) -- one modulator
visit per RewardModulatedWeight connection and step (TraceRSTDP::dw and the counter live across steps), the presynaptic
lattice's STDP on the plain Weight connections -- k_reward_cross against the oracle on random networks with a random reward
sequence; connections of kind 0 keep the plain network's rule next to them."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

RM_KEYS = ("rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")


def draw(seed):
    rng = np.random.default_rng(4000 + seed)
    n_lat = int(rng.integers(2, 5))
    lattices = [(2 * i + int(rng.integers(0, 2)), int(rng.integers(1, 6)), int(rng.integers(1, 7))) for i in range(n_lat)]
    st = [(100 + i, int(rng.integers(2, 4)), int(rng.integers(2, 5))) for i in range(int(rng.integers(0, 3)) if seed % 4 == 3 else int(rng.integers(1, 3)))]
    st_kind = ob.ST_POISSON if st else ob.ST_NONE
    net = parity.make_oracle(parity.Layout(lattices, st), st_kind=st_kind, model=[ob.IZHIKEVICH, ob.LIF][seed % 2])
    nn, nc = net.n_neurons, net.n_cells
    lo, hi = ((-65, 30) if seed % 2 == 0 else (-80, -50))
    net["current_voltage"] = ob.uniform_array(seed, nn, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, nn, 8.0, 14.0)
    if seed % 2:
        net["tref"] = ob.uniform_array(seed + 2, nn, 0.2, 2.0)
        net["tau_m"] = 10.0
    if nc:
        net["st_chance_of_firing"] = ob.uniform_array(seed + 4, nc, 0.1, 0.5)
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
    net.fill_graph(seed + 6, 0.5, 3.0, with_diagonal=bool(rng.integers(0, 2)))
    net["connections"][...] &= (rng.random(net["connections"].shape) < float(rng.choice([0.3, 0.8, 1.0])))
    net["weights"][...] *= net["connections"]
    modulated = rng.random(n_lat) < 0.6
    modulated[int(rng.integers(0, n_lat))] = True                       # at least one modulated lattice
    for slot in range(n_lat):
        if modulated[slot]:
            net["rm_do_modulation"][slot] = 1
            net["rm_dopamine"][slot] = float(rng.uniform(0.0, 0.02))
            net["rm_tau_d"][slot] = float(rng.uniform(2.0, 10.0))
            net["rm_tau_c"][slot] = float(rng.uniform(0.02, 0.2))
            net["rm_a_plus"][slot] = float(rng.uniform(0.001, 0.01))
            net["rm_a_minus"][slot] = float(rng.uniform(0.001, 0.01))
        else:
            net["do_plasticity"][slot] = int(rng.integers(0, 2))
            net["stdp_a_plus"][slot] = float(rng.uniform(0.5, 2.5))
            net["stdp_tau_minus"][slot] = float(rng.uniform(2.0, 6.0))
    # connection kinds: into modulated lattices from every other source, at random (0 keeps the plain network's rule)
    for source in range(n_lat + len(st)):
        for post in range(n_lat):
            if modulated[post] and source != post:
                net["conn_kind"][source, post] = int(rng.integers(0, 3))
    net["traces"][...] = ob.uniform_array(seed + 9, net["traces"].size, -0.001, 0.001).reshape(net["traces"].shape) * net["connections"]
    pending = ob.uniform_array(seed + 10, net["pending"].size, -0.01, 0.01).reshape(net["pending"].shape)
    net["pending"][...] = pending * (net["connections"] != 0) * (rng.random(net["pending"].shape) < 0.3)
    net["rm_cross_counter"][...] = rng.integers(0, 2, n_lat)
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    for k in ("dt", "st_dt", "stdp_dt", "rm_dt"):
        net[k] = dt
    steps = int(rng.integers(60, 200))
    rewards = ob.uniform_array(seed + 11, steps, -0.02, 0.03)
    rewards[::3] = 0.0
    return net, steps, rewards


@pytest.mark.parametrize("seed", range(24))
def test_connections_into_modulated_lattices(snn, seed):
    net, steps, rewards = draw(seed)
    dn = parity.device_from_oracle(snn, net)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot]:
            dn.set_reward_modulator(i, *(float(net[k][slot]) for k in RM_KEYS), do_modulation=True)
    dn.set_trace_rows(0, net["traces"])
    parity.push_connection_kinds(dn, net)
    if seed % 3 == 0:
        dn.set_option("defer_rstdp", 0)
    dn.set_history(voltage=True, spikes=True)
    for r in rewards:
        dn.run_with_reward(float(r))
    net.run(steps, voltage_history=True, spike_history=True, rewards=rewards)
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert np.array_equal(parity.bits(dn.get_trace_rows(0, net.n_tot)), parity.bits(net["traces"]))
    assert np.array_equal(parity.bits(dn.get_pending_rows(0, net.n_tot)), parity.bits(net["pending"]))
    assert [dn.connection_counter(i) for i, _, _ in net.layout.lattices] == [int(c) for c in net["rm_cross_counter"]]
    dn.close()


def test_connection_kinds_are_refused_where_they_are_not_built(snn):
    net, _, _ = draw(1)
    dn = parity.device_from_oracle(snn, net, csr=True)
    a, b = net.layout.lattices[0][0], net.layout.lattices[1][0]
    with pytest.raises(snn.SnnError) as e:
        dn.set_connection_kind(a, b, 1)
    assert e.value.code == 12                                  # SNN_ERR_BAD_STATE: dense, unsharded handles
    dn.close()
    dn = parity.device_from_oracle(snn, net)
    with pytest.raises(snn.SnnError):
        dn.set_connection_kind(a, a, 1)                        # a lattice's own edges follow its own rule
    with pytest.raises(snn.SnnError):
        dn.set_connection_kind(a, b, 3)
    dn.close()

"""Connections BETWEEN the lattices of a reward-modulated network (RewardModulatedLatticeNetwork, neuron/mod.rs:3419-3453):
update_weights_from_neurons_across_lattices (:4707-4802) and _across_reward_lattices (:4855-4977), both halves -- the incoming
connections take their rule, the outgoing ones are replaced by the updated copy of their reverse -- k_reward_cross (one thread
per pair of neurons) against the oracle (one visit after the other) on random networks with a random reward sequence;
connections of kind 0 keep the plain network's rule next to them.  Outside the domain the reference defines, the run is refused."""
import numpy as np
import pytest

import parity
import reward_network_cases as cases

pytestmark = pytest.mark.gpu

RM_KEYS = ("rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")


def device_for(snn, net, csr=False):
    dn = parity.device_from_oracle(snn, net, csr=csr)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot] or net["rm_is_modulated"][slot]:
            dn.set_reward_modulator(i, *(float(net[k][slot]) for k in RM_KEYS), do_modulation=bool(net["rm_do_modulation"][slot]))
    if csr:
        dn.set_traces_csr(parity.csr_values(net, net["traces"], dn.owned))
    else:
        dn.set_trace_rows(0, net["traces"])
    parity.push_connection_kinds(dn, net)
    return dn


@pytest.mark.parametrize("csr", [False, True])
@pytest.mark.parametrize("seed", range(16))
def test_a_paused_modulated_lattice_stays_a_modulated_partner(snn, seed, csr):
    """do_modulation off on one reward-modulated lattice (RewardModulatedLattice::do_modulation, neuron/mod.rs:2744): never
    visited, no weight of its own updated, but for its partners' visits still a modulated lattice, and its dopamine follows the
    rewards"""
    net, steps, rewards = cases.draw(seed, paused=True)
    assert net.reward_cross_check() == 0
    dn = device_for(snn, net, csr=csr)
    for r in rewards:
        dn.run_with_reward(float(r))
    net.run(steps, rewards=rewards)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot] or net["rm_is_modulated"][slot]:
            assert np.array_equal(parity.bits(np.array([dn.dopamine(i)])), parity.bits(net["rm_dopamine"][slot:slot + 1]))
    if csr:
        assert np.array_equal(parity.bits(dn.get_traces_csr()), parity.bits(parity.csr_values(net, net["traces"], dn.owned)))
    else:
        assert np.array_equal(parity.bits(dn.get_trace_rows(0, net.n_tot)), parity.bits(net["traces"]))
    dn.close()


@pytest.mark.parametrize("seed", range(32))
def test_connections_between_lattices_on_a_sparse_handle(snn, seed):
    """the same networks with the graph held as CSR rows: k_reward_cross_csr finds the pair of an edge by binary search in the
    row of its presynaptic neuron; trace, dw and counter live per stored edge"""
    net, steps, rewards = cases.draw(seed)
    dn = device_for(snn, net, csr=True)
    if seed % 3 == 0:
        dn.set_option("fused_step", 0)
    if seed % 4 == 1:
        dn.run(steps)
        rewards = None
    else:
        for r in rewards:
            dn.run_with_reward(float(r))
    net.run(steps, rewards=rewards)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    posts = dn.owned
    assert np.array_equal(parity.bits(dn.get_traces_csr()), parity.bits(parity.csr_values(net, net["traces"], posts)))
    if net["conn_kind"].any():
        assert np.array_equal(parity.bits(dn.get_pending_csr()), parity.bits(parity.csr_values(net, net["pending"], posts)))
        assert np.array_equal(dn.get_counters_csr(), parity.csr_values(net, net["edge_counter"], posts))
    dn.close()


@pytest.mark.parametrize("violation", [1, 2, 3])
def test_kinds_outside_the_reference_domain_are_refused_on_a_sparse_handle(snn, violation):
    net, _, _ = cases.draw(3, violation=violation)
    dn = device_for(snn, net, csr=True)
    with pytest.raises(snn.SnnError) as e:
        dn.run(1)
    assert e.value.code == 12 and "neuron/mod.rs" in str(e.value)
    dn.close()


@pytest.mark.parametrize("seed", range(32))
def test_connections_between_lattices(snn, seed):
    net, steps, rewards = cases.draw(seed)
    assert net.reward_cross_check() == 0
    dn = device_for(snn, net)
    if seed % 3 == 0:
        dn.set_option("defer_rstdp", 0)
    dn.set_history(voltage=True, spikes=True)
    if seed % 4 == 1:                       # constant dopamine, one run call
        dn.run(steps)
        rewards = None
    else:
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
    if net["conn_kind"].any():              # (a network whose pairs all drew kind 0 has no such connection: nothing was pushed)
        assert np.array_equal(parity.bits(dn.get_pending_rows(0, net.n_tot)), parity.bits(net["pending"]))
        assert np.array_equal(dn.get_counter_rows(0, net.n_tot), net["edge_counter"])
    dn.close()


@pytest.mark.parametrize("violation", [1, 2, 3])
def test_kinds_outside_the_reference_domain_are_refused(snn, violation):
    """where the reference's visit unwraps None (a missing reverse connection, no modulator for reward-modulated weights, plain
    weights between a plastic and a modulated lattice) the run call says so instead of stepping"""
    net, _, _ = cases.draw(3, violation=violation)
    assert net.reward_cross_check() == violation
    dn = device_for(snn, net)
    with pytest.raises(snn.SnnError) as e:
        dn.run(1)
    assert e.value.code == 12 and "neuron/mod.rs" in str(e.value)          # SNN_ERR_BAD_STATE, the reference line named
    dn.close()


def test_a_repaired_network_runs(snn):
    net, _, _ = cases.draw(3, violation=1)
    dn = device_for(snn, net)
    with pytest.raises(snn.SnnError):
        dn.run(1)
    a, b = net.layout.lattices[0][0], net.layout.lattices[1][0]
    dn.set_connection_kind(a, b, 0)
    dn.set_connection_kind(b, a, 0)
    net["conn_kind"][0, 1] = net["conn_kind"][1, 0] = 0
    assert net.reward_cross_check() == 0
    dn.run(20)
    net.run(20)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    dn.close()


def test_connection_kinds_are_refused_where_they_are_not_built(snn):
    net, _, _ = cases.draw(1)
    dn = parity.device_from_oracle(snn, net, shard=(0, 2), csr=True)
    a, b = net.layout.lattices[0][0], net.layout.lattices[1][0]
    with pytest.raises(snn.SnnError) as e:
        dn.set_connection_kind(a, b, 1)
    assert e.value.code == 12                                  # SNN_ERR_BAD_STATE: unsharded handles
    dn.close()
    dn = parity.device_from_oracle(snn, net)
    with pytest.raises(snn.SnnError):
        dn.set_connection_kind(a, a, 1)                        # a lattice's own edges follow its own rule
    with pytest.raises(snn.SnnError):
        dn.set_connection_kind(a, b, 3)
    dn.close()

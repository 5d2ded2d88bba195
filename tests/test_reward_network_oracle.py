"""The connections between the lattices of a reward-modulated network on the CPU: the oracle walks the reference's visits one
neuron after the other (snn_o_reward_cross), the numpy restatement replays them pair by pair (NumpyNet.reward_cross) -- two
independently written orders of the same updates have to agree bit for bit; and the refusal classes are recognised."""
import numpy as np
import pytest

import numpy_net
import parity
import reward_network_cases as cases


@pytest.mark.parametrize("paused", [False, True])
@pytest.mark.parametrize("seed", range(12))
def test_sequential_visits_equal_the_pair_form(seed, paused):
    net, steps, rewards = cases.draw(seed, paused=paused)
    assert net.reward_cross_check() == 0
    steps = min(steps, 80)
    twin = numpy_net.NumpyNet(net).run(steps, rewards=rewards[:steps])
    net.run(steps, voltage_history=True, spike_history=True, rewards=rewards[:steps])
    assert np.array_equal(twin.spike_history, net.spike_history)
    for k in ("weights", "traces", "pending", "edge_counter", "current_voltage", "last_firing_time"):
        assert np.array_equal(parity.bits(twin.a[k]) if twin.a[k].dtype == np.float32 else twin.a[k],
                              parity.bits(net[k]) if net[k].dtype == np.float32 else net[k]), k


def test_the_cases_exercise_both_halves():
    """over the seeds: reward-modulated and plain kinds, spiking plastic neurons (plain visits) and weights that moved"""
    kinds, moved = set(), 0
    for seed in range(12):
        net, steps, rewards = cases.draw(seed)
        kinds |= set(np.unique(net["conn_kind"]).tolist())
        before = net["weights"].copy()
        net.run(40, rewards=rewards[:40])
        lat = net["lattice"]
        cross = lat[:, None] != lat[None, :]
        moved += int((parity.bits(before[:net.n_neurons]) != parity.bits(net["weights"][:net.n_neurons]))[cross].sum())
    assert kinds == {0, 1, 2} and moved > 100


@pytest.mark.parametrize("violation", [1, 2, 3])
def test_refusal_classes(violation):
    net, _, _ = cases.draw(3, violation=violation)
    assert net.reward_cross_check() == violation


def test_a_paused_modulated_lattice_is_still_a_modulated_partner():
    """neuron/mod.rs:4869 / :4937: the visit of a neuron of an ACTIVE modulated lattice leaves a plain-weight connection with a
    partner in another modulated lattice alone -- whether or not that partner's do_modulation is set; a plain lattice in its
    place would lend its STDP rule.  And the paused lattice's dopamine still follows the rewards (:5287-5291)."""
    for seed in range(40):
        net, steps, rewards = cases.draw(seed, paused=True)
        lat, nn = net["lattice"], net.n_neurons
        paused = int(np.flatnonzero(net["rm_is_modulated"])[0])
        active = np.flatnonzero(net["rm_do_modulation"])
        pairs = [a for a in active if net["conn_kind"][paused, a] == 2]
        if not pairs:
            continue
        a = int(pairs[0])
        rows, cols = np.flatnonzero(lat == paused), np.flatnonzero(lat == a)
        block = np.ix_(rows, cols)
        if not net["connections"][:nn][block].any():
            continue
        before, dop = net["weights"][:nn][block].copy(), float(net["rm_dopamine"][paused])
        net.run(30, rewards=rewards[:30])
        assert np.array_equal(parity.bits(before), parity.bits(net["weights"][:nn][block])), "kind-2 weights between two modulated lattices never move"
        assert float(net["rm_dopamine"][paused]) != dop, "a paused modulator still takes the rewards"
        return
    raise AssertionError("no case with a kind-2 connection between a paused and an active modulated lattice")

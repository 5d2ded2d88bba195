"""The connections between the lattices of a reward-modulated network on the CPU: the oracle walks the reference's visits one
neuron after the other (snn_o_reward_cross), the numpy restatement replays them pair by pair (NumpyNet.reward_cross) -- two
independently written orders of the same updates have to agree bit for bit; and the refusal classes are recognised."""
import numpy as np
import pytest

import numpy_net
import parity
import reward_network_cases as cases


@pytest.mark.parametrize("seed", range(12))
def test_sequential_visits_equal_the_pair_form(seed):
    net, steps, rewards = cases.draw(seed)
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

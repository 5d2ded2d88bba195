"""STDP INSIDE the one-launch run (k_run_resident<..., STDP>): electrical lattices of neurons, <= 1024 of them, whose weights
live in the workgroups' registers / LDS for the run and take the deferred STDP update of every step there (the last step's by
the plain kernels) -- against the oracle and against one launch per step: weights, state, rasters, traces, bit for bit; several
lattices with their own rules and plasticity switches, absent edges, split run calls, a faulted launch rolled back."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(model, lattices, seed, density=0.8, drive=(-70, 29.9)):
    net = parity.make_oracle(parity.Layout(lattices), model=model, electrical=True, chemical=False)
    n = net.n_neurons
    rng = np.random.default_rng(seed)
    net["current_voltage"] = ob.uniform_array(seed, n, *drive)
    net["gap_conductance"] = ob.uniform_array(seed + 5, n, 0.2, 1.0)      # weak coupling: the neurons fire one after the other
    if model in (ob.LIF, ob.QIF):
        net["tref"] = ob.uniform_array(seed + 1, n, 0.3, 1.5)
        net["tau_m"] = 10.0
    net.fill_graph(seed + 3, 0.5, 2.5, with_diagonal=bool(seed % 2))
    net["connections"][rng.random(net["connections"].shape) >= density] = 0
    net["weights"][...] *= net["connections"]
    for l in range(len(lattices)):
        net["do_plasticity"][l] = int(rng.random() < 0.75)
        net["stdp_a_plus"][l] = float(rng.uniform(0.5, 2.5))
        net["stdp_a_minus"][l] = float(rng.uniform(0.5, 2.5))
        net["stdp_tau_plus"][l] = float(rng.uniform(2.0, 6.0))
        net["stdp_tau_minus"][l] = float(rng.uniform(2.0, 6.0))
    net["do_plasticity"][int(rng.integers(0, len(lattices)))] = 1
    # some neurons have fired before the run starts
    fired = rng.random(n) < 0.3
    net["last_firing_time"][...] = np.where(fired, 0, -1).astype(np.int32)
    return net


def run_device(snn, net, calls, persistent, fault=0, chunk=0):
    dn = parity.device_from_oracle(snn, net)
    dn.set_option("persistent_run", int(persistent))
    if chunk:
        dn.set_option("run_resident_chunk_steps", chunk)
    if fault:
        dn.set_option("run_resident_fault_step", fault)
        dn.set_option("run_resident_spin_limit", 20000)
    dn.set_history(voltage=True, spikes=True)
    for steps in calls:
        dn.run(steps)
    w, _ = dn.get_graph_rows(0, net.n_tot)
    out = {"state": parity.pull_state(dn, net), "w": w, "stdp_steps": dn.stat("persistent_run_stdp_steps"),
           "launches": dn.stat("persistent_run_launches"), "fallbacks": dn.stat("persistent_run_fallbacks"),
           "v": [dn.voltage_history(i) for i, _, _ in net.layout.lattices], "s": [dn.spike_history(i) for i, _, _ in net.layout.lattices]}
    dn.close()
    return out


def check(net, out):
    parity.assert_state_equal(net, out["state"])
    want = np.where(net["connections"] != 0, net["weights"], np.float32(0))
    assert np.array_equal(parity.bits(out["w"]), parity.bits(want))
    rng = net.layout.ranges()
    for k, (i, _, _) in enumerate(net.layout.lattices):
        first, count, _ = rng[i]
        assert np.array_equal(out["s"][k], net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(out["v"][k]), parity.bits(net.voltage_history[:, first:first + count]))


CASES = [(ob.IZHIKEVICH, [(3, 4, 4)], 1), (ob.IZHIKEVICH, [(3, 8, 8)], 2), (ob.IZHIKEVICH, [(3, 16, 16)], 3), (ob.IZHIKEVICH, [(3, 32, 32)], 4),
         (ob.IZHIKEVICH, [(0, 9, 9), (1, 7, 11)], 5), (ob.IZHIKEVICH, [(0, 10, 10), (2, 5, 5), (5, 12, 9), (7, 3, 21)], 6),
         (ob.LIF, [(3, 12, 13)], 7), (ob.LIF, [(1, 6, 6), (4, 15, 15)], 8), (ob.QIF, [(3, 10, 10)], 9), (ob.HH, [(3, 8, 9)], 10),
         (ob.IZHIKEVICH, [(3, 31, 33)], 11), (ob.IZHIKEVICH, [(3, 1, 3)], 12)]


@pytest.mark.parametrize("model,lattices,seed", CASES)
def test_stdp_inside_the_one_launch_run(snn, model, lattices, seed):
    calls = [120, 3, 1, 60, 4]
    net = build(model, lattices, seed, drive=(-75, -40) if model == ob.HH else (-70, 29.9))
    before = net["weights"].copy()
    a = run_device(snn, net, calls, True)
    b = run_device(snn, net, calls, False)
    net.run(sum(calls), voltage_history=True, spike_history=True)
    assert a["stdp_steps"] == 120 + 60 + 4 and a["launches"] == 3 and b["launches"] == 0
    check(net, a)
    check(net, b)
    if net.spike_history[1:].sum() > 2 and net["do_plasticity"].all():     # spikes after the first step (the preset firing times are 0)
        assert not np.array_equal(before, net["weights"])
    if model == ob.IZHIKEVICH and net.n_neurons >= 64:
        assert net.spike_history[1:].sum() > 10                            # (weakly coupled: they keep firing)


def test_a_faulted_run_leaves_the_weights_to_the_per_step_repeat(snn):
    net = build(ob.IZHIKEVICH, [(0, 12, 12), (1, 10, 10)], 21)
    a = run_device(snn, net, [90, 40], True, fault=37)
    net.run(130, voltage_history=True, spike_history=True)
    assert a["fallbacks"] == 1 and a["stdp_steps"] == 0
    check(net, a)


@pytest.mark.parametrize("fault", [5, 21, 40, 63])
def test_a_fault_in_a_later_chunk_rolls_back_to_that_chunk_only(snn, fault):
    """A run call longer than one launch's chunk (2^20 steps; 16 here through the test hook) commits the weights of every
    completed chunk.  A launch that gives up later in the call must be rolled back to ITS start: replaying the whole call per step
    would apply the committed chunks' STDP updates a second time (round-4 review of snn_network_step.hpp:890)."""
    net = build(ob.IZHIKEVICH, [(0, 12, 12), (1, 10, 10)], 23)
    a = run_device(snn, net, [64, 30], True, fault=fault, chunk=16)
    net.run(94, voltage_history=True, spike_history=True)
    assert a["fallbacks"] == 1
    assert a["stdp_steps"] == 16 * ((fault - 1) // 16), "the chunks before the faulted one stay committed"
    check(net, a)


def test_chunked_runs_equal_one_launch(snn):
    net = build(ob.IZHIKEVICH, [(0, 9, 9), (1, 7, 11)], 24)
    a = run_device(snn, net, [100, 7], True, chunk=16)
    net.run(107, voltage_history=True, spike_history=True)
    assert a["launches"] == 7 + 1 and a["fallbacks"] == 0          # 6 chunks of 16 + one of 4, then 7 steps in one launch
    check(net, a)


def test_networks_outside_its_reach_keep_one_launch_per_step(snn):
    """five lattices (the kernel's table holds four rules), a BCM lattice, or the option switched off"""
    net = build(ob.IZHIKEVICH, [(i, 4, 4) for i in range(5)], 31)
    a = run_device(snn, net, [50], True)
    net.run(50, voltage_history=True, spike_history=True)
    assert a["launches"] == 0
    check(net, a)


@pytest.mark.parametrize("seed", [2008404, 2002804, 2008236])
def test_voltages_that_leave_the_plain_range_use_the_resident_weights(snn, seed):
    """found by the contention campaign: networks whose voltages explode (beyond 1e15 the chains skip absent edges explicitly) read
    their weights from the matrix in memory, which is stale while the run updates them in registers"""
    import test_gpu_persistent_run as tp
    tp.test_random_electrical_networks(snn, seed)

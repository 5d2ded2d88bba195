"""The tools that localise a device / oracle mismatch (tests/checkpoint.py, round 5), tested on deliberate faults:
handle checkpoints (snn_debug_checkpoint), handles that continue at a running network's clocks (snn_set_clock), the stepper's
self-check (option "verify"), and the localiser itself -- a fault planted on the oracle's side, and one planted on the device's
side, must each be named as such."""
import json
import os

import numpy as np
import pytest

import checkpoint
import parity
from test_gpu_randomized import draw, make_handle

pytestmark = pytest.mark.gpu


def plain(seed):
    net, plan = draw(1000 + seed)
    return plan["shards"] == 1 and plan["rewards"] is None and net.n_neurons > 0 and net.n_tot > 0


SEEDS = [s for s in range(60) if plain(s)][:24]


@pytest.mark.parametrize("seed", SEEDS)
def test_a_checkpoint_brings_back_what_a_run_call_reads(snn, seed):
    net, plan = draw(1000 + seed)
    dn = make_handle(snn, net, plan)
    dn.run(17)
    net.run(17)
    dn.checkpoint()
    dn.run(31)
    first = checkpoint.pull_all(dn, net)
    clock = dn.clock
    for _ in range(2):
        dn.restore_checkpoint()
        assert dn.clock == 17
        dn.run(31)
        again = checkpoint.pull_all(dn, net)
        assert dn.clock == clock and checkpoint.same_state(first[0], again[0])
        if first[1] is not None:
            assert np.array_equal(parity.bits(first[1][0]), parity.bits(again[1][0]))
    net.run(31)
    assert not checkpoint.state_diffs(net, *checkpoint.pull_all(dn, net))
    dn.close()


@pytest.mark.parametrize("seed", SEEDS[:12])
def test_a_handle_built_from_a_running_network_continues_at_its_clocks(snn, seed):
    """LatticeNetworkGPU::from_network hands internal_clock over (neuron/gpu_lattices/mod.rs:1630): firing times are absolute"""
    net, plan = draw(1000 + seed)
    net.run(43)
    dn = checkpoint.fresh_handle(snn, checkpoint.clone(net), plan, {})
    assert dn.clock == 43
    for i, _, _ in net.layout.st_lattices:
        assert dn.spike_train_clock(i) == int(net["st_clock"][[j for j, _, _ in net.layout.st_lattices].index(i)])
    dn.run(29)
    net.run(29)
    assert not checkpoint.state_diffs(net, *checkpoint.pull_all(dn, net))
    dn.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_the_self_check_steps_every_call_twice_and_agrees(snn, seed):
    net, plan = draw(1000 + seed)
    dn = make_handle(snn, net, plan)
    dn.set_option("verify", 1)
    dn.set_history(voltage=True, spikes=True)
    for k in (23, 1, 40):
        dn.run(k)
    net.run(64, voltage_history=True, spike_history=True)
    assert dn.stat("verify_mismatches") == 0, dn.verify_report()
    assert dn.stat("verify_runs") + dn.stat("verify_skipped") == 3 and dn.stat("verify_runs") >= 2
    assert dn.clock == 64 and dn.history_steps() == 64
    assert not checkpoint.state_diffs(net, *checkpoint.pull_all(dn, net))
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    dn.close()


def test_the_self_check_reports_a_planted_difference(snn, capfd):
    net, plan = draw(1000 + SEEDS[0])
    net["do_plasticity"][...] = 0
    dn = make_handle(snn, net, plan)
    dn.set_option("verify", 1)
    dn.run(9)                                                  # (the first run may lay the snapshot out anew: skipped or compared)
    dn.set_option("verify_fault", 1)                           # the voltage of neuron 0, after the second pass
    dn.run(12)
    assert dn.stat("verify_mismatches") == 1
    text = dn.verify_report()
    assert "1 words differ" in text and "exchange buffer, plane 0, neuron 0" in text and "run of 12 steps" in text, text
    assert "[snn verify] MISMATCH" in capfd.readouterr().err
    dn.run(5)
    assert dn.stat("verify_mismatches") == 1                    # (the hook fires once)
    dn.close()


@pytest.mark.parametrize("csr", [False, True])
@pytest.mark.parametrize("seed", range(8))
def test_the_self_check_covers_runs_with_weight_updates(snn, seed, csr):
    """reward-modulated networks with connections between their lattices: weights, traces, dw and counters are part of what is
    put back before the second pass and of what is compared after it"""
    import reward_network_cases as cases
    from test_gpu_reward_network import device_for
    net, steps, rewards = cases.draw(seed)
    dn = device_for(snn, net, csr=csr)
    dn.set_option("verify", 1)
    if seed % 4 == 1:
        dn.run(steps)
        rewards = None
    else:
        for r in rewards:
            dn.run_with_reward(float(r))
    net.run(steps, rewards=rewards)
    assert dn.stat("verify_mismatches") == 0, dn.verify_report()
    assert dn.stat("verify_runs") >= (1 if rewards is None else steps - 1)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    if csr:
        assert np.array_equal(parity.bits(dn.get_traces_csr()), parity.bits(parity.csr_values(net, net["traces"], dn.owned)))
    else:
        assert np.array_equal(parity.bits(dn.get_trace_rows(0, net.n_tot)), parity.bits(net["traces"]))
    dn.close()


def test_the_self_check_reports_a_planted_difference_in_the_weights(snn):
    net, plan = draw(1000 + next(s for s in SEEDS if draw(1000 + s)[0]["do_plasticity"].any()))
    dn = make_handle(snn, net, plan)
    dn.set_option("verify", 1)
    dn.run(9)
    dn.set_option("verify_fault", (1 << 30) + 1)
    dn.run(12)
    assert dn.stat("verify_mismatches") == 1
    text = dn.verify_report()
    assert "1 words differ" in text and ("W, word 1" in text or "sparse weights, word 1" in text), text
    dn.close()


class FaultyHandle:
    """a device handle whose `fault_on`-th run call ends with one voltage nudged -- a transient of the device's side"""

    def __init__(self, dn, lattice, fault_on):
        self._dn, self._lattice, self._fault_on, self._runs = dn, lattice, fault_on, 0

    def __getattr__(self, name):
        return getattr(self._dn, name)

    def run(self, k):
        self._dn.run(k)
        self._runs += 1
        if self._runs == self._fault_on:
            v = self._dn.get_attr(self._lattice, "current_voltage")
            v[0] = np.nextafter(v[0], np.float32(1e9))
            self._dn.set_attr(self._lattice, "current_voltage", v)


def first_populated(net):
    return next(i for i, r, c in net.layout.lattices if r * c)


@pytest.mark.parametrize("side", ["oracle", "device"])
def test_the_localiser_names_the_side_that_was_wrong(snn, side, monkeypatch, tmp_path):
    monkeypatch.setattr(checkpoint, "REPEATS", 8)
    monkeypatch.setenv("SNN_REPRO_DIR", str(tmp_path))
    seed = SEEDS[3]
    net, plan = draw(1000 + seed)
    dn = make_handle(snn, net, plan)
    real_run, runs = net.run, [0]

    def faulty_oracle(k, **kw):
        real_run(k, **kw)
        runs[0] += 1
        if runs[0] == 2:                                       # a transient of the checker's side
            net["current_voltage"][0] = np.nextafter(net["current_voltage"][0], np.float32(1e9))

    if side == "oracle":
        net.run = faulty_oracle
        handle = dn
    else:
        handle = FaultyHandle(dn, first_populated(net), 2)
    tr = checkpoint.Tracker(snn, handle, net, plan, f"planted-{side}")
    tr.run(20)
    with pytest.raises(AssertionError) as e:
        tr.run(33)
    text = str(e.value)
    assert "run call 2 (33 steps)" in text and "this run call diverged" in text
    if side == "oracle":
        assert "ORACLE does not reproduce its own result" in text
    else:
        assert "transient of the device side" in text
    bundle = [f for f in os.listdir(tmp_path) if f.endswith(".json") and f.startswith(f"planted-{side}")]
    assert len(bundle) == 1
    meta = json.load(open(tmp_path / bundle[0]))
    loc = meta["localisation"]
    assert loc["same_handle"]["runs"] == 8 and loc["same_handle"]["equal_to_oracle_replay"] == 8
    assert loc["fresh_handle"]["equal_to_oracle_replay"] == loc["fresh_handle"]["runs"]
    assert loc["child_process"].get("equal_to_oracle_replay") == loc["child_process"].get("runs") == 2
    assert loc["first_differing_step"] is None                  # (executed again, nothing differs: the fault was a transient)
    assert "sysfs" in meta["ras_after"]
    dn.close()


def test_a_difference_that_predates_the_call_is_blamed_on_the_calls_before_it(snn, monkeypatch, tmp_path):
    monkeypatch.setenv("SNN_REPRO_DIR", str(tmp_path))
    seed = SEEDS[5]
    net, plan = draw(1000 + seed)
    dn = make_handle(snn, net, plan)
    tr = checkpoint.Tracker(snn, dn, net, plan, "planted-setter")
    tr.run(11)
    i = first_populated(net)
    v = dn.get_attr(i, "gap_conductance")
    v[0] += 1.0
    dn.set_attr(i, "gap_conductance", v)                        # (the oracle is not told)
    with pytest.raises(AssertionError) as e:
        tr.run(5)
    assert "differed BEFORE this run call" in str(e.value) and "gap_conductance" in str(e.value)
    dn.close()

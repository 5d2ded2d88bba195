"""GPU parity for reward modulation: RewardModulatedLattice (neuron/mod.rs:2719-3417) with
RewardModulatedSTDP + TraceRSTDP (plasticity/mod.rs:126-242) -- every internal edge of a modulated lattice is
updated every step (dopamine-gated trace), in the deferred form.  Weights, traces, dopamine and all neuron state
bit-identical to the oracle; dense, sparse and sharded handles; the update as a standalone pass and fused into the
next step's input pass."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(lattices=((0, 7, 8),), modulated=(0,), st=(), seed=1, density=0.7):
    lay = parity.Layout(list(lattices), list(st))
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE if st else ob.ST_NONE)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    if st:
        net["st_rate"] = ob.uniform_array(seed + 3, net.n_cells, 1.0, 5.0)
    net.fill_graph(seed + 1, 0.5, 1.5)
    rng = np.random.default_rng(seed)
    net["connections"][rng.random(net["connections"].shape) > density] = 0
    net["weights"][...] *= net["connections"]
    ids = [i for i, _, _ in lay.lattices]
    for slot, i in enumerate(ids):
        if i in modulated:
            net["rm_do_modulation"][slot] = 1
            net["rm_tau_c"][slot] = 0.05 + 0.01 * slot
            net["rm_tau_d"][slot] = 5.0 + slot
            net["rm_a_plus"][slot] = 0.002
            net["rm_a_minus"][slot] = 0.0015
        else:
            net["do_plasticity"][slot] = 1
    net["traces"][...] = ob.uniform_array(seed + 5, net["traces"].size, -0.001, 0.001).reshape(net["traces"].shape)
    net["traces"][...] *= net["connections"]
    return net


def to_device(snn, net, **kw):
    dn = parity.device_from_oracle(snn, net, **kw)
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot]:
            dn.set_reward_modulator(i, *(float(net[k][slot]) for k in (
                "rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")),
                do_modulation=True)
    return dn


def rewards_for(steps, seed):
    r = ob.uniform_array(seed, steps, -0.02, 0.03)
    r[::3] = 0.0
    return r


def check_dense(dn, net, b=0, e=None):
    e = net.n_neurons if e is None else e
    w, c = dn.get_graph_rows(0, net.n_tot)
    oc = net["connections"].astype(np.uint32)
    ow = np.where(oc != 0, net["weights"], np.float32(0))
    assert np.array_equal(c[:, b:e], oc[:, b:e])
    assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e])), "weights"
    t = dn.get_trace_rows(0, net.n_tot)
    assert np.array_equal(parity.bits(t[:, b:e]), parity.bits(net["traces"][:, b:e])), "traces"


@pytest.mark.parametrize("case", ["single", "two_lattices_one_modulated", "both_modulated_with_spike_trains"])
def test_reward_modulated_lattice_equals_oracle(snn, case):
    if case == "single":
        net = build()
    elif case == "two_lattices_one_modulated":
        net = build(lattices=((0, 5, 6), (2, 6, 7)), modulated=(2,), seed=3)     # lattice 0 keeps plain STDP
    else:
        net = build(lattices=((0, 5, 6), (2, 6, 7)), modulated=(0, 2), st=((5, 2, 3),), seed=5)
    steps = 400
    rewards = rewards_for(steps, 9)
    dn = to_device(snn, net)
    dn.set_trace_rows(0, net["traces"])
    dn.set_history(voltage=True, spikes=True)
    for r in rewards:
        dn.run_with_reward(float(r))
    w0 = net["weights"].copy()
    net.run(steps, voltage_history=True, spike_history=True, rewards=rewards)
    assert net.spike_history.sum() > 10 and not np.array_equal(w0, net["weights"])
    rng = net.layout.ranges()
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
        if net["rm_do_modulation"][slot]:
            assert dn.dopamine(i) == net["rm_dopamine"][slot] and net["rm_dopamine"][slot] != 0
    check_dense(dn, net)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


def test_run_without_reward_keeps_dopamine(snn):
    """RunLattice::run_lattice on a reward-modulated lattice (neuron/mod.rs:3361-3375) iterates without touching the
    modulator: dopamine stays, traces and weights still move; a mixed sequence of both calls matches the oracle."""
    net = build(seed=7)
    net["rm_dopamine"][0] = 0.01
    dn = to_device(snn, net)
    dn.set_trace_rows(0, net["traces"])
    dn.run(50)
    net.run(50)
    assert dn.dopamine(0) == np.float32(0.01)
    dn.run_with_reward(0.5)
    dn.run(30)
    net.run(1, rewards=[0.5])
    net.run(30)
    assert dn.dopamine(0) == net["rm_dopamine"][0]
    check_dense(dn, net)
    # do_modulation = false freezes weights and traces (neuron/mod.rs:3142-3144)
    dn.set_reward_modulator(0, *(float(net[k][0]) for k in ("rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus",
                                                            "rm_tau_plus", "rm_tau_minus", "rm_dt")), do_modulation=False)
    w, _ = dn.get_graph_rows(0, net.n_tot)
    dn.run_with_reward(0.3)
    dn.run(20)
    w2, _ = dn.get_graph_rows(0, net.n_tot)
    assert np.array_equal(parity.bits(w), parity.bits(w2))
    dn.close()


def test_sparse_handle_matches_dense(snn):
    net = build(lattices=((0, 6, 6), (1, 5, 7)), modulated=(0, 1), st=((4, 2, 2),), seed=11, density=0.3)
    steps = 300
    rewards = rewards_for(steps, 12)
    dn = to_device(snn, net, csr=True)
    ptr, pre, w = parity.csr_from_dense(net, 0, net.n_neurons)
    traces_csr = np.concatenate([net["traces"][pre[ptr[q]:ptr[q + 1]], q] for q in range(net.n_neurons)])
    dn.set_traces_csr(traces_csr)
    for r in rewards:
        dn.run_with_reward(float(r))
    net.run(steps, rewards=rewards)
    expect_w = np.concatenate([net["weights"][pre[ptr[q]:ptr[q + 1]], q] for q in range(net.n_neurons)])
    expect_c = np.concatenate([net["traces"][pre[ptr[q]:ptr[q + 1]], q] for q in range(net.n_neurons)])
    assert np.array_equal(parity.bits(dn.get_graph_csr()), parity.bits(expect_w))
    assert np.array_equal(parity.bits(dn.get_traces_csr()), parity.bits(expect_c))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_sharded_handles(snn, n_shards):
    """Each shard owns its postsynaptic columns of W and of the trace matrix; dopamine is replicated."""
    import torch
    from snn_amd import parallel
    net = build(lattices=((0, 8, 8), (3, 9, 10)), modulated=(0, 3), st=((5, 2, 3),), seed=13)
    steps = 250
    rewards = rewards_for(steps, 14)
    handles = [to_device(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    for h in handles:
        h.set_trace_rows(0, net["traces"])
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    for rwd in rewards:
        for h in handles:
            h.apply_reward(float(rwd))
            h.step_begin_local()          # refused internally while modulation is on
            h.step_begin()
        ex.exchange()
        for h in handles:
            h.step_end()
    net.run(steps, rewards=rewards)
    for h in handles:
        check_dense(h, net, h.post_begin, h.post_end)
        assert h.dopamine(3) == net["rm_dopamine"][1]
        h.close()


def test_reward_errors(snn):
    net = build(st=((4, 1, 2),))
    dn = parity.device_from_oracle(snn, net)
    with pytest.raises(snn.SnnError):
        dn.set_reward_modulator(4)                 # a spike-train lattice
    with pytest.raises(snn.SnnError):
        dn.set_reward_modulator(99)
    with pytest.raises(snn.SnnError):
        dn.get_traces_csr()                        # dense handle
    with pytest.raises(snn.SnnError):
        dn.get_trace_rows(0, net.n_tot + 1)
    dn.apply_reward(1.0)                           # nothing modulated: a no-op
    dn.run(2)
    dn.close()


def test_sharded_stepper_applies_rewards_in_step_order(snn):
    """parallel.ShardedStepper.run(k, rewards=...) on a shard handle (world size 1, stream-ordered overlap loop):
    each reward must land after the previous step's weight update and before its own step."""
    import torch
    from snn_amd import parallel
    net = build(lattices=((0, 8, 8), (3, 6, 7)), modulated=(0,), st=((5, 2, 3),), seed=17)
    steps = 200
    rewards = rewards_for(steps, 18)
    dn = to_device(snn, net, shard=(0, 1))
    dn.set_trace_rows(0, net["traces"])
    side = torch.cuda.Stream()
    dn.set_stream(side.cuda_stream)
    stepper = parallel.ShardedStepper(dn, 0, 1, stream=side, device=torch.device("cuda", 0))
    stepper.run(steps // 2, rewards=rewards[:steps // 2])
    stepper.run(steps - steps // 2, rewards=rewards[steps // 2:])
    dn.synchronize()
    net.run(steps, rewards=rewards)
    check_dense(dn, net)
    assert dn.dopamine(0) == net["rm_dopamine"][0]
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


def _env(name, value):
    import contextlib
    import os

    @contextlib.contextmanager
    def cm():
        old = os.environ.get(name)
        os.environ[name] = value
        try:
            yield
        finally:
            if old is None:
                del os.environ[name]
            else:
                os.environ[name] = old
    return cm()


@pytest.mark.parametrize("case", ["two_kernel_small", "five_chunks", "streaming_shape"])
def test_weight_update_deferred_into_the_next_input_pass(snn, case):
    """Dense handles on the two-kernel path apply the update of step t inside the input pass of step t+1
    (k_inputs_rstdp).  Same results as the oracle, with host reads in between (which flush the pending update), two
    rewards before one step, and a reward arriving between the deferral and the pass."""
    if case == "two_kernel_small":
        net = build(lattices=((0, 9, 9), (2, 7, 8)), modulated=(0, 2), st=((5, 2, 3),), seed=21)
        steps, env = 240, ("SNN_AMD_FUSED_STEP", "0")
    elif case == "five_chunks":
        net = build(lattices=((0, 34, 34),), seed=23)            # 1156 rows: beyond the one-launch step
        steps, env = 90, ("SNN_AMD_FUSED_STEP", "1")
    else:
        net = build(lattices=((0, 66, 64),), seed=25, density=0.9)   # 71 MB matrix: streamed (2 columns per lane)
        steps, env = 12, ("SNN_AMD_FUSED_STEP", "1")
    rewards = rewards_for(steps, 26)
    with _env(*env):
        dn = to_device(snn, net)
    dn.set_trace_rows(0, net["traces"])
    third = steps // 3
    for r in rewards[:third]:
        dn.run_with_reward(float(r))
    net.run(third, rewards=rewards[:third])
    check_dense(dn, net)                                     # host read: flushes the pending update
    # two rewards before one step (the second must not disturb the dopamine of the pending update)
    dn.apply_reward(0.02)
    dn.apply_reward(-0.01)
    dn.run(1)
    net.apply_reward(0.02)
    net.apply_reward(-0.01)
    net.run(1)
    for r in rewards[third:]:
        dn.run_with_reward(float(r))
    net.run(steps - third, rewards=rewards[third:])
    check_dense(dn, net)
    assert dn.dopamine(0) == net["rm_dopamine"][0]
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()


def test_deferred_and_standalone_update_agree_at_streaming_size(snn):
    """160x160 neurons (2.6 GB matrix + 2.6 GB of traces: the 4-columns-per-lane streaming shape, too large for the
    oracle): the deferred update (k_inputs_rstdp) against the standalone pass (SNN_AMD_DEFER_RSTDP=0), device against
    device, sampled rows bit-identical."""
    n = 160 * 160
    out = []
    for defer in ("1", "0"):
        with _env("SNN_AMD_DEFER_RSTDP", defer):
            dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
        dn.add_lattice(0, 160, 160)
        dn.finalize()
        dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", ob.uniform_array(31, n, -65.0, 30.0))
        dn.fill_graph_synthetic(32, 0.5, 1.5)
        # firing times on record from the start, so that every pair of neurons has a non-zero STDP delta
        dn.set_attr(0, "last_firing_time", np.random.default_rng(34).integers(0, 60, n).astype(np.int32))
        dn.set_reward_modulator(0, tau_c=0.05, tau_d=5.0, a_plus=0.002, a_minus=0.0015)
        for r in rewards_for(40, 33):
            dn.run_with_reward(float(r))
        rows = [0, 1, 255, 256, 12345, n - 1]
        w = np.stack([dn.get_graph_rows(p, 1)[0][0] for p in rows])
        t = np.stack([dn.get_trace_rows(p, 1)[0] for p in rows])
        v = dn.get_attr(0, "current_voltage")
        lft = dn.get_attr(0, "last_firing_time", dtype=np.int32)
        out.append((w, t, v, lft, dn.dopamine(0)))
        dn.close()
    a, b = out
    assert np.abs(a[1]).max() > 0 and a[4] != 0, "traces and dopamine must have moved"
    for x, y in zip(a, b):
        assert np.array_equal(parity.bits(np.asarray(x)), parity.bits(np.asarray(y)))

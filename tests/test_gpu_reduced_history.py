"""GPU parity for the reduced histories (snn_set_reduced_history): per-lattice AverageVoltageHistory
(neuron/mod.rs:305-322) and EEGHistory (:233-284) values per step, and the per-neuron spike totals of
SpikeHistory::aggregate (:331-360).  Bit-exact against the oracle, which sums in the same canonical 256-chunk
order; the oracle's values are also compared with a float64 numpy mean within float32 rounding."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(lattices, st=(), seed=1):
    lay = parity.Layout(lattices, st)
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE if st else ob.ST_NONE)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    if st:
        net["st_rate"] = ob.uniform_array(seed + 3, net.n_cells, 1.0, 5.0)
    net.fill_graph(seed + 1, 0.5, 1.5)
    rng = np.random.default_rng(seed)
    net["connections"][rng.random(net["connections"].shape) < 0.4] = 0
    return net


def check(dn, net, steps):
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        assert np.array_equal(parity.bits(dn.average_voltage_history(i)), parity.bits(net.avg_history[:steps, slot])), i
        assert np.array_equal(parity.bits(dn.eeg_history(i)), parity.bits(net.eeg_history[:steps, slot])), i
        first, count, _ = net.layout.ranges()[i]
        assert np.array_equal(dn.spike_counts(i), net.spike_counts[first:first + count]), i


@pytest.mark.parametrize("lattices", [[(0, 5, 7)], [(0, 9, 10), (3, 20, 20), (7, 1, 1)], [(2, 40, 40)]])
def test_reduced_histories_equal_oracle(snn, lattices):
    """One lattice below a chunk, three lattices (one spanning two chunks, one of a single neuron), and a lattice
    of 7 chunks with a ragged last one; no voltage history is kept on the device."""
    net = build(lattices, st=[(9, 2, 3)] if len(lattices) == 3 else ())
    steps = 400
    dn = parity.device_from_oracle(snn, net)
    dn.set_reduced_history(average_voltage=True, eeg=True, spike_counts=True)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    assert dn.history_steps() == steps
    net.run(steps, voltage_history=True, summaries=True, spike_counts=True)
    check(dn, net, steps)
    assert net.spike_counts.sum() > 10
    # the oracle's own reduction against float64 numpy
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        first, count, _ = net.layout.ranges()[i]
        v = net.voltage_history[:, first:first + count].astype(np.float64)
        np.testing.assert_allclose(net.avg_history[:, slot], v.mean(axis=1), rtol=2e-5, atol=1e-4)
        eeg = (v - 0.007).sum(axis=1) / (4 * np.pi * 251.0 * 0.8)
        np.testing.assert_allclose(net.eeg_history[:, slot], eeg, rtol=2e-5, atol=1e-4)
    dn.close()


def test_eeg_parameters_and_reset(snn):
    """Non-default EEG constants; snn_reset_history clears the rows and the spike totals; the reduced rows share
    the step axis with the voltage history when both are on."""
    net = build([(0, 6, 6), (1, 17, 17)], seed=5)
    net.eeg_reference_voltage, net.eeg_distance, net.eeg_conductivity = -20.0, 1.3, 100.0
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=False)
    dn.set_reduced_history(True, True, True, reference_voltage=-20.0, distance=1.3, conductivity=100.0)
    dn.run(120)
    net.run(120, summaries=True, spike_counts=True)
    check(dn, net, 120)
    assert dn.voltage_history(1).shape == (120, 17 * 17)
    dn.reset_history()
    assert dn.history_steps() == 0 and dn.spike_counts(1).sum() == 0
    dn.run(60)
    net.spike_counts = None
    net.run(60, summaries=True, spike_counts=True)
    check(dn, net, 60)
    dn.close()


def test_reduced_history_errors(snn):
    net = build([(0, 4, 4)], st=[(1, 2, 2)])
    dn = parity.device_from_oracle(snn, net)
    dn.run(3)
    with pytest.raises(snn.SnnError):
        dn.average_voltage_history(0)            # off
    dn.set_reduced_history(average_voltage=True)
    dn.run(3)
    assert dn.average_voltage_history(0).shape == (3,)
    with pytest.raises(snn.SnnError):
        dn.eeg_history(0)                        # only the average is on
    with pytest.raises(snn.SnnError):
        dn.average_voltage_history(1)            # a spike-train lattice has no voltage reduction
    with pytest.raises(snn.SnnError):
        dn.average_voltage_history(42)
    assert dn.spike_counts(0).sum() == 0         # counting is off: totals stay zero
    dn.close()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_sharded_handles_reduce_over_the_whole_lattice(snn, n_shards):
    """Every shard handle holds the full exchanged voltage plane after snn_step_end, so each reports the
    complete per-lattice reductions; spike totals are kept for the handle's own neurons."""
    import torch
    from snn_amd import parallel
    net = build([(0, 9, 10), (3, 20, 20)], st=[(5, 3, 4)], seed=1)
    steps = 300
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    for h in handles:
        h.set_reduced_history(True, True, True)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    for _ in range(steps):
        ex.step()
    net.run(steps, summaries=True, spike_counts=True)
    assert net.spike_counts.sum() > 10
    rng = net.layout.ranges()
    for h in handles:
        for slot, (i, _, _) in enumerate(net.layout.lattices):
            assert np.array_equal(parity.bits(h.average_voltage_history(i)), parity.bits(net.avg_history[:, slot]))
            assert np.array_equal(parity.bits(h.eeg_history(i)), parity.bits(net.eeg_history[:, slot]))
            first, count, _ = rng[i]
            b, e = max(first, h.post_begin), min(first + count, h.post_end)
            if b < e:
                assert np.array_equal(h.spike_counts(i)[b - first:e - first], net.spike_counts[b:e])
        h.close()


def chunked_sum_f32(v):
    """Canonical order with numpy: sequential float32 sum inside each 256-chunk (accumulate is sequential),
    then the chunk partials added in ascending order."""
    steps, n = v.shape
    pad = (-n) % 256
    v = np.concatenate([v, np.zeros((steps, pad), np.float32)], axis=1).reshape(steps, -1, 256)
    # appended +0.0f terms leave a partial unchanged (partials start at +0.0f and so never hold -0.0f)
    parts = np.add.accumulate(v, axis=2, dtype=np.float32)[:, :, -1]
    return np.add.accumulate(parts, axis=1, dtype=np.float32)[:, -1]


def test_lattice_of_more_than_256_chunks(snn):
    """512 x 160 neurons = 320 chunks: the reduction workgroup loops over its 256-chunk window.  Sparse handle
    with a ring graph (the oracle needs the dense matrix, 27 GB here); the device's own voltage history, reduced
    with numpy in the canonical order, is the expected value."""
    rows, cols = 512, 160
    n = rows * cols
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
    dn.add_lattice(0, rows, cols)
    dn.finalize(csr=True)
    row_ptr = np.arange(n + 1, dtype=np.uint64)
    pre = ((np.arange(n) + 1) % n).astype(np.uint32)
    dn.set_graph_csr(row_ptr, pre, np.ones(n, np.float32))
    dn.set_attr(0, "current_voltage", ob.uniform_array(11, n, -65.0, 30.0))
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
    dn.set_history(voltage=True, spikes=True)
    dn.set_reduced_history(True, True, True)
    dn.run(40)
    v = dn.voltage_history(0)
    tot = chunked_sum_f32(v)
    assert np.array_equal(parity.bits(dn.average_voltage_history(0)), parity.bits(tot / np.float32(n)))
    k = np.float32(1) / (np.float32(4) * np.float32(np.pi) * np.float32(251.0) * np.float32(0.8))
    tot_e = chunked_sum_f32(v - np.float32(0.007))
    assert np.array_equal(parity.bits(dn.eeg_history(0)), parity.bits(k * tot_e))
    assert np.array_equal(dn.spike_counts(0), dn.spike_history(0).sum(axis=0).astype(np.uint32))
    assert dn.spike_counts(0).sum() > 10
    dn.close()


@pytest.mark.parametrize("every", [1, 3, 7])
def test_strided_capture(snn, every):
    """snn_set_history_stride: rows of steps 0, every, 2*every, ... of the oracle's full record, across split
    runs whose lengths are not multiples of the stride; spike totals still count every step."""
    net = build([(0, 9, 10), (3, 12, 12)], st=[(5, 2, 3)], seed=1)
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.set_reduced_history(True, True, True)
    dn.set_history_stride(every)
    for part in (100, 1, 149, 50):
        dn.run(part)
    steps = 300
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=True, summaries=True, spike_counts=True)
    assert net.spike_counts.sum() > 10
    keep = np.arange(0, steps, every)
    assert dn.history_steps() == keep.size
    rng = net.layout.ranges()
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        first, count, _ = rng[i]
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[keep, first:first + count]))
        assert np.array_equal(dn.spike_history(i), net.spike_history[keep, first:first + count])
        assert np.array_equal(parity.bits(dn.average_voltage_history(i)), parity.bits(net.avg_history[keep, slot]))
        assert np.array_equal(parity.bits(dn.eeg_history(i)), parity.bits(net.eeg_history[keep, slot]))
        assert np.array_equal(dn.spike_counts(i), net.spike_counts[first:first + count])
    first, count, _ = rng[5]
    assert np.array_equal(parity.bits(dn.voltage_history(5)), parity.bits(net.st_voltage_history[keep, first:first + count]))
    with pytest.raises(snn.SnnError):
        dn.set_history_stride(0)
    dn.close()


@pytest.mark.parametrize("plasticity,order", [("stdp", 1), ("reward", 1), ("stdp", 2)])
def test_graph_history(snn, plasticity, order):
    """update_graph_history: the lattice's internal weights after every recorded step (stride 3), for STDP and for a
    reward-modulated lattice (which then keeps its weight update as a standalone pass), next to a second lattice
    without history; the oracle is stepped one step at a time and its weights snapshotted."""
    net = build([(0, 6, 7), (3, 5, 5)], seed=3)
    if plasticity == "stdp":
        net["do_plasticity"] = 1
    else:
        net["rm_do_modulation"][0] = 1
        net["rm_tau_c"][0] = 0.05
        net["rm_a_plus"][0] = 0.01
        net["rm_a_minus"][0] = 0.01
        net["rm_dopamine"][0] = 0.02
    steps, every = 600, 3
    dn = parity.device_from_oracle(snn, net)
    if plasticity == "reward":
        dn.set_reward_modulator(0, dopamine=0.02, tau_c=0.05, a_plus=0.01, a_minus=0.01)
    dn.set_graph_history(0, order)          # 1: after the step's weight updates (Lattice), 2: before (LatticeNetwork)
    dn.set_history_stride(every)
    dn.run(steps)
    n0 = 6 * 7
    snaps = []
    for t in range(steps):
        if order == 1:
            net.run(1)
        if t % every == 0:
            snaps.append(np.where(net["connections"][:n0, :n0] != 0, net["weights"][:n0, :n0], np.float32(0)).copy())
        if order == 2:
            net.run(1)
    hist = dn.graph_history(0)
    assert hist.shape == (len(snaps), n0, n0)
    assert np.array_equal(parity.bits(hist), parity.bits(np.array(snaps)))
    assert not np.array_equal(hist[0], hist[-1]), "weights must have moved"
    with pytest.raises(snn.SnnError):
        dn.graph_history(3)                      # off for that lattice
    dn.close()

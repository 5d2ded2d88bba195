"""STDP under load: the synthetic drive (snn_set_synthetic_drive) makes a chosen fraction of the population spike on
every step, so the spike compaction and the column / row weight updates are exercised at rates the quiescent parity
networks never reach.  The oracle gets the same drive from the numpy twin of the generator before each of its steps.
Dense, sparse and sharded handles; weights, rasters and firing times bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

V_KICK = np.float32(35.0)


def drive_oracle(snn, net, seed, fraction):
    nn = net.n_neurons
    thr = min(4294967295, int(np.float64(np.float32(fraction)) * 4294967296.0))
    h = snn.synthetic.hash32(seed, np.uint64(net.clock) * np.uint64(nn) + np.arange(nn, dtype=np.uint64))
    net["current_voltage"][h < np.uint32(thr)] = V_KICK


def build(seed=5):
    lay = parity.Layout([(0, 12, 13), (2, 9, 9)], [(7, 3, 3)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON)
    nn, nc = net.n_neurons, net.n_cells
    net["gap_conductance"] = 4.0
    net["current_voltage"] = ob.uniform_array(seed, nn, -70.0, 0.0)
    net["st_chance_of_firing"] = 0.05
    net["st_seed"] = np.arange(11, 11 + nc, dtype=np.uint32)
    net.fill_graph(seed + 1, 0.2, 1.0)
    rng = np.random.default_rng(seed)
    net["connections"][rng.random(net["connections"].shape) < 0.4] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    net["stdp_a_plus"][1] = 1.5
    net["stdp_tau_minus"][1] = 3.0
    return net


@pytest.mark.parametrize("csr", [False, True])
@pytest.mark.parametrize("fraction", [0.01, 0.25])
def test_stdp_under_synthetic_drive(snn, csr, fraction):
    net = build()
    steps, seed = 220, 99
    dn = parity.device_from_oracle(snn, net, csr=csr)
    dn.set_history(voltage=False, spikes=True)
    dn.set_synthetic_drive(seed, fraction, V_KICK)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    raster = []
    w0 = net["weights"].copy()
    for _ in range(steps):
        drive_oracle(snn, net, seed, fraction)
        net.run(1)
        raster.append(net["is_spiking"].copy())
    raster = np.array(raster)
    rate = raster.mean()
    assert rate >= 0.6 * fraction, rate                     # the drive did make that share of the population spike
    assert not np.array_equal(w0, net["weights"])
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), raster[:, first:first + count]), i
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    dn.close()


def test_stdp_under_drive_on_shard_handles(snn):
    import torch
    from snn_amd import parallel
    net = build(seed=8)
    steps, seed, fraction, g = 150, 7, 0.1, 3
    handles = [parity.device_from_oracle(snn, net, shard=(r, g)) for r in range(g)]
    for h in handles:
        h.set_synthetic_drive(seed, fraction, V_KICK)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    for _ in range(steps):
        ex.step()
    for _ in range(steps):
        drive_oracle(snn, net, seed, fraction)
        net.run(1)
    for h in handles:
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        b, e = h.post_begin, h.post_end
        w, c = h.get_graph_rows(0, net.n_tot)
        ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()


def build_streamed(side_rows, side_cols, seed, st=True):
    """a lattice pair big enough for the STREAMED input pass (matrix > 64 MiB): there the STDP update of step t is
    applied by the input pass of step t + 1 (k_inputs_dense<..., STDP>), not by the scatter kernels"""
    lay = parity.Layout([(0, side_rows, side_cols), (3, 20, 20)], [(7, 4, 4)] if st else [])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON if st else ob.ST_NONE)
    nn, nc = net.n_neurons, net.n_cells
    net["gap_conductance"] = 3.0
    net["current_voltage"] = ob.uniform_array(seed, nn, -70.0, 0.0)
    if nc:
        net["st_chance_of_firing"] = 0.2
        net["st_seed"] = np.arange(11, 11 + nc, dtype=np.uint32)
    net.fill_graph(seed + 1, 0.2, 1.0)
    rng = np.random.default_rng(seed)
    net["connections"][rng.random(net["connections"].shape) < 0.2] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    net["stdp_a_plus"][1] = 1.5
    net["stdp_tau_minus"][1] = 3.0
    net.n_threads = 8
    return net


@pytest.mark.parametrize("mode", [1, 3])
def test_deferred_stdp_in_the_streamed_input_pass_equals_oracle(snn, mode):
    """4 500 neurons (81 MB matrix): weights, state and rasters bit for bit with the oracle, with a host read of the
    weights in the middle (which applies the pending update through the standalone pass) and a resumed run.
    mode 1: the whole update rides on the next input pass; mode 3: its row half only (the columns are scattered at once)."""
    net = build_streamed(64, 64, seed=41)
    steps, seed, fraction = 36, 5, 0.02
    dn = parity.device_from_oracle(snn, net)
    dn.set_option("defer_stdp", mode)                       # the update rides on the next input pass
    dn.set_synthetic_drive(seed, fraction, V_KICK)
    w0 = net["weights"].copy()
    done = 0
    for k in (1, 9, 14):
        dn.run(k)
        for _ in range(k):
            drive_oracle(snn, net, seed, fraction)
            net.run(1)
        done += k
        parity.assert_graph_equal(net, dn)                 # host read: flushes the pending update
        parity.assert_state_equal(net, parity.pull_state(dn, net))
    for _ in range(12):                                    # one call per step: the update stays pending ACROSS calls
        dn.run(1)
        drive_oracle(snn, net, seed, fraction)
        net.run(1)
        done += 1
    parity.assert_graph_equal(net, dn)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    assert done == steps and not np.array_equal(w0, net["weights"])
    assert (net["last_firing_time"] >= 0).sum() > 200
    dn.close()


def test_deferred_stdp_on_streamed_shard_handles(snn):
    import torch
    from snn_amd import parallel
    net = build_streamed(90, 90, seed=43)          # 8 500 neurons: each of the two shards streams a 145 MB matrix
    steps, seed, fraction, g = 14, 9, 0.02, 2
    handles = [parity.device_from_oracle(snn, net, shard=(r, g)) for r in range(g)]
    for r, h in enumerate(handles):
        h.set_option("defer_stdp", (1, 3)[r % 2])           # one shard fuses the whole update, the other its row half
        h.set_synthetic_drive(seed, fraction, V_KICK)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    for _ in range(steps):
        ex.step()
    for _ in range(steps):
        drive_oracle(snn, net, seed, fraction)
        net.run(1)
    for h in handles:
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        b, e = h.post_begin, h.post_end
        w, c = h.get_graph_rows(0, net.n_tot)
        ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()


def test_c4_size_deferred_stdp_equals_the_standalone_kernels(snn):
    """BASELINE configs[3] (81 920 neurons, 26.8 GB matrix) with 1 % of the population spiking per step: the update
    fused into the input pass (SNN_AMD_DEFER_STDP=1) against the scatter kernels (=0, the default), device against device; sampled rows
    and columns of the matrix and the whole state bit-identical."""
    import os
    n_inh, n_exc = 128 * 128, 256 * 256
    n = n_inh + n_exc
    out = []
    for defer in ("1", "0", "3"):
        old = os.environ.get("SNN_AMD_DEFER_STDP")
        os.environ["SNN_AMD_DEFER_STDP"] = defer
        try:
            dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
        finally:
            if old is None:
                del os.environ["SNN_AMD_DEFER_STDP"]
            else:
                os.environ["SNN_AMD_DEFER_STDP"] = old
        dn.add_lattice(0, 128, 128)
        dn.add_lattice(1, 256, 256)
        dn.finalize()
        for i, m in ((0, n_inh), (1, n_exc)):
            dn.set_attr(i, "gap_conductance", np.full(m, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", snn.synthetic.uniform(4, n_inh, -65.0, 30.0))
        dn.set_attr(1, "current_voltage", snn.synthetic.uniform(4, n_exc, -65.0, 30.0, offset=n_inh))
        dn.fill_graph_synthetic(5, 0.5, 1.5, with_diagonal=False)
        dn.set_plasticity(0, a_plus=1.5)
        dn.set_plasticity(1)
        dn.set_synthetic_drive(77, 0.01, 35.0)
        # a neuron the drive kicks at clock 0: it spikes in the first step, so its row and column change from then on
        j0 = int(np.flatnonzero(snn.synthetic.hash32(77, np.arange(n, dtype=np.uint64)) < np.uint32(0.01 * 4294967296.0))[0])
        w_init = dn.get_graph_rows(j0, 1)[0][0].copy()
        dn.run(6)
        dn.run(5)
        rows = [0, 1, 255, 256, 16383, 16384, j0, n - 1]
        w = np.stack([dn.get_graph_rows(p, 1)[0][0] for p in rows])
        v = np.concatenate([dn.get_attr(i, "current_voltage") for i in (0, 1)])
        lft = np.concatenate([dn.get_attr(i, "last_firing_time", dtype=np.int32) for i in (0, 1)])
        out.append((w, v, lft, w_init))
        dn.close()
    a, b, c = out
    assert (a[2] >= 0).sum() > 3000, "the drive must have made thousands of neurons spike"
    assert (a[0][6] != a[3]).mean() > 0.01, "the outgoing / incoming weights of spiking neurons must have moved"
    for other in (b, c):
        for x, y in zip(a, other):
            assert np.array_equal(parity.bits(np.asarray(x)), parity.bits(np.asarray(y)))

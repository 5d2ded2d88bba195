"""The HIP sharded path (snn_network_finalize_shard / snn_step_begin / snn_step_end / exchange plan) on
ONE GPU: G shard handles on cuda:0, the all-gather emulated by device-to-device copies of each shard's
packed segment (voltage / live transmitter planes / spike bitmap).  Merged result must equal the oracle bit for bit -- rasters, voltages, adaptation
variables, last_firing_time and the STDP-updated weight columns."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(chemical):
    lay = parity.Layout([(0, 9, 10), (3, 12, 12)], [(5, 3, 4)])
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE, chemical=chemical)
    nn, nc = net.n_neurons, net.n_cells
    net["current_voltage"] = ob.uniform_array(1, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 2.0
    net["st_nt_flags"][:, 0] = 1
    net["st_rate"] = ob.uniform_array(4, nc, 1.0, 5.0)
    net.fill_graph(2, 0.5, 1.5)
    rng = np.random.default_rng(3)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["do_plasticity"] = 1
    return net


@pytest.mark.parametrize("n_shards,chemical", [(2, False), (3, True), (8, False)])
def test_sharded_handles_equal_oracle(snn, n_shards, chemical):
    import torch
    from snn_amd import parallel
    net = build(chemical)
    steps = 300
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    stride, shards = parallel.shard_geometry(net.n_neurons, n_shards)
    for h, (b, e) in zip(handles, shards):
        assert (h.post_begin, h.post_end) == (b, e)
        h.set_history(voltage=True, spikes=True)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    # what travels: 4 B voltage (+ 4 B for the one transmitter type in use) + 1 bit per neuron of every other slot
    planes = 2 if chemical else 1
    assert all(p["mode"] == "allgather" and p["planes"] == planes for p in ex.plans)
    assert ex.bytes_per_step() == [4 * (n_shards - 1) * (planes * stride + stride // 32)] * n_shards
    for _ in range(steps):
        ex.step()
    net.run(steps, voltage_history=True, spike_history=True)
    assert net.spike_history.sum() > 20

    rng = net.layout.ranges()
    for r, h in enumerate(handles):
        b, e = shards[r]
        assert h.clock == net.clock
        st = parity.pull_state(h, net)
        # what is exchanged (voltage, spikes -> last_firing_time, t when chemical synapses read it) is complete on
        # every shard; the replicated cells agree everywhere
        parity.assert_shard_view_equal(h, st, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time", "st_last_firing_time", "st_current_voltage",
                     "st_step") + (("nt_t",) if chemical else ()):
            assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), (r, name)
        # owned state
        for name in ("w_value", "rc_r", "rc_current"):
            assert np.array_equal(parity.bits(st[name][b:e]), parity.bits(net[name][b:e])), (r, name)
        w, c = h.get_graph_rows(0, net.n_tot)
        oc = net["connections"].astype(np.uint32)
        ow = np.where(oc != 0, net["weights"], np.float32(0))
        assert np.array_equal(c[:, b:e], oc[:, b:e])
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e])), f"weights of shard {r}"
        # histories hold the local neurons
        for i, _, _ in net.layout.lattices:
            first, count, _ = rng[i]
            lo, hi = max(first, b), min(first + count, e)
            if lo >= hi:
                continue
            vh = h.voltage_history(i)[:, lo - first:hi - first]
            assert np.array_equal(parity.bits(vh), parity.bits(net.voltage_history[:, lo:hi]))
            sh = h.spike_history(i)[:, lo - first:hi - first]
            assert np.array_equal(sh, net.spike_history[:, lo:hi])
    for h in handles:
        h.close()


def test_stream_ordered_stepping_without_host_sync(snn):
    """snn_set_stream: the shard handles adopt torch's current stream, so kernels and the (emulated) exchange
    are ordered on the device and the step loop never synchronises the host -- the shape bench.py --gpus N uses."""
    import torch
    from snn_amd import parallel
    net = build(False)
    n_shards, steps = 2, 250
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    side = torch.cuda.Stream()
    for h in handles:
        h.set_stream(side.cuda_stream)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    with torch.cuda.stream(side):
        for _ in range(steps):
            ex.step(sync=False)
    for h in handles:
        h.synchronize()
    net.run(steps)
    for h in handles:
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time", "st_last_firing_time"):
            assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), name
        b, e = h.post_begin, h.post_end
        w, c = h.get_graph_rows(0, net.n_tot)
        ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.set_stream(None)
        h.close()


@pytest.mark.parametrize("plastic", [False, True])
def test_split_input_pass_is_result_neutral(snn, plastic):
    """snn_step_begin_local + snn_step_begin (own-rows chunks first, the rest later) == one full pass; with
    plasticity on the split is refused internally and the schedule still yields the oracle's result."""
    import torch
    from snn_amd import parallel
    net = build(False)
    net["do_plasticity"] = int(plastic)
    n_shards, steps = 3, 200
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    gather = ex.exchange

    started = False
    for _ in range(steps):
        for h in handles:
            h.step_begin_local()          # before the previous step's exchange has been applied
        if started:
            gather()
            for h in handles:
                h.step_end()
        for h in handles:
            h.step_begin()
        started = True
    gather()
    for h in handles:
        h.step_end()
    net.run(steps)
    for h in handles:
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time", "st_last_firing_time"):
            assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), name
        b, e = h.post_begin, h.post_end
        assert np.array_equal(parity.bits(st["w_value"][b:e]), parity.bits(net["w_value"][b:e]))
        w, c = h.get_graph_rows(0, net.n_tot)
        ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()

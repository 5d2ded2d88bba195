"""Error behaviour of the C ABI on a live device: every misuse returns its documented status code and a
message, nothing aborts, the handle stays usable (the reference returns Result<_, GPUError> / GraphError and
panics only on missing buffers, gpu_lattices/mod.rs:817)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAD_ATTR, DIM_MISMATCH, BAD_ARG, BAD_STATE = 9, 10, 11, 12


def code(snn, fn):
    with pytest.raises(snn.SnnError) as e:
        fn()
    assert str(e.value).split(":", 1)[1].strip(), "an error message must accompany the code"
    return e.value.code


def test_call_order_and_argument_errors(snn):
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH, spike_train=snn.ST_RATE)
    assert code(snn, lambda: dn.set_attr(0, "current_voltage", np.zeros(4, np.float32))) == BAD_STATE   # not finalized
    assert code(snn, lambda: dn.run(1)) == BAD_STATE
    dn.add_lattice(0, 2, 2)
    assert code(snn, lambda: dn.add_lattice(0, 3, 3)) == BAD_ARG                  # GraphIDAlreadyPresent
    assert code(snn, lambda: dn.add_spike_train_lattice(0, 1, 1)) == BAD_ARG      # ids are shared with spike trains
    dn.add_spike_train_lattice(5, 1, 3)
    dn.finalize()
    assert code(snn, lambda: dn.add_lattice(9, 1, 1)) == BAD_STATE                # after finalize
    assert code(snn, lambda: dn.finalize()) == BAD_STATE
    v = np.zeros(4, np.float32)
    assert code(snn, lambda: dn.set_attr(0, "no_such_field", v)) == BAD_ATTR
    assert code(snn, lambda: dn.set_attr(0, "current_voltage", v.astype(np.uint32))) == BAD_ATTR   # wrong scalar type
    assert code(snn, lambda: dn.set_attr(0, "current_voltage", np.zeros(5, np.float32))) == DIM_MISMATCH
    assert code(snn, lambda: dn.set_attr(0, "neurotransmitters$t", v)) == DIM_MISMATCH             # needs 3 per cell
    assert code(snn, lambda: dn.set_attr(7, "current_voltage", v)) == BAD_ARG                      # unknown lattice
    assert code(snn, lambda: dn.set_attr(5, "w_value", np.zeros(3, np.float32))) == BAD_ATTR       # neuron field on a spike train
    assert code(snn, lambda: dn.set_attr(0, "rate", v)) == BAD_ATTR                                # and vice versa
    assert code(snn, lambda: dn.set_attr(0, "v_reset", v)) == BAD_ATTR                             # LIF field on Izhikevich
    assert code(snn, lambda: dn.set_plasticity(5)) == BAD_ARG                                      # plasticity of a spike train
    w = np.zeros((6, 6), np.float32)
    assert code(snn, lambda: dn.set_graph_dense(w, w.astype(np.uint32))) == DIM_MISMATCH            # n_tot is 7
    assert dn._L.snn_get_voltage_history(dn._h, 0, None, 0) == BAD_STATE                           # history is off
    assert code(snn, lambda: dn.set_graph_csr([0, 0, 0, 0, 0], [], [])) == BAD_STATE                # dense handle
    assert code(snn, lambda: dn.step_end()) == BAD_STATE                                            # no step open
    # the handle is still fully usable
    dn.set_attr(0, "current_voltage", np.array([29.0, 29.5, 29.9, 29.99], np.float32))
    dn.set_history(voltage=True, spikes=True)
    dn.run(3)
    assert dn.clock == 3 and dn.voltage_history(0).shape == (3, 4) and dn.voltage_history(5).shape == (3, 3)
    buf = np.zeros(5, np.float32)
    assert dn._L.snn_get_voltage_history(dn._h, 0, buf.ctypes.data_as(snn._lib.f32p), 5) == DIM_MISMATCH   # 3 x 4 expected
    dn.close()
    dn.close()                                                                     # idempotent


def test_bad_device_and_selectors(snn):
    L = snn._lib.load()
    h = snn._lib.H()
    assert L.snn_network_create(4096, 0, 0, 0, 0, C.byref(h)) == 7                 # GetDeviceFailure
    assert L.snn_network_create(-1, 0, 0, 0, 0, C.byref(h)) == 7
    assert L.snn_network_create(0, 9, 0, 0, 0, C.byref(h)) == BAD_ARG
    assert L.snn_network_create(0, 0, 4, 0, 0, C.byref(h)) == BAD_ARG
    assert L.snn_network_create(0, 0, 0, 3, 0, C.byref(h)) == BAD_ARG
    assert L.snn_network_create(0, 0, 0, 0, 5, C.byref(h)) == BAD_ARG
    assert b"selector" in L.snn_last_error()
    dn = snn.DeviceNetwork()                                                        # no spike-train model
    assert code(snn, lambda: dn.add_spike_train_lattice(1, 2, 2)) == BAD_STATE
    dn.close()
    # firing times only exist for the preset spike-train model; the pointer array must span the time array
    dn = snn.DeviceNetwork(spike_train=snn.ST_RATE)
    dn.add_lattice(0, 1, 1)
    dn.add_spike_train_lattice(1, 1, 2)
    dn.finalize()
    assert code(snn, lambda: dn.set_firing_times(1, [0, 1, 2], [1.0, 2.0])) == BAD_STATE
    dn.close()
    dn = snn.DeviceNetwork(spike_train=snn.ST_PRESET)
    dn.add_lattice(0, 1, 1)
    dn.add_spike_train_lattice(1, 1, 2)
    dn.finalize()
    assert code(snn, lambda: dn.set_firing_times(1, [0, 1, 3], [1.0, 2.0])) == DIM_MISMATCH
    assert code(snn, lambda: dn.set_firing_times(1, [0, 2, 1], [1.0])) == BAD_ARG
    assert code(snn, lambda: dn.set_firing_times(0, [0, 1], [1.0])) == BAD_ARG        # a neuron lattice
    dn.set_firing_times(1, [0, 1, 2], [1.0, 2.0])
    dn.close()


def test_sharded_handle_refuses_whole_population_run(snn):
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 16, 16)
    dn.finalize(1, 2)
    assert (dn.post_begin, dn.post_end) == (128, 256)
    assert code(snn, lambda: dn.run(1)) == BAD_STATE
    assert dn._L.snn_network_finalize_shard(dn._h, 0, 2) == BAD_STATE              # already finalized
    dn.close()
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 4, 4)
    assert code(snn, lambda: dn.finalize(3, 2)) == BAD_ARG                          # shard_index >= n_shards
    dn.close()


def test_multi_gpu_entry_points_refuse_misuse(snn):
    """exchange plan / halo / by-lattice sharding / options / drive: documented codes, handles stay usable"""
    import torch
    from snn_amd import parallel
    whole = snn.DeviceNetwork()
    whole.add_lattice(0, 4, 4)
    assert code(snn, lambda: whole.finalize(0, 2, by_lattice=True)) == BAD_STATE       # by lattice needs a sparse handle
    whole.finalize()
    assert code(snn, lambda: whole.exchange_plan()) == BAD_STATE                       # not a shard handle
    assert code(snn, lambda: whole.halo_needs(0)) == BAD_STATE
    assert code(snn, lambda: whole.run_sharded(12345, 1)) == BAD_STATE
    assert code(snn, lambda: whole.set_option("no_such_switch", 1)) == BAD_ARG
    assert code(snn, lambda: whole.set_synthetic_drive(1, 1.5, 35.0)) == BAD_ARG       # fraction outside [0, 1]
    whole.set_option("input_shape", 2)
    whole.set_option("defer_stdp", 0)
    whole.run(3)
    assert whole.clock == 3 and whole.ranges == [(0, 16)] and whole.cells_read().size == 0
    whole.close()

    dense = snn.DeviceNetwork()
    dense.add_lattice(0, 16, 16)
    dense.finalize(1, 2)
    assert code(snn, lambda: dense.run(1)) == BAD_STATE                                # sharded handles are stepped with an exchange
    assert code(snn, lambda: dense.halo_needs(0)) == BAD_STATE                         # halo plans belong to sparse handles
    assert code(snn, lambda: dense.halo_commit()) == BAD_STATE
    plan = dense.exchange_plan()
    assert plan["mode"] == "allgather" and plan["n_shards"] == 2 and plan["shard_index"] == 1 and plan["plane_id"] == [0]
    dense.set_synapses(False, True)                                                    # nothing releases a transmitter:
    assert dense.exchange_plan()["planes"] == 0                                        # only the spike bits travel
    dense.close()

    handles = []
    for r in range(2):
        h = snn.DeviceNetwork(spike_train=snn.ST_POISSON)
        for k in range(2):
            h.add_lattice(k, 16, 16)
        h.add_spike_train_lattice(5, 2, 2)
        h.finalize(r, 2, csr=True, by_lattice=True)
        handles.append(h)
    a, b = handles
    assert a.ranges == [(0, 128), (256, 384)] and b.ranges == [(128, 256), (384, 512)]
    assert code(snn, lambda: a.halo_needs(2)) == BAD_ARG                               # peer out of range
    assert code(snn, lambda: a.halo_set_sends(0, [1])) == BAD_ARG                      # a handle is not its own peer
    assert code(snn, lambda: a.halo_set_sends(1, [200])) == DIM_MISMATCH               # neuron 200 belongs to the peer
    with pytest.raises(ValueError):
        a.set_graph_csr([0, 1], [3], [1.0])                                            # one row per OWNED neuron (binding check)
    rp = np.zeros(257, np.uint64)
    rp[1:] = 2                                                                         # ... and row_ptr must end at nnz
    assert code(snn, lambda: a.set_graph_csr(rp, [3], [1.0])) == DIM_MISMATCH
    assert code(snn, lambda: a.set_reduced_history(True, False, False) or a.step_begin()) == BAD_STATE   # needs every voltage
    a.set_reduced_history(False, False, False)
    # without rows or a committed plan the two handles trade whole ownerships and still step
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
    assert [p["mode"] for p in ex.plans] == ["halo", "halo"] and ex.bytes_per_step() == [4 * (256 + 8), 4 * (256 + 8)]
    for _ in range(3):
        ex.step()
    assert a.clock == 3 and b.clock == 3
    for h in handles:
        h.close()


def test_a_nan_weight_on_a_connected_edge_is_refused(snn):
    """Some(NaN) is an edge in the reference (graph/mod.rs:204-213); the device matrix marks ABSENT edges with NaN, so such an edge
    cannot be stored: both graph forms refuse it with SNN_ERR_BAD_ARG instead of silently dropping it"""
    import numpy as np
    n = 12
    w = np.full((n, n), 0.5, np.float32)
    c = np.ones((n, n), np.uint32)
    w[3, 7] = np.nan
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 3, 4)
    dn.finalize()
    with pytest.raises(snn.SnnError) as e:
        dn.set_graph_dense(w, c)
    assert e.value.code == 11 and "NaN" in str(e.value) and "pre 3" in str(e.value) and "post 7" in str(e.value)
    c[3, 7] = 0                                    # the same weight on an ABSENT edge is nobody's business
    dn.set_graph_dense(w, c)
    got_w, got_c = dn.get_graph_dense()
    assert got_c[3, 7] == 0 and got_c.sum() == n * n - 1
    dn.close()
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 3, 4)
    dn.finalize(csr=True)
    ptr = np.arange(0, n * 2 + 1, 2, dtype=np.uint64)
    pre = np.tile(np.array([1, 5], np.uint32), n)
    ws = np.ones(2 * n, np.float32)
    ws[9] = np.nan
    with pytest.raises(snn.SnnError) as e:
        dn.set_graph_csr(ptr, pre, ws)
    assert e.value.code == 11 and "NaN" in str(e.value)
    ws[9] = 1.0
    dn.set_graph_csr(ptr, pre, ws)
    dn.close()


def test_counter_rows_check_their_range_before_they_allocate(snn):
    import ctypes
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 4, 4)
    dn.finalize()
    buf = (ctypes.c_uint8 * 16)()
    assert dn._L.snn_set_counter_rows(dn._h, 0, 0xFFFFFFF0, buf) == 10          # SNN_ERR_DIM_MISMATCH, not std::bad_alloc
    assert dn._L.snn_get_counter_rows(dn._h, 10, 0xFFFFFFF0, buf) == 10
    dn.close()


@pytest.mark.parametrize("pinned", [0, 1])
def test_setters_and_getters_agree_with_and_without_the_page_locked_buffer(snn, pinned):
    """option "pinned_copies": the same bytes arrive whether a transfer goes through the handle's page-locked buffer (default) or the
    caller's pointer is handed to the runtime (0)"""
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
    dn.add_lattice(0, 20, 30)
    dn.finalize()
    dn.set_option("pinned_copies", pinned)
    n = 600
    rng = np.random.default_rng(5)
    v = rng.uniform(-70, 20, n).astype(np.float32)
    dn.set_attr(0, "current_voltage", v)
    assert np.array_equal(dn.get_attr(0, "current_voltage"), v)
    flags = (rng.random((n, 3)) < 0.5).astype(np.uint32)
    dn.set_attr(0, "neurotransmitters$flags", flags)
    assert np.array_equal(dn.get_attr(0, "neurotransmitters$flags", dtype=np.uint32, per_type=True), flags)
    dn.fill_graph_synthetic(3, 0.5, 1.5)
    dn.set_synapses(True, False)
    dn.set_history(voltage=True, spikes=True)
    dn.run(12)
    hist, spikes = dn.voltage_history(0), dn.spike_history(0)
    assert hist.shape == (12, n) and spikes.shape == (12, n) and np.isfinite(hist).all()
    assert np.array_equal(dn.get_attr(0, "current_voltage"), hist[-1])        # (a row of the history is the state after its step)
    dn.close()

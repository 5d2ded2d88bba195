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

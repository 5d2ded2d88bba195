"""Generated neuron models on the device (SNN_MODEL_CUSTOM): the description is parsed, turned into HIP, compiled into
its own library (`_lib.build_custom`) and stepped through the same C ABI.  Checked against (a) the hand expansion the
reference's own test compares its generated code with (build_test/nb_macro/tests/basic_lif.rs:22-50 with
tests/lif_reference.rs) and (b) a numpy float32 interpreter of the description inside the canonical lattice step."""
import numpy as np
import pytest

import numpy_ref as nr
import oracle_binding as ob
import modelgen_ref
import parity
from test_modelgen import CURRENTS, EXPECTED_FLAG, IF_DSL, IZH_DSL, LIF_NB, lif_reference_trace

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def libs(snn):
    from snn_amd import _lib, modelgen
    out = {}
    for text in (LIF_NB, IZH_DSL, IF_DSL):
        m = modelgen.parse(text)
        out[m.name] = (m, _lib.build_custom(m))
    return out


def test_reference_known_answer_for_the_generated_lif(snn, libs):
    """11 neurons of the reference's lif.nb model, each fed a constant current by a never-firing spike-train cell
    (its v_resting enters the sum without the conductance factor, neuron/mod.rs:126-128): 1000 iterations, voltages
    and spikes equal to ReferenceIntegrateAndFire for every input current of basic_lif.rs."""
    model, lib = libs["BasicIntegrateAndFire"]
    currents = np.array([-50., -40., -30., -20., -10., 0., 10., 20., 30., 40., 50.], f32)
    n = currents.size
    dn = snn.DeviceNetwork(model=snn.CUSTOM, spike_train=snn.ST_RATE, lib_path=lib)
    assert dn._L.snn_custom_model() == b"BasicIntegrateAndFire"
    dn.add_lattice(1, 1, n)
    dn.add_spike_train_lattice(0, 1, n)
    dn.finalize()
    assert np.array_equal(dn.get_attr(1, "v_th"), np.full(n, -55.0, f32))            # DSL defaults
    assert np.array_equal(dn.get_attr(1, "current_voltage"), np.zeros(n, f32))
    assert np.array_equal(dn.get_attr(1, "gap_conductance"), np.full(n, 10.0, f32))
    dn.set_attr(0, "v_resting", currents)
    w = np.zeros((2 * n, n), f32)
    c = np.zeros((2 * n, n), np.uint32)
    w[n + np.arange(n), np.arange(n)] = 1.0
    c[n + np.arange(n), np.arange(n)] = 1
    dn.set_graph_rows(0, w, c)
    dn.set_history(voltage=True, spikes=True)
    dn.run(1000)
    vh, sh = dn.voltage_history(1), dn.spike_history(1)
    v = np.zeros(n, f32)
    with np.errstate(over="ignore", invalid="ignore"):
        for t in range(1000):
            dv = (((v - f32(0.0)).astype(f32) + currents).astype(f32) * f32(0.1)).astype(f32)
            v = (v + dv).astype(f32)
            spike = v >= f32(-55.0)
            v = np.where(spike, f32(-75.0), v).astype(f32)
            assert np.array_equal(sh[t].astype(bool), spike), t
            assert np.array_equal(parity.bits(vh[t]), parity.bits(v)), t
    assert sh.sum() >= n                           # every neuron starts above v_th, fires once and then runs away
    dn.close()


def test_if_statements_known_answers(snn, libs):
    """The nested if / elseif / else model of build_test/nb_macro/tests/if_statements.rs on the device: the flag every
    input current must leave (its assert_eq! lines) and the plain LIF voltages."""
    model, lib = libs["ElseIfNestedBasicIntegrateAndFire"]
    n = CURRENTS.size
    dn = snn.DeviceNetwork(model=snn.CUSTOM, spike_train=snn.ST_RATE, lib_path=lib)
    dn.add_lattice(1, 1, n)
    dn.add_spike_train_lattice(0, 1, n)
    dn.finalize()
    assert np.array_equal(dn.get_attr(1, "flag"), np.zeros(n, f32))
    dn.set_attr(0, "v_resting", CURRENTS)
    w = np.zeros((2 * n, n), f32)
    c = np.zeros((2 * n, n), np.uint32)
    w[n + np.arange(n), np.arange(n)] = 1.0
    c[n + np.arange(n), np.arange(n)] = 1
    dn.set_graph_rows(0, w, c)
    dn.set_history(voltage=True, spikes=False)
    dn.run(1)
    assert np.array_equal(dn.get_attr(1, "flag"), EXPECTED_FLAG)
    dn.run(299)
    assert np.array_equal(dn.get_attr(1, "flag"), EXPECTED_FLAG)
    assert np.array_equal(parity.bits(dn.voltage_history(1)), parity.bits(lif_reference_trace(300)))
    dn.close()


def test_generated_izhikevich_lattice_equals_the_interpreter(snn, libs):
    """A 6x7 lattice of an Izhikevich neuron written in the DSL, random dense gap junctions, heterogeneous variables:
    raster, voltages and the model's own variables bit-identical to the numpy interpreter inside numpy_ref.run_lattice
    (canonical chunked input sums)."""
    model, lib = libs["DslIzhikevich"]
    n = 42
    net = ob.Net(n)                                  # only as a container for the synthetic graph
    net.fill_graph(5, 0.5, 1.5)
    rng = np.random.default_rng(5)
    net["connections"][rng.random((n, n)) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    v0 = ob.uniform_array(6, n, -65.0, 30.0)
    a = ob.uniform_array(7, n, 0.01, 0.05)
    dn = snn.DeviceNetwork(model=snn.CUSTOM, lib_path=lib)
    dn.add_lattice(0, 6, 7)
    dn.finalize()
    assert np.array_equal(dn.get_attr(0, "c_m"), np.full(n, 100.0, f32)) and dn.get_attr(0, "w")[0] == f32(30.0)
    dn.set_attr(0, "current_voltage", v0)
    dn.set_attr(0, "a", a)
    dn.set_graph_rows(0, net["weights"], net["connections"].astype(np.uint32))
    dn.set_history(voltage=True, spikes=True)
    dn.run(400)
    dn.run(400)
    st = {"current_voltage": v0.copy(), "dt": np.full(n, 0.1, f32), "c_m": np.full(n, 100.0, f32),
          "gap_conductance": np.full(n, 10.0, f32)}
    for name, default in model.variables:
        st[name] = np.full(n, default, f32)
    st["a"] = a.copy()
    vh, sh, lft = nr.run_lattice(modelgen_ref.make_step(model), st, st["gap_conductance"].copy(), net["weights"].copy(),
                                 net["connections"].copy(), 800)
    assert sh.sum() > 20
    assert np.array_equal(dn.spike_history(0), sh)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(vh))
    assert np.array_equal(parity.bits(dn.get_attr(0, "w")), parity.bits(st["w"]))
    assert np.array_equal(dn.get_attr(0, "last_firing_time", dtype=np.int32), lft)
    dn.close()


@pytest.mark.parametrize("variant", ["electrical_stdp", "both_synapses", "chemical_only", "sparse", "sharded"])
def test_generated_model_in_a_network_equals_the_oracle(snn, libs, variant):
    """The DSL-written Izhikevich neuron through the whole path -- two lattices, Poisson spike-train rows, electrical
    and/or chemical synapses (AMPA + NMDA receptors), STDP, a sparse handle, shard handles -- against the C oracle,
    which steps the same description as a stack program (oracle/snn_oracle.c::step_custom)."""
    import torch
    from snn_amd import parallel
    model, lib = libs["DslIzhikevich"]
    electrical = variant != "chemical_only"
    chemical = variant in ("both_synapses", "chemical_only", "sharded")
    lay = parity.Layout([(0, 6, 7), (2, 5, 5)], [(5, 3, 4)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_POISSON, electrical=electrical, chemical=chemical)
    modelgen_ref.attach(net, model)
    net.custom_lib = lib
    n, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(8)
    net["current_voltage"] = ob.uniform_array(8, n, -65.0, 30.0)
    net["custom_vars"][0] = ob.uniform_array(9, n, 0.01, 0.05)                 # a, heterogeneous
    net["nt_flags"][...] = rng.random((n, 3)) < 0.7
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, :2] = 1
    net["rc_g"][:, 0] = 2.0
    net["st_nt_flags"][:, 0] = 1
    net["st_chance_of_firing"] = ob.uniform_array(10, nc, 0.0, 0.05)
    net["st_seed"] = np.arange(70, 70 + nc, dtype=np.uint32)
    net.fill_graph(11, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    steps = 600
    if variant == "sharded":
        handles = [parity.device_from_oracle(snn, net, shard=(r, 2)) for r in range(2)]
        bufs = [parallel.exchange_tensor(h, torch.device("cuda", 0)) for h in handles]
        block = bufs[0].numel() // 2
        for _ in range(steps):
            for h in handles:
                h.step_begin()
            for r in range(2):
                bufs[1 - r][r * block:(r + 1) * block].copy_(bufs[r][r * block:(r + 1) * block])
            torch.cuda.synchronize()
            for h in handles:
                h.step_end()
        net.run(steps, spike_history=True)
        for h in handles:
            st = parity.pull_state(h, net)
            b, e = h.post_begin, h.post_end
            for name in ("current_voltage", "is_spiking", "last_firing_time", "nt_t"):
                assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), name
            assert np.array_equal(parity.bits(st["custom_vars"][:, b:e]), parity.bits(net["custom_vars"][:, b:e]))
            w, c = h.get_graph_rows(0, net.n_tot)
            ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
            assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
            h.close()
        assert net.spike_history.sum() > 20
        return
    dn = parity.device_from_oracle(snn, net, csr=(variant == "sparse"))
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    w0 = net["weights"].copy()
    net.run(steps, voltage_history=True, spike_history=True)
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert net.spike_history.sum() > 20 and not np.array_equal(w0, net["weights"])
    dn.close()


def test_default_library_has_no_generated_model(snn):
    assert snn._lib.load().snn_custom_model() == b""
    with pytest.raises(snn.SnnError):
        snn.DeviceNetwork(model=snn.CUSTOM)

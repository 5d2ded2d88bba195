"""Generated neuron models on the device (SNN_MODEL_CUSTOM): the description is parsed, turned into HIP, compiled into
its own library (`_lib.build_custom`) and stepped through the same C ABI.  Checked against (a) the hand expansion the
reference's own test compares its generated code with (build_test/nb_macro/tests/basic_lif.rs:22-50 with
tests/lif_reference.rs) and (b) a numpy float32 interpreter of the description inside the canonical lattice step."""
import os

import numpy as np
import pytest

import numpy_ref as nr
import oracle_binding as ob
import modelgen_ref
import parity
from test_modelgen import (BOOL_DSL, CURRENTS, EXPECTED_FLAG, IF_DSL, IZH_DSL, LIF_NB, bool_expected_out,
                           lif_reference_trace)
from test_modelgen_channels import CALCIUM_CLAMP, HODGKIN_HUXLEY, MORRIS_LECAR, VOLTAGES, calcium_reference
from test_modelgen_receptors import IONOTROPIC_LIKE, LIF, MIXED, STEP_NEURON
from test_modelgen_spike_trains import BURST_DSL, RATE_DSL, REFRACTORINESS_DSL, _mixed_network
from test_modelgen_kinetics import (APPROXIMATE_NT, BOUNDED_RC, DESTEXHE_PAIR, ELECTROCHEMICAL_REF, RESTATED_STEP,
                                    built_in_approximate, chemical_network, custom_chemical_network,
                                    generated_approximate)

FUNCTIONS_DSL = """
[neuron]
    type: FunctionSampler
    vars: v_th = 50000000, f_exp = 0, f_tanh = 0, f_sinh = 0, f_cosh = 0, f_sin = 0, f_cos = 0, f_tan = 0, f_min = 0, f_max = 0, f_heaviside = 0, f_cube = 0, f_inverse_square = 0, f_minus_square = 0, f_nan = false
    spike_detection: v >= v_th
    on_iteration:
        f_exp = exp(i)
        f_tanh = tanh(i)
        f_sinh = sinh(i)
        f_cosh = cosh(i)
        f_sin = sin(i)
        f_cos = cos(i)
        f_tan = tan(i)
        f_nan = isnan(i - i) || isnan(f_exp - f_exp)
        f_min = min(0.5, i)
        f_max = max(0.5, i)
        f_heaviside = heaviside(i)
        f_cube = i ^ 3
        f_inverse_square = i ^ -2
        f_minus_square = -i ^ 2
[end]"""          # the functions of build_test/nb_macro/tests/function_usage.rs in one model

import random_descriptions

pytestmark = pytest.mark.gpu
RANDOM_DSL = [random_descriptions.description(seed, count=28, name=f"RandomExpressions{seed}") for seed in (7, 8)]
f32 = np.float32


@pytest.fixture(scope="module")
def libs(snn):
    from snn_amd import _lib, modelgen
    from concurrent.futures import ThreadPoolExecutor
    models = [modelgen.parse(text) for text in (LIF_NB, IZH_DSL, IF_DSL, CALCIUM_CLAMP, MORRIS_LECAR, FUNCTIONS_DSL,
                                                 BOOL_DSL, ELECTROCHEMICAL_REF, RESTATED_STEP, HODGKIN_HUXLEY, *RANDOM_DSL)]
    models += [modelgen.parse_description(text) for text in (
        RATE_DSL + REFRACTORINESS_DSL, APPROXIMATE_NT + BOUNDED_RC, IZH_DSL + BURST_DSL + DESTEXHE_PAIR,
        MIXED + LIF.format(name="MixedIntegrateAndFire", receptors="MixedReceptors"),
        IONOTROPIC_LIKE + STEP_NEURON.format(name="OwnReceptors", receptors="receptors: AmpaGabaReceptors\n    "))]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:          # one hipcc each
        paths = list(pool.map(_lib.build_custom, models))
    return {m.name: (m, path) for m, path in zip(models, paths)}


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


def test_bool_variables_known_answers(snn, libs):
    """build_test/nb_macro/tests/bool_vars.rs on the device: `out` is 1 / 2 by the bool `flag` for every input current,
    a bool written by on_spike is read back through `!`, `&&`, `==`, and the voltages stay the plain LIF's."""
    model, lib = libs["BoolIntegrateAndFire"]
    currents = np.concatenate([CURRENTS, CURRENTS])
    flag = np.concatenate([np.zeros(CURRENTS.size, bool), np.ones(CURRENTS.size, bool)])
    n = currents.size
    dn = snn.DeviceNetwork(model=snn.CUSTOM, spike_train=snn.ST_RATE, lib_path=lib)
    dn.add_lattice(1, 1, n)
    dn.add_spike_train_lattice(0, 1, n)
    dn.finalize()
    assert np.array_equal(dn.get_attr(1, "flag"), np.zeros(n, f32))                 # false
    dn.set_attr(1, "flag", flag.astype(f32))
    dn.set_attr(0, "v_resting", currents)
    w = np.zeros((2 * n, n), f32)
    c = np.zeros((2 * n, n), np.uint32)
    w[n + np.arange(n), np.arange(n)] = 1.0
    c[n + np.arange(n), np.arange(n)] = 1
    dn.set_graph_rows(0, w, c)
    dn.set_history(voltage=True, spikes=False)
    dn.run(1)
    assert np.array_equal(dn.get_attr(1, "out"), bool_expected_out(flag, 1))
    dn.run(299)
    assert np.array_equal(dn.get_attr(1, "out"), bool_expected_out(flag, 300))
    assert np.array_equal(dn.get_attr(1, "seen"), np.ones(n, f32))
    ref = np.concatenate([lif_reference_trace(300)] * 2, axis=1)
    assert np.array_equal(parity.bits(dn.voltage_history(1)), parity.bits(ref))
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
        ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
        for _ in range(steps):
            ex.step()
        net.run(steps, spike_history=True)
        for h in handles:
            st = parity.pull_state(h, net)
            b, e = h.post_begin, h.post_end
            parity.assert_shard_view_equal(h, st, net)
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


def test_gated_ion_channel_known_answer(snn, libs):
    """timestep_dependent_ion_channel.rs on the device: the calcium channel (one gating variable, exp, `^ 2`) inside a
    neuron that holds its voltage, 1000 iterations at each of the test's voltages with the channel state carried
    over -- current and gate equal to the float32 hand expansion of ReferenceCalciumIonChannel."""
    model, lib = libs["VoltageClamp"]
    want, (alpha, beta, state) = calcium_reference(VOLTAGES, 1000, 0.01)
    dn = snn.DeviceNetwork(model=snn.CUSTOM, lib_path=lib)
    dn.add_lattice(0, 1, 1)
    dn.finalize()
    assert dn.get_attr(0, "dt")[0] == f32(0.01) and dn.get_attr(0, "ca$g")[0] == f32(0.025)
    for k, v in enumerate(VOLTAGES):
        dn.set_attr(0, "current_voltage", np.full(1, v, f32))
        dn.run(999)
        assert parity.bits(dn.get_attr(0, "ca$current"))[0] == parity.bits(want[1000 * k + 998:1000 * k + 999])[0]
        dn.run(1)
        assert parity.bits(dn.get_attr(0, "ca$current"))[0] == parity.bits(want[1000 * (k + 1) - 1:1000 * (k + 1)])[0]
        assert dn.get_attr(0, "current_voltage")[0] == f32(v)
    assert (dn.get_attr(0, "ca$s$alpha")[0], dn.get_attr(0, "ca$s$beta")[0], dn.get_attr(0, "ca$s$state")[0]) == \
        (alpha, beta, state)
    dn.close()


def test_functions_and_powers_equal_the_oracle(snn, libs):
    """exp / tanh / sinh / cosh / min / max / heaviside / integer powers: one model that stores f(i) for each, fed a
    spread of inputs through never-firing rate cells -- bit-identical to the C oracle's own implementations, and
    within 1 ULP of the exact value."""
    model, lib = libs["FunctionSampler"]
    x = np.concatenate([np.array([-88.0, -30.0, -9.5, -3.0, -1.0, -0.3, -0.05, -0.01, -1e-6, 0.0, 1e-6, 0.02, 0.05, 0.5,
                                  1.0, 2.5, 7.25, 11.0, 19.9, 20.1, 45.0, 88.0, 91.0], f32),
                        ob.uniform_array(12, 233, -12.0, 12.0)])
    n = x.size
    lay = parity.Layout([(1, 1, n)], [(0, 1, n)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_RATE)
    modelgen_ref.attach(net, model)
    net.custom_lib = lib
    net["st_v_resting"] = x
    net["connections"][n + np.arange(n), np.arange(n)] = 1
    net["weights"][n + np.arange(n), np.arange(n)] = 1.0
    dn = parity.device_from_oracle(snn, net)
    dn.run(2)
    net.run(2)
    names = [name for name, _ in model.variables]
    exact = {"f_exp": np.exp, "f_tanh": np.tanh, "f_sinh": np.sinh, "f_cosh": np.cosh, "f_sin": np.sin, "f_cos": np.cos,
             "f_tan": np.tan}
    with np.errstate(all="ignore"):
        for k, name in enumerate(names):
            got = dn.get_attr(1, name)
            assert np.array_equal(parity.bits(got), parity.bits(net["custom_vars"][k])), name
            if name in exact:
                want = exact[name](x.astype(np.float64)).astype(f32)
                ok = np.isfinite(want) & (np.abs(want) > 1e-37)
                ulp = np.abs(got[ok].view(np.int32).astype(np.int64) - want[ok].view(np.int32).astype(np.int64))
                assert ulp.max() <= 1, (name, ulp.max())
                assert np.array_equal(np.isinf(got), np.isinf(want)), name
    assert np.array_equal(dn.get_attr(1, "f_minus_square"), -(x * x))
    assert np.array_equal(dn.get_attr(1, "f_nan") != 0, x > f32(88.73))            # exp overflows: inf - inf
    dn.close()


def test_hodgkin_huxley_in_the_dsl_equals_the_built_in_neuron(snn, libs):
    """HodgkinHuxleyNeuron written in the DSL (three ion channels with gating variables, `^ 3` / `^ 4`,
    continuous() spike detection) in a gap-junction lattice: the generated library against the oracle's stack program
    AND against the default library's built-in Hodgkin-Huxley model -- same voltages, same raster, same gates."""
    model, lib = libs["DslHodgkinHuxley"]
    names = [n for n, _ in model.variables]
    v0 = ob.uniform_array(90, 12, -70.0, -40.0)
    gates = {k: ob.uniform_array(91 + j, 12, 0.05, 0.6) for j, k in enumerate(("m", "h", "n"))}
    nets = []
    for generated in (False, True):
        net = parity.make_oracle(parity.Layout([(0, 3, 4)]), model=ob.CUSTOM if generated else ob.HH)
        if generated:
            modelgen_ref.attach(net, model)
            net.custom_lib = lib
        net["current_voltage"] = v0
        net["gap_conductance"] = 0.5
        for k, arr in gates.items():
            if generated:
                net["custom_vars"][names.index(f"{'k_channel' if k == 'n' else 'na_channel'}${k}$state")] = arr
            else:
                net[f"{k}_state"] = arr
        net.fill_graph(94, 0.5, 1.5)
        nets.append(net)
    steps = 6000
    devices = [parity.device_from_oracle(snn, net) for net in nets]
    for dn in devices:
        dn.set_history(voltage=True, spikes=True)
        dn.run(steps)
    nets[1].run(steps, voltage_history=True, spike_history=True)
    built_in, gen = devices
    assert np.array_equal(gen.spike_history(0), nets[1].spike_history) and nets[1].spike_history.sum() > 5
    assert np.array_equal(parity.bits(gen.voltage_history(0)), parity.bits(nets[1].voltage_history))
    assert np.array_equal(gen.spike_history(0), built_in.spike_history(0))
    assert np.array_equal(parity.bits(gen.voltage_history(0)), parity.bits(built_in.voltage_history(0)))
    for attr in ("na_channel$m$state", "na_channel$h$state", "k_channel$n$state", "na_channel$current", "k_channel$current"):
        assert np.array_equal(parity.bits(gen.get_attr(0, attr)), parity.bits(built_in.get_attr(0, attr))), attr
    parity.assert_state_equal(nets[1], parity.pull_state(gen, nets[1]))
    for dn in devices:
        dn.close()


@pytest.mark.parametrize("seed", [7, 8])
def test_random_descriptions_equal_the_oracle(snn, libs, seed):
    """Property test of the HIP emitter: 28 random statements (every operator and function, powers, if / else with
    && || ! isnan, earlier results as operands) compiled into a model, 256 neurons with different inputs and
    parameters -- every variable bit-identical to the C oracle's stack program (which the numpy interpreter checks on
    the CPU side, test_modelgen.py)."""
    model, lib = libs[f"RandomExpressions{seed}"]
    n = 256
    x = ob.uniform_array(100 + seed, n, -4.0, 4.0)
    lay = parity.Layout([(1, 1, n)], [(0, 1, n)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_RATE)
    modelgen_ref.attach(net, model)
    net.custom_lib = lib
    names = [name for name, _ in model.variables]
    for k, (name, lo, hi) in enumerate((("a", -2.0, 2.0), ("b", -2.0, 2.0), ("c", 0.1, 3.0))):
        net["custom_vars"][names.index(name)] = ob.uniform_array(200 + 100 * k + seed, n, lo, hi)
    net["st_v_resting"] = x
    net["connections"][n + np.arange(n), np.arange(n)] = 1
    net["weights"][n + np.arange(n), np.arange(n)] = 1.0
    dn = parity.device_from_oracle(snn, net)
    dn.run(3)
    net.run(3)
    finite = 0
    for k, name in enumerate(names):
        got = dn.get_attr(1, name)
        assert np.array_equal(parity.bits(got), parity.bits(net["custom_vars"][k])), (seed, name)
        finite += int(np.isfinite(got).sum())
    assert finite > 0.7 * n * len(names)
    dn.close()


@pytest.mark.parametrize("variant", ["electrical", "both_synapses", "sparse"])
def test_morris_lecar_network_equals_the_oracle(snn, libs, variant):
    """A Morris-Lecar neuron assembled from three DSL ion channels (tanh, cosh, a channel-local differential equation,
    19 generated variables) in two lattices with Poisson rows, gap junctions and AMPA/NMDA synapses, STDP on:
    raster, voltages, every channel variable and the weights bit-identical to the C oracle."""
    model, lib = libs["MorrisLecarNeuron"]
    chemical = variant == "both_synapses"
    lay = parity.Layout([(0, 6, 6), (2, 4, 5)], [(5, 3, 3)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_POISSON, electrical=True, chemical=chemical)
    modelgen_ref.attach(net, model)
    net.custom_lib = lib
    n, nc = net.n_neurons, net.n_cells
    names = [name for name, _ in model.variables]
    rng = np.random.default_rng(21)
    net["current_voltage"] = ob.uniform_array(21, n, -70.0, -20.0)
    net["gap_conductance"] = 1.0                      # explicit Euler at dt = 0.1: stronger coupling diverges
    net["custom_vars"][names.index("k_channel$phi")] = ob.uniform_array(22, n, 0.04, 0.09)      # heterogeneous
    net["nt_flags"][:, 0] = 1
    net["nt_flags"][:, 1] = rng.random(n) < 0.5
    net["rc_flags"][:, :2] = 1
    net["rc_g"][:, 0] = 1.5
    net["st_nt_flags"][:, 0] = 1
    net["st_chance_of_firing"] = ob.uniform_array(23, nc, 0.0, 0.03)
    net["st_seed"] = np.arange(170, 170 + nc, dtype=np.uint32)
    net.fill_graph(24, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.4] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    steps = 2400
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
    vh = net.voltage_history
    assert np.isfinite(vh).all() and vh.max() > 0.0 and net.spike_history.sum() > 10
    assert not np.array_equal(w0, net["weights"])
    dn.close()


def test_generated_rate_train_and_refractoriness_on_the_device(snn, libs):
    """The reference's RateSpikeTrain and delta-Dirac refractoriness descriptions (rate_spike_train.rs,
    delta_dirac_refractoriness.rs) compiled into a library: a two-lattice Izhikevich network driven by them is
    bit-identical to the C oracle stepping the same descriptions -- and therefore (test_modelgen_spike_trains.py) to the
    built-in pair they restate, which the default library runs here for a direct comparison."""
    desc, lib = libs["RateSpikeTrain_TestRefractoriness"]
    L = snn._lib.load(lib)
    assert (L.snn_custom_model(), L.snn_custom_spike_train(), L.snn_custom_refractoriness()) == \
        (b"", b"RateSpikeTrain", b"TestRefractoriness")
    rates = np.array([3.1, 0.0, 7.7, 12.0, 5.0, 1.3, 40.0, 9.9], f32)
    decay = ob.uniform_array(33, 8, 50.0, 4000.0)
    net = _mixed_network(ob, parity, ob.ST_CUSTOM)
    modelgen_ref.attach_spike_train(net, desc.spike_train)
    modelgen_ref.attach_refractoriness(net, desc.refractoriness)
    net.custom_lib = lib
    net["st_custom_vars"][1] = rates
    net["st_k"] = decay
    built_in = _mixed_network(ob, parity, ob.ST_RATE)
    built_in["st_rate"] = rates
    built_in["st_k"] = decay
    steps = 700
    dn = parity.device_from_oracle(snn, net)
    assert np.array_equal(dn.get_attr(5, "neural_refractoriness$decay"), decay)       # alias of neural_refractoriness$k
    dn0 = parity.device_from_oracle(snn, built_in)                                    # default library, built-in pair
    for d in (dn, dn0):
        d.set_history(voltage=True, spikes=True)
        d.run(steps // 2)
        d.run(steps - steps // 2)
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=True)
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(dn0.voltage_history(i)))
    assert np.array_equal(parity.bits(dn.voltage_history(5)), parity.bits(net.st_voltage_history))
    assert np.array_equal(parity.bits(dn.voltage_history(5)), parity.bits(dn0.voltage_history(5)))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert np.array_equal(parity.bits(dn.get_attr(5, "step")), parity.bits(dn0.get_attr(5, "step")))
    assert (net.st_voltage_history == f32(30.0)).sum() > 100 and net.spike_history.sum() > 20
    with pytest.raises(snn.SnnError):
        dn.set_attr(5, "neural_refractoriness$kind", np.full(8, 3, np.uint32))
    dn.close()
    dn0.close()


def test_generated_kinetics_on_the_device(snn, libs):
    """Transmitter and receptor kinetics from descriptions (the reference's receptor_kinetics.rs block and the
    Approximate transmitter kinetics written with the built-in's association) in a two-lattice network with Poisson
    rows, AMPA / NMDA / GABA, STDP: bit-identical to the C oracle stepping the same descriptions, and to the default
    library's built-in Approximate kinetics they restate."""
    desc, lib = libs["ApproximateKinetics_BoundedReceptorKinetics"]
    _, net = generated_approximate(ob, parity, modelgen_ref)
    net.custom_lib = lib
    built_in = built_in_approximate(ob, parity)
    steps = 600
    dn = parity.device_from_oracle(snn, net)
    dn0 = parity.device_from_oracle(snn, built_in)
    for d in (dn, dn0):
        d.set_history(voltage=True, spikes=True)
        d.run(steps // 2)
        d.run(steps - steps // 2)
    net.run(steps, voltage_history=True, spike_history=True)
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(dn0.voltage_history(i)))
        assert np.array_equal(parity.bits(dn.get_attr(i, "neurotransmitters$t", per_type=True)),
                              parity.bits(dn0.get_attr(i, "neurotransmitters$t", per_type=True)))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    assert net.spike_history.sum() > 20 and net["rc_r"].max() > 0.01 and net["nt_t"].max() > 0.1
    dn.close()
    dn0.close()


@pytest.mark.parametrize("which", ["ElectroChemicalIntegrateAndFire", "RestatedStep"])
@pytest.mark.parametrize("variant", ["dense", "sparse", "electrical_only"])
def test_on_electrochemical_iteration_equals_the_oracle(snn, libs, which, variant):
    """Neurons with their own chemical step (the reference's gpu_custom_electrochemical.rs model; the default sequence
    written out by hand) in a network with Poisson rows and AMPA / NMDA / GABA synapses: raster, voltages, receptor and
    transmitter state bit-identical to the C oracle; with transmission off the plain on_iteration runs."""
    model, lib = libs[which]
    text = ELECTROCHEMICAL_REF if which == "ElectroChemicalIntegrateAndFire" else RESTATED_STEP
    _, net = custom_chemical_network(ob, parity, modelgen_ref, text, chemical=(variant != "electrical_only"))
    net.custom_lib = lib
    steps = 400
    dn = parity.device_from_oracle(snn, net, csr=(variant == "sparse"))
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    net.run(steps, voltage_history=True, spike_history=True)
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    assert np.isfinite(net.voltage_history).all()
    if variant != "electrical_only":
        assert net["rc_r"].max() > 0.01 and np.abs(net["rc_current"]).max() > 0.0
    dn.close()


@pytest.mark.parametrize("which", ["OwnReceptors_AmpaGabaReceptors", "MixedIntegrateAndFire_MixedReceptors"])
@pytest.mark.parametrize("variant", ["dense", "sparse"])
def test_generated_receptor_sets_equal_the_oracle(snn, libs, which, variant):
    """Generated neurons with their own [receptors] set -- one that spells out the AMPA / GABA currents (bit-identical
    to the built-in ionotropic receptors on the oracle, test_modelgen_receptors.py) and shared_receptors.rs's
    ionotropic + metabotropic pair -- in a network with Poisson rows: raster, voltages, receptor states and the set's
    variables bit-identical to the C oracle."""
    desc, lib = libs[which]
    assert snn._lib.load(lib).snn_custom_receptors() == desc.receptors.name.encode()
    net = chemical_network(ob, parity, ob.NT_APPROX, ob.RC_APPROX, model=ob.CUSTOM)
    modelgen_ref.attach(net, desc.neuron)
    modelgen_ref.attach_receptors(net, desc.receptors)
    net.custom_lib = lib
    names = [n for n, _ in desc.receptors.variables]
    if which.startswith("Own"):
        net["current_voltage"] = ob.uniform_array(80, net.n_neurons, -68.0, -52.0)
        net["rx_vars"][names.index("AMPA$g")] = ob.uniform_array(81, net.n_neurons, 0.5, 2.0)
        net["rc_flags"][:, 1] = 0
    else:
        net["current_voltage"] = ob.uniform_array(82, net.n_neurons, -80.0, -56.0)
        net["custom_vars"][0] = -60.0                                               # e: a stable leak would need -(v - e)
        net["rx_vars"][names.index("Meta$s")] = ob.uniform_array(83, net.n_neurons, 0.5, 1.5)
        net["rc_flags"][:, 2] = 0
    net["do_plasticity"] = 0
    steps = 300
    dn = parity.device_from_oracle(snn, net, csr=(variant == "sparse"))
    assert np.array_equal(dn.get_attr(0, "receptors$" + names[1]), net["rx_vars"][1][:25])
    dn.set_history(voltage=True, spikes=True)
    dn.run(steps // 2)
    dn.run(steps - steps // 2)
    net.run(steps, voltage_history=True, spike_history=True)
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net), skip=("rc_current",))       # the set keeps its own currents
    assert net["rc_r"].max() > 0.01 and np.abs(net["rx_vars"][names.index(names[1])]).max() > 0.0
    dn.close()


@pytest.mark.parametrize("variant", ["dense", "sparse", "sharded"])
def test_whole_description_in_one_library_equals_the_oracle(snn, libs, variant):
    """One library carrying all five generated blocks -- the DSL Izhikevich neuron, a bursting spike train
    (differential equation, exp, its own bool), a refractoriness with an extra variable, Destexhe transmitter and
    receptor kinetics written in the DSL -- through a network with gap junctions, AMPA / NMDA synapses and STDP, on
    dense, sparse and shard handles, against the C oracle."""
    import torch
    from snn_amd import parallel
    desc, lib = libs["DslIzhikevich_BurstSpikeTrain_PlateauRefractoriness_DslDestexheNeurotransmitter_DslDestexheReceptor"]
    L = snn._lib.load(lib)
    assert (L.snn_custom_neurotransmitter_kinetics(), L.snn_custom_receptor_kinetics()) == \
        (b"DslDestexheNeurotransmitter", b"DslDestexheReceptor")
    lay = parity.Layout([(0, 6, 7), (2, 5, 5)], [(5, 3, 4)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_CUSTOM, nt_kind=ob.NT_CUSTOM, rc_kind=ob.RC_CUSTOM,
                             electrical=True, chemical=True)
    modelgen_ref.attach(net, desc.neuron)
    modelgen_ref.attach_spike_train(net, desc.spike_train)
    modelgen_ref.attach_refractoriness(net, desc.refractoriness)
    modelgen_ref.attach_nt_kinetics(net, desc.nt_kinetics)
    modelgen_ref.attach_receptor_kinetics(net, desc.receptor_kinetics)
    net["nt_custom_vars"][2] = ob.uniform_array(47, net.n_neurons * 3, 3.0, 8.0).reshape(-1, 3)         # k_p
    net["st_nt_custom_vars"][0] = ob.uniform_array(48, net.n_cells * 3, 0.5, 1.0).reshape(-1, 3)         # t_max
    net["rc_custom_vars"][1] = ob.uniform_array(49, net.n_neurons * 3, 0.5, 2.0).reshape(-1, 3)         # beta
    net.custom_lib = lib
    n, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(41)
    st_names = [name for name, _ in desc.spike_train.variables]
    net["current_voltage"] = ob.uniform_array(41, n, -65.0, 30.0)
    net["custom_vars"][0] = ob.uniform_array(42, n, 0.01, 0.05)
    net["st_custom_vars"][st_names.index("freq")] = ob.uniform_array(43, nc, 0.01, 0.06)
    net["refr_vars"][0] = ob.uniform_array(44, nc, 0.0, 8.0)                    # plateau
    net["st_k"] = ob.uniform_array(45, nc, 200.0, 4000.0)
    net["nt_flags"][:, 0] = 1
    net["nt_flags"][:, 1] = rng.random(n) < 0.6
    net["rc_flags"][:, :2] = 1
    net["rc_g"][:, 0] = 2.0
    net["st_nt_flags"][:, 0] = 1
    net.fill_graph(46, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    steps = 900
    if variant == "sharded":
        handles = [parity.device_from_oracle(snn, net, shard=(r, 2)) for r in range(2)]
        ex = parallel.LocalExchange(handles, torch.device("cuda", 0))
        for _ in range(steps):
            ex.step()
        net.run(steps, spike_history=True)
        for h in handles:
            st = parity.pull_state(h, net)
            b, e = h.post_begin, h.post_end
            parity.assert_shard_view_equal(h, st, net)
            for name in ("st_current_voltage", "st_last_firing_time", "st_custom_vars", "st_nt_t", "st_nt_custom_vars"):
                assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), name
            assert np.array_equal(parity.bits(st["custom_vars"][:, b:e]), parity.bits(net["custom_vars"][:, b:e]))
            for name in ("rc_r", "rc_current"):
                assert np.array_equal(parity.bits(st[name][b:e]), parity.bits(net[name][b:e])), name
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
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=True)
    assert net["rc_r"].max() > 0.01 and net["nt_t"].max() > 0.1
    ranges = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = ranges[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    assert np.array_equal(parity.bits(dn.voltage_history(5)), parity.bits(net.st_voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    fired = net.st_voltage_history == f32(25.0)
    assert fired.sum() > 50 and net.spike_history.sum() > 20 and not np.array_equal(w0, net["weights"])
    dn.close()


def test_default_library_has_no_generated_model(snn):
    L = snn._lib.load()
    assert (L.snn_custom_model(), L.snn_custom_spike_train(), L.snn_custom_refractoriness(),
            L.snn_custom_neurotransmitter_kinetics(), L.snn_custom_receptor_kinetics(), L.snn_custom_receptors()) == \
        (b"",) * 6
    with pytest.raises(snn.SnnError):
        snn.DeviceNetwork(model=snn.CUSTOM)
    with pytest.raises(snn.SnnError):
        snn.DeviceNetwork(model=snn.IZHIKEVICH, spike_train=snn.ST_CUSTOM)
    with pytest.raises(snn.SnnError):
        snn.DeviceNetwork(model=snn.IZHIKEVICH, nt_kinetics=snn.NT_CUSTOM)
    with pytest.raises(snn.SnnError):
        snn.DeviceNetwork(model=snn.IZHIKEVICH, receptor_kinetics=snn.RC_CUSTOM)
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH, spike_train=snn.ST_RATE)
    dn.add_lattice(0, 2, 2)
    dn.add_spike_train_lattice(1, 1, 3)
    dn.finalize()
    dn.set_attr(1, "neural_refractoriness$kind", np.ones(3, np.uint32))
    with pytest.raises(snn.SnnError):
        dn.set_attr(1, "neural_refractoriness$kind", np.full(3, 2, np.uint32))          # needs a generated refractoriness
    dn.close()

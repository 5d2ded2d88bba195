"""[spike_train] and [neural_refractoriness] blocks of the description generator (CPU side): parsing, the numpy
interpreter and the C oracle's stack programs against the reference's own tests
(/root/reference/build_test/nb_macro/tests/{rate_spike_train,delta_dirac_refractoriness}.rs) and against the built-in
RateSpikeTrain / DeltaDiracRefractoriness, which the two reference descriptions restate."""
import numpy as np
import pytest

from snn_amd import modelgen
from snn_amd.examples_dsl import BURST_DSL  # noqa: E402,F401

f32 = np.float32

RATE_DSL = """
[spike_train]
    type: RateSpikeTrain
    vars: step = 0., rate = 0.
    on_iteration:
        step += dt
        [if] rate != 0. && step >= rate [then]
            step = 0
            current_voltage = v_th
            is_spiking = true
        [else]
            current_voltage = v_resting
            is_spiking = false
        [end]
[end]"""          # build_test/nb_macro/tests/rate_spike_train.rs:8-21, restated as data

REFRACTORINESS_DSL = """
[neural_refractoriness]
    type: TestRefractoriness
    effect: (v_th - v_resting) * exp((-1 / (decay / dt)) * (time_difference ^ 2)) + v_resting
[end]"""          # build_test/nb_macro/tests/delta_dirac_refractoriness.rs:9-12

# a train the reference has no built-in for: a phase oscillator (differential equation, exp, a bool of its own) whose
# spikes come in bursts, with a refractoriness that has a variable besides `decay`


def _st_state(model, n):
    st = {"current_voltage": np.full(n, model.mandatory["current_voltage"], f32), "is_spiking": np.zeros(n, f32),
          "dt": np.full(n, model.mandatory["dt"], f32), "v_resting": np.full(n, model.mandatory["v_resting"], f32),
          "v_th": np.full(n, model.mandatory["v_th"], f32)}
    for name, default in model.variables:
        st[name] = np.full(n, default, f32)
    return st


def test_blocks_are_parsed_and_emitted():
    d = modelgen.parse_description(RATE_DSL + REFRACTORINESS_DSL)
    assert d.neuron is None and d.name == "RateSpikeTrain_TestRefractoriness"
    st, rf = d.spike_train, d.refractoriness
    assert st.variables == [("step", 0.0), ("rate", 0.0)] and st.bools == {"is_spiking"}
    assert st.mandatory == {"dt": 0.1, "v_resting": 0.0, "v_th": 30.0, "current_voltage": 0.0}      # lib.rs:4845-4848
    assert rf.variables == [] and rf.decay == 10000.0                                               # lib.rs:5685-5706
    src = modelgen.hip_source(d)
    assert "namespace custom_st {" in src and "namespace custom_refr {" in src and "namespace custom {" not in src
    assert "is_spiking = true;" in src and "v = v_th;" in src
    assert "powif_glibc(time_difference, 2)" in src
    b = modelgen.parse_description(BURST_DSL)
    assert b.spike_train.mandatory["v_th"] == 25.0 and b.spike_train.bools == {"bursting", "is_spiking"}
    assert b.refractoriness.variables == [("plateau", 3.0)] and b.refractoriness.decay == 2000.0
    with pytest.raises(modelgen.ModelError):
        modelgen.parse(RATE_DSL)                       # parse() is the neuron-only entry point


def test_rate_spike_train_reference_tests_on_both_cpu_evaluators():
    """rate_spike_train.rs: test_expected_rate (10 000 iterations, |spikes - iterations / (rate / dt)| <= 1, none at
    rate 0) and test_spacing (rate 100, dt 1: a spike exactly when (i + 1) % 100 == 0)."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    m = modelgen.parse_description(RATE_DSL).spike_train
    rates = np.array([0, 100, 200, 300, 400, 500], f32)
    iterate = modelgen_ref.make_spike_train_step(m)
    st = _st_state(m, rates.size)
    st["rate"] = rates.copy()
    spikes = np.zeros(rates.size, np.int64)
    for _ in range(10_000):
        fired = iterate(st)
        spikes += fired
        assert np.array_equal(st["current_voltage"], np.where(fired, f32(30.0), f32(0.0)))
    assert spikes[0] == 0
    assert (np.abs(spikes[1:] - 10_000 / (rates[1:] / 0.1)) <= 1).all(), spikes
    st = _st_state(m, 1)
    st["rate"][:] = 100.0
    st["dt"][:] = 1.0
    for i in range(1001):
        assert bool(iterate(st)[0]) == (i != 0 and (i + 1) % 100 == 0), i
    # C oracle: the same six cells (a spike shows as v_th in the cell voltage history)
    net = parity.make_oracle(parity.Layout([(1, 1, 1)], [(0, 1, rates.size)]), st_kind=ob.ST_CUSTOM)
    modelgen_ref.attach_spike_train(net, m)
    net["st_custom_vars"][1] = rates
    net.run(10_000, st_voltage_history=True)
    assert np.array_equal((net.st_voltage_history == f32(30.0)).sum(axis=0), spikes)
    assert np.array_equal(net["st_custom_vars"][0].view(np.uint32), _final_steps(m, rates).view(np.uint32))


def _final_steps(m, rates):
    import modelgen_ref
    iterate = modelgen_ref.make_spike_train_step(m)
    st = _st_state(m, rates.size)
    st["rate"] = rates.copy()
    for _ in range(10_000):
        iterate(st)
    return st["step"]


def _mixed_network(ob, parity, st_kind, seed=31):
    lay = parity.Layout([(0, 5, 5), (2, 3, 4)], [(5, 2, 4)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH, st_kind=st_kind, electrical=True, chemical=True)
    n, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(seed)
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, :2] = 1
    net["st_nt_flags"][:, 0] = 1
    net["st_nt_flags"][:, 1] = rng.random(nc) < 0.5
    net.fill_graph(seed + 1, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    return net


def test_generated_rate_train_and_refractoriness_equal_the_built_in_ones():
    """The two reference descriptions restate RateSpikeTrain::iterate (spike_train/mod.rs:1016-1031) and
    DeltaDiracRefractoriness::get_effect (:79-88): a network driven by the generated pair is bit-identical to the same
    network driven by the built-in pair -- cell voltages, neuron voltages, raster, weights."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    d = modelgen.parse_description(RATE_DSL + REFRACTORINESS_DSL)
    rates = np.array([3.1, 0.0, 7.7, 12.0, 5.0, 1.3, 40.0, 9.9], f32)
    decay = ob.uniform_array(33, 8, 50.0, 4000.0)
    built_in = _mixed_network(ob, parity, ob.ST_RATE)
    built_in["st_rate"] = rates
    built_in["st_k"] = decay
    generated = _mixed_network(ob, parity, ob.ST_CUSTOM)
    modelgen_ref.attach_spike_train(generated, d.spike_train)
    modelgen_ref.attach_refractoriness(generated, d.refractoriness)
    generated["st_custom_vars"][1] = rates
    generated["st_k"] = decay
    for net in (built_in, generated):
        net.run(700, voltage_history=True, spike_history=True, st_voltage_history=True)
    assert (built_in.st_voltage_history == f32(30.0)).sum() > 100 and built_in.spike_history.sum() > 20
    assert np.array_equal(generated.st_voltage_history.view(np.uint32), built_in.st_voltage_history.view(np.uint32))
    assert np.array_equal(generated.voltage_history.view(np.uint32), built_in.voltage_history.view(np.uint32))
    assert np.array_equal(generated.spike_history, built_in.spike_history)
    assert np.array_equal(generated["weights"].view(np.uint32), built_in["weights"].view(np.uint32))
    assert np.array_equal(generated["st_custom_vars"][0].view(np.uint32), built_in["st_step"].view(np.uint32))
    assert np.array_equal(generated["st_nt_t"].view(np.uint32), built_in["st_nt_t"].view(np.uint32))


def test_refractoriness_effect_reference_test():
    """delta_dirac_refractoriness.rs:15-35: 50 random draws against DeltaDiracRefractoriness::get_effect (the reference
    asserts |difference| < 0.01; here the restated expression is the same arithmetic, so the bits agree) -- numpy
    interpreter and the C oracle's stack program inside the input calculation."""
    import modelgen_ref
    import oracle_binding as ob
    m = modelgen.parse_description(REFRACTORINESS_DSL).refractoriness
    rng = np.random.default_rng(35)
    for _ in range(50):
        decay = f32(rng.uniform(0.0, 20_000.0))
        last = int(rng.integers(0, 1000))
        timestep = int(rng.integers(last, last + 1000))
        v_max = f32(rng.uniform(10.0, 30.0))
        want = f32(ob.lib().snn_o_delta_dirac_effect(timestep, last, v_max, 0.0, decay, f32(0.1)))
        with np.errstate(all="ignore"):
            got = modelgen_ref.refractoriness_effect(m, timestep - last, v_max, 0.0, 0.1, decay=decay)
        assert np.array_equal(np.asarray(got, f32).view(np.uint32), np.asarray(want, f32).view(np.uint32))


def test_burst_train_cpu_evaluators_agree_and_it_bursts():
    """A train with a differential equation, exp, its own bool and `!is_spiking`, and a refractoriness with an extra
    variable and max(): the C oracle's stack programs against the numpy interpreter; bursts of alternating spikes."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    d = modelgen.parse_description(BURST_DSL)
    m = d.spike_train
    nc = 6
    freq = ob.uniform_array(36, nc, 0.01, 0.05)
    iterate = modelgen_ref.make_spike_train_step(m)
    st = _st_state(m, nc)
    st["freq"] = freq.copy()
    want_v = []
    for _ in range(3000):
        iterate(st)
        want_v.append(st["current_voltage"].copy())
    want_v = np.array(want_v)
    fired = want_v == f32(25.0)
    assert fired.sum() > 50 and (fired[1:] & fired[:-1]).sum() == 0            # never two steps in a row
    net = parity.make_oracle(parity.Layout([(1, 2, 2)], [(0, 1, nc)]), st_kind=ob.ST_CUSTOM)
    modelgen_ref.attach_spike_train(net, m)
    modelgen_ref.attach_refractoriness(net, d.refractoriness)
    names = [n for n, _ in m.variables]
    net["st_custom_vars"][names.index("freq")] = freq
    net["connections"][4:, :] = 1
    net["weights"][4:, :] = 0.5
    net.run(3000, st_voltage_history=True, voltage_history=True)
    assert np.array_equal(net.st_voltage_history.view(np.uint32), want_v.view(np.uint32))
    for k, name in enumerate(names):
        assert np.array_equal(net["st_custom_vars"][k].view(np.uint32), st[name].view(np.uint32)), name
    assert np.isfinite(net.voltage_history).all()
    # the plateau: for time_difference <= plateau the effect is exactly v_th
    rf = d.refractoriness
    assert modelgen_ref.refractoriness_effect(rf, 2.0, 25.0, -5.0, 0.1) == f32(25.0)
    assert modelgen_ref.refractoriness_effect(rf, 200.0, 25.0, -5.0, 0.1) < f32(25.0)


@pytest.mark.parametrize("text,needle", [
    (RATE_DSL.replace("is_spiking = true", "is_spiking = 1"), "assignment to is_spiking"),
    (RATE_DSL.replace("current_voltage = v_th", "v_th = 3"), "cannot assign to 'v_th'"),
    (RATE_DSL.replace("step += dt", "step += i"), "unknown variable 'i'"),
    (RATE_DSL.replace("vars: step = 0., rate = 0.", "vars: step = 0., rate = 0., is_spiking = 0"), "is a bool"),
    (RATE_DSL + RATE_DSL, "more than one [spike_train]"),
    (REFRACTORINESS_DSL.replace("time_difference ^ 2", "elapsed ^ 2"), "unknown variable 'elapsed'"),
    (REFRACTORINESS_DSL.replace("    effect:", "    vars: on = true\n    effect:"), "bool variables have no use"),
    (REFRACTORINESS_DSL.replace("+ v_resting", "+ v_resting > 0"), "effect needs a number"),
    (REFRACTORINESS_DSL.replace("    effect: (v_th", "    impact: (v_th"), "section 'effect' is missing"),
    ("[ion_channel]\n    type: L\n    on_iteration:\n        current = v\n[end]" + RATE_DSL, "without a [neuron]"),
])
def test_spike_train_errors_name_the_problem(text, needle):
    with pytest.raises(modelgen.ModelError) as e:
        modelgen.parse_description(text)
    assert needle in str(e.value), (needle, str(e.value))

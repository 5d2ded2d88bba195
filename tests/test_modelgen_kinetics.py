"""[neurotransmitter_kinetics] and [receptor_kinetics] blocks of the description generator (CPU side): the reference's
own tests (/root/reference/build_test/nb_macro/tests/{neurotransmitter_kinetics,receptor_kinetics}.rs) and the built-in
Approximate / Destexhe kinetics, which generated descriptions restate bit for bit."""
import numpy as np
import pytest

from snn_amd import modelgen
from snn_amd.examples_dsl import DESTEXHE_PAIR  # noqa: E402,F401

f32 = np.float32

BASIC_NT = """
[neurotransmitter_kinetics]
    type: BasicNeurotransmitterKinetics
    vars: t_max = 1, c = 0.001, conc = 0
    on_iteration:
        [if] is_spiking [then]
            conc = t_max
        [else]
            conc = 0
        [end]

        t = t + dt * -c * t + conc

        t = min(max(t, 0), t_max)
[end]"""          # build_test/nb_macro/tests/neurotransmitter_kinetics.rs:17-31, restated as data

BOUNDED_RC = """
[receptor_kinetics]
    type: BoundedReceptorKinetics
    vars: r_max = 1
    on_iteration:
        r = min(max(t, 0), r_max)
[end]"""          # build_test/nb_macro/tests/receptor_kinetics.rs:6-12

# ApproximateNeurotransmitter::apply_t_change (iterate_and_spike/mod.rs:193-196) with ITS association:
# t += dt * -clearance_constant * t + spike * t_max, then the clamp
APPROXIMATE_NT = BASIC_NT.replace("BasicNeurotransmitterKinetics", "ApproximateKinetics") \
                         .replace("vars: t_max = 1, c = 0.001, conc = 0", "vars: t_max = 1, c = 0.01, conc = 0") \
                         .replace("t = t + dt * -c * t + conc", "t += dt * -c * t + conc")

# DestexheNeurotransmitter (:148-150) and DestexheReceptor (:404-406)


def test_blocks_are_parsed_and_emitted():
    d = modelgen.parse_description(BASIC_NT + BOUNDED_RC)
    nt, rc = d.nt_kinetics, d.receptor_kinetics
    assert d.name == "BasicNeurotransmitterKinetics_BoundedReceptorKinetics" and d.neuron is None
    assert (nt.state, nt.variables, nt.bools) == ("t", [("t_max", 1.0), ("c", 0.001), ("conc", 0.0)], {"is_spiking"})
    assert (rc.state, rc.variables) == ("r", [("r_max", 1.0)])
    src = modelgen.hip_source(d)
    assert "namespace custom_nt {" in src and "namespace custom_rc {" in src
    assert "t = ((t + ((dt * (-x[1])) * t)) + x[2]);" in src and "r = min_rs(max_rs(t, 0.0f), x[0]);" in src
    src = modelgen.hip_source(modelgen.parse_description(DESTEXHE_PAIR))
    assert "const float d_r = " in src and "r += d_r;" in src


def test_reference_tests_on_the_numpy_interpreter():
    """neurotransmitter_kinetics.rs:33-71: 1000 silent iterations, one spike, 999 silent ones -- t equal to
    ApproximateNeurotransmitter's at every step; receptor_kinetics.rs:15-24: r stays inside [0, r_max]."""
    import modelgen_ref
    d = modelgen.parse_description(BASIC_NT + BOUNDED_RC)
    apply_t = modelgen_ref.make_kinetics_step(d.nt_kinetics)
    st = {"t": np.zeros(1, f32), "t_max": np.ones(1, f32), "c": np.full(1, 0.001, f32), "conc": np.zeros(1, f32)}
    t = f32(0.0)
    for it in range(2000):
        spiking = it == 1000
        apply_t(st, is_spiking=np.array([spiking]), v=np.zeros(1, f32), dt=np.full(1, 0.1, f32))
        t = t + (f32(0.1) * -f32(0.001) * t + (f32(1.0) if spiking else f32(0.0)) * f32(1.0))       # :193-196
        t = min(f32(1.0), max(t, f32(0.0)))
        assert st["t"][0] == t, it
    assert 0.8 < t < 1.0
    apply_r = modelgen_ref.make_kinetics_step(d.receptor_kinetics)
    st = {"r": np.zeros(1, f32), "r_max": np.ones(1, f32)}
    for t_in in [-2., -1.5, -1., -0.5, 0., 0.5, 1., 1.5, 2.]:
        apply_r(st, t=np.full(1, t_in, f32), dt=np.full(1, 0.1, f32))
        assert st["r"][0] == min(max(f32(t_in), f32(0.0)), f32(1.0))


def chemical_network(ob, parity, nt_kind, rc_kind, model=None, seed=51):
    lay = parity.Layout([(0, 5, 5), (2, 3, 4)], [(5, 2, 3)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH if model is None else model, nt_kind=nt_kind, rc_kind=rc_kind,
                             st_kind=ob.ST_POISSON, electrical=True, chemical=True)
    n, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(seed)
    net["current_voltage"] = ob.uniform_array(seed, n, -65.0, 30.0)
    net["nt_flags"][...] = rng.random((n, 3)) < 0.7
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][...] = 1
    net["rc_g"][:, 0] = 2.0
    net["st_nt_flags"][:, :2] = 1
    net["st_chance_of_firing"] = ob.uniform_array(seed + 1, nc, 0.0, 0.05)
    net["st_seed"] = np.arange(300, 300 + nc, dtype=np.uint32)
    net.fill_graph(seed + 2, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    return net


def heterogeneous(ob, n, nc, seed):
    return {"t_max": (ob.uniform_array(seed, n * 3, 0.5, 1.0).reshape(n, 3),
                      ob.uniform_array(seed + 1, nc * 3, 0.5, 1.0).reshape(nc, 3)),
            "c": (ob.uniform_array(seed + 2, n * 3, 0.005, 0.05).reshape(n, 3),
                  ob.uniform_array(seed + 3, nc * 3, 0.005, 0.05).reshape(nc, 3))}


def generated_approximate(ob, parity, modelgen_ref):
    d = modelgen.parse_description(APPROXIMATE_NT + BOUNDED_RC)
    net = chemical_network(ob, parity, ob.NT_CUSTOM, ob.RC_CUSTOM)
    modelgen_ref.attach_nt_kinetics(net, d.nt_kinetics)
    modelgen_ref.attach_receptor_kinetics(net, d.receptor_kinetics)
    het = heterogeneous(ob, net.n_neurons, net.n_cells, 60)
    net["nt_custom_vars"][0], net["st_nt_custom_vars"][0] = het["t_max"]
    net["nt_custom_vars"][1], net["st_nt_custom_vars"][1] = het["c"]
    return d, net


def built_in_approximate(ob, parity):
    net = chemical_network(ob, parity, ob.NT_APPROX, ob.RC_APPROX)
    het = heterogeneous(ob, net.n_neurons, net.n_cells, 60)
    net["nt_t_max"], net["st_nt_t_max"] = het["t_max"]
    net["nt_clearance"], net["st_nt_clearance"] = het["c"]
    return net


def assert_same_run(a, b, steps):
    for net in (a, b):
        net.run(steps, voltage_history=True, spike_history=True)
    assert a.spike_history.sum() > 20
    assert np.array_equal(a.spike_history, b.spike_history)
    assert np.array_equal(a.voltage_history.view(np.uint32), b.voltage_history.view(np.uint32))
    for name in ("nt_t", "st_nt_t", "rc_r", "rc_current", "weights"):
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name
    assert a["nt_t"].max() > 0.1 and a["rc_r"].max() > 0.01


def test_generated_approximate_kinetics_equal_the_built_in_ones():
    """A network whose transmitter and receptor kinetics come from descriptions that restate the Approximate kinetics
    (and a receptor clamp that never bites: 0 <= t <= r_max) is bit-identical to the built-in kinetics."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    _, generated = generated_approximate(ob, parity, modelgen_ref)
    assert_same_run(generated, built_in_approximate(ob, parity), 600)


def test_generated_destexhe_kinetics_equal_the_built_in_ones():
    """DestexheNeurotransmitter / DestexheReceptor written in the DSL (exp; a differential equation for r) against the
    built-in pair that BASELINE config C3 runs."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    d = modelgen.parse_description(DESTEXHE_PAIR)
    generated = chemical_network(ob, parity, ob.NT_CUSTOM, ob.RC_CUSTOM)
    modelgen_ref.attach_nt_kinetics(generated, d.nt_kinetics)
    modelgen_ref.attach_receptor_kinetics(generated, d.receptor_kinetics)
    built_in = chemical_network(ob, parity, ob.NT_DESTEXHE, ob.RC_DESTEXHE)
    n, nc = built_in.n_neurons, built_in.n_cells
    beta = ob.uniform_array(70, n * 3, 0.5, 2.0).reshape(n, 3)
    k_p = ob.uniform_array(71, n * 3, 3.0, 8.0).reshape(n, 3)
    built_in["rc_beta"] = beta
    built_in["nt_k_p"] = k_p
    generated["rc_custom_vars"][1] = beta
    generated["nt_custom_vars"][2] = k_p
    assert_same_run(generated, built_in, 600)


@pytest.mark.parametrize("text,needle", [
    (BASIC_NT.replace("t = min(max(t, 0), t_max)", "t = min(max(t, 0), t_max) + r"), "unknown variable 'r'"),
    (BASIC_NT.replace("conc = 0\n        [end]", "is_spiking = false\n        [end]"), "cannot assign to 'is_spiking'"),
    (BASIC_NT.replace("c = 0.001, conc = 0", "c = 0.001, conc = 0, t = 0.5"), "starts at 0"),
    (BASIC_NT.replace("vars: t_max = 1", "vars: dt = 1, t_max = 1"), "'dt' is reserved"),
    (BASIC_NT + BASIC_NT, "more than one [neurotransmitter_kinetics]"),
    (BOUNDED_RC.replace("r = min(max(t, 0), r_max)", "r = min(max(t, 0), r_max) * v"), "unknown variable 'v'"),
    (BOUNDED_RC.replace("r = min(max(t, 0), r_max)", "t = 0"), "cannot assign to 't'"),
])
def test_kinetics_errors_name_the_problem(text, needle):
    with pytest.raises(modelgen.ModelError) as e:
        modelgen.parse_description(text)
    assert needle in str(e.value), (needle, str(e.value))


# ---- on_electrochemical_iteration ------------------------------------------------------------------------
ELECTROCHEMICAL_REF = """
[neuron]
    type: ElectroChemicalIntegrateAndFire
    vars: e = 0, v_reset = -75, v_th = -55, modifier = 2
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        dv/dt = -(v - e) + i
    on_electrochemical_iteration:
        receptors.update_receptor_kinetics(t, dt)
        receptors.set_receptor_currents(v, dt)
        dv/dt = -(v - e) + i
        v = (modifier * -receptors.get_receptor_currents(dt, (modifier / 2) * c_m)) + v
        synaptic_neurotransmitters.apply_t_changes()
[end]"""          # build_test/nb_macro/tests/gpu_custom_electrochemical.rs:10-26, restated as data

# the default chemical step (lib.rs:2317-2333: receptor update, on_iteration, v -= currents, transmitter update) written
# out by hand, next to the same neuron without the section
PLAIN_STEP = """
[neuron]
    type: PlainStep
    vars: e = -48, v_reset = -70, v_th = -50, current_voltage = -65, c_m = 2, gap_conductance = 1
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        v = v + (-(v - e) + i) * dt
[end]"""
RESTATED_STEP = PLAIN_STEP.replace("PlainStep", "RestatedStep").replace("[end]", """    on_electrochemical_iteration:
        receptors.update_receptor_kinetics(t, dt)
        receptors.set_receptor_currents(v, dt)
        v = v + (-(v - e) + i) * dt
        v -= receptors.get_receptor_currents(dt, c_m)
        synaptic_neurotransmitters.apply_t_changes()
[end]""")


def custom_chemical_network(ob, parity, modelgen_ref, text, electrical=True, chemical=True):
    m = modelgen.parse(text)
    net = chemical_network(ob, parity, ob.NT_APPROX, ob.RC_APPROX, model=ob.CUSTOM)
    net.electrical, net.chemical = int(electrical), int(chemical)
    modelgen_ref.attach(net, m)
    net["current_voltage"] = ob.uniform_array(80, net.n_neurons, -68.0, -52.0)
    net["do_plasticity"] = 0            # these fast-firing leaky neurons drive STDP weights (and with them t) without bound
    return m, net


def test_electrochemical_section_is_parsed_and_emitted():
    m = modelgen.parse(ELECTROCHEMICAL_REF)
    kinds = [st[0] for st in m.on_electrochemical_iteration]
    assert kinds == ["rc_update", "rc_set", "diff", "assign", "nt_apply"]
    src = modelgen.hip_source(m)
    assert "constexpr bool HAS_ELECTROCHEMICAL = true;" in src
    body = src[src.index("void on_electrochemical_iteration"):]
    order = [body.index(t) for t in ("chem.update_receptor_kinetics();", "chem.set_receptor_currents(v);",
                                     "const float d_v", "chem.get_receptor_currents(dt, ((x[3] / 2.0f) * c_m))",
                                     "chem.apply_t_changes(v);", "v += d_v;")]
    assert order == sorted(order)                      # the differential equation is applied after the last statement
    assert "HAS_ELECTROCHEMICAL = false" in modelgen.hip_source(modelgen.parse(PLAIN_STEP))


def test_restated_default_step_equals_the_default_step():
    """With transmission on, a neuron whose on_electrochemical_iteration spells out the default sequence is
    bit-identical to the same neuron without the section; with transmission off the section is not used at all."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    for chemical in (True, False):
        runs = []
        for text in (PLAIN_STEP, RESTATED_STEP):
            _, net = custom_chemical_network(ob, parity, modelgen_ref, text, chemical=chemical)
            net.run(500, voltage_history=True, spike_history=True)
            runs.append(net)
        a, b = runs
        assert a.custom_has_chem == 0 and b.custom_has_chem == 1
        assert a.spike_history.sum() > 20
        assert np.array_equal(a.spike_history, b.spike_history)
        assert np.array_equal(a.voltage_history.view(np.uint32), b.voltage_history.view(np.uint32))
        for name in ("nt_t", "rc_r", "rc_current", "weights"):
            assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name
        if chemical:
            assert a["rc_r"].max() > 0.01 and np.abs(a["rc_current"]).max() > 0.0


def test_reference_electrochemical_model_differs_from_its_default_step():
    """gpu_custom_electrochemical.rs's model doubles the receptor current's effect (modifier = 2 with c_m scaled by
    modifier / 2): its chemical run is NOT the run of the same neuron without the section -- the section is in use."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    _, with_section = custom_chemical_network(ob, parity, modelgen_ref, ELECTROCHEMICAL_REF)
    _, without = custom_chemical_network(
        ob, parity, modelgen_ref, ELECTROCHEMICAL_REF[:ELECTROCHEMICAL_REF.index("    on_electrochemical_iteration:")] + "[end]")
    for net in (with_section, without):
        net.run(300, voltage_history=True)
    assert np.isfinite(with_section.voltage_history).all()
    assert not np.array_equal(with_section.voltage_history, without.voltage_history)


@pytest.mark.parametrize("text,needle", [
    (ELECTROCHEMICAL_REF.replace("update_receptor_kinetics(t, dt)", "update_receptor_kinetics(dt)"), "takes (t, dt)"),
    (ELECTROCHEMICAL_REF.replace("set_receptor_currents(v, dt)", "set_receptor_currents(v)"), "takes (voltage, dt)"),
    (ELECTROCHEMICAL_REF.replace("apply_t_changes()", "apply_t_changes(v)"), "takes no arguments"),
    (ELECTROCHEMICAL_REF.replace("receptors.set_receptor_currents(v, dt)", "receptors.reset()"), "cannot call receptors.reset()"),
    (ELECTROCHEMICAL_REF.replace("        dv/dt = -(v - e) + i\n    on_electrochemical",
                                 "        dv/dt = -(v - e) + i - receptors.get_receptor_currents(dt, c_m)\n    on_electrochemical"),
     "belongs to on_electrochemical_iteration"),
    (PLAIN_STEP.replace("v = v + (-(v - e) + i) * dt", "receptors.update_receptor_kinetics(t, dt)"), "cannot call"),
])
def test_electrochemical_errors_name_the_problem(text, needle):
    with pytest.raises(modelgen.ModelError) as e:
        modelgen.parse(text)
    assert needle in str(e.value), (needle, str(e.value))

"""[receptors] blocks of the description generator (CPU side): the reference's own receptor sets
(/root/reference/build_test/nb_macro/tests/{receptors,shared_receptors,neuron_receptor_integration}.rs) on the C
oracle, and a set that restates the built-in ionotropic AMPA / GABA currents against the built-in receptors."""
import numpy as np
import pytest

from snn_amd import modelgen
from snn_amd.examples_dsl import MIXED, STEP_NEURON  # noqa: E402,F401

f32 = np.float32

LIF = """
[neuron]
    type: {name}
    receptors: {receptors}
    vars: e = 0, v_reset = -75, v_th = -55
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        dv/dt = (v - e) + i
[end]"""          # the neurons of neuron_receptor_integration.rs:21-71

MULTIPLE = """
[receptors]
    type: MultipleReceptors
    neurotransmitter: A
    vars: current = 0, g = 1, e = 0
    on_iteration:
        current = g * r * (v - e)
    neurotransmitter: B
    vars: current = 0, g = 1, e = 0
    on_iteration:
        current = 2 * g * r * (v - e)
[end]"""          # shared_receptors.rs:5-15


# Ionotropic AMPA and GABA currents (iterate_and_spike/mod.rs:1103-1105, 1164-1166) in slots 0 and 2 of the exchange,
# nothing in slot 1
IONOTROPIC_LIKE = """
[receptors]
    type: AmpaGabaReceptors
    neurotransmitter: AMPA
    vars: current = 0, g = 1, e = 0
    on_iteration:
        current = g * r * (v - e)
    neurotransmitter: Unused
    vars: idle = 0
    on_iteration:
        idle = idle
    neurotransmitter: GABA
    vars: current = 0, g = 1.2, e = -80
    on_iteration:
        current = g * r * (v - e)
[end]"""



def test_receptor_sets_are_parsed_and_emitted():
    d = modelgen.parse_description(MIXED + LIF.format(name="MixedIntegrateAndFire", receptors="MixedReceptors"))
    rx = d.receptors
    assert d.name == "MixedIntegrateAndFire_MixedReceptors" and d.neuron.receptors == "MixedReceptors"
    assert rx.variables == [("m", 0.0), ("Iono$current", 0.0), ("Iono$g", 1.0), ("Iono$e", 0.0), ("Meta$s", 1.0)]
    assert [(t[0], t[2]) for t in rx.types] == [("Iono", 1), ("Meta", None)]
    src = modelgen.hip_source(d)
    assert "constexpr int CURRENT_INDEX[3] = {1, -1, -1};" in src
    assert "x[1] = (((x[2] * x[0]) * r) * (v - x[3]));" in src and "x[0] = (x[4] * r);" in src
    multi = modelgen.parse_description(MULTIPLE + LIF.format(name="MultiIntegrateAndFire", receptors="MultipleReceptors"))
    assert [(t[0], t[2]) for t in multi.receptors.types] == [("A", 0), ("B", 3)]


def single_neuron(ob, parity, modelgen_ref, text, flags):
    """one generated neuron fed by one silent spike-train cell whose transmitter concentrations the test sets by hand
    (t reaches the receptor as the weighted average over the present edges: weight 1, one edge)"""
    d = modelgen.parse_description(text)
    lay = parity.Layout([(1, 1, 1)], [(0, 1, 1)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_RATE, electrical=False, chemical=True)
    modelgen_ref.attach(net, d.neuron)
    modelgen_ref.attach_receptors(net, d.receptors)
    net["connections"][1, 0] = 1
    net["weights"][1, 0] = 1.0
    net["rc_flags"][0, :] = flags
    net["st_nt_flags"][0, :] = 1
    net["st_nt_t_max"] = 10.0
    net["st_nt_clearance"] = 0.0                       # the cell never fires: its t stays what the test sets
    return d, net


VOLTAGES = [-50., -40., -30., -20., -10., 0., 10., 20., 30., 40., 50.]      # receptors.rs:34
T = [0., 0.25, 0.5, 0.75, 0.1, 0.75, 0.5, 0.25, 0.]                          # receptors.rs:35


def test_receptor_currents_known_answers():
    """receptors.rs: no receptor inserted -> no current; with the X receptor (Approximate kinetics: r = t) the current
    after update_receptor_kinetics / set_receptor_currents is t * voltage; shared_receptors' B receptor doubles it and
    the sum runs over the receptors present."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    text = MULTIPLE + LIF.format(name="MultiIntegrateAndFire", receptors="MultipleReceptors")
    names = [n for n, _ in modelgen.parse_description(text).receptors.variables]
    for flags, scale in (((0, 0, 0), (0.0, 0.0)), ((1, 0, 0), (1.0, 0.0)), ((1, 1, 0), (1.0, 2.0))):
        d, net = single_neuron(ob, parity, modelgen_ref, text, flags)
        for voltage in VOLTAGES:
            for t in T:
                net["current_voltage"] = voltage
                net["st_nt_t"][0, :] = t
                net.inputs()
                net.update_neurons()
                a = net["rx_vars"][names.index("A$current")][0]
                b = net["rx_vars"][names.index("B$current")][0]
                assert a == f32(scale[0]) * (f32(1.0) * f32(t) * (f32(voltage) - f32(0.0)))
                assert b == f32(scale[1]) * (f32(1.0) * f32(t) * (f32(voltage) - f32(0.0)))
                # iterate_with_neurotransmitter_and_spike: dv = ((v - e) + i) * dt; v += dv; v -= total * (dt / c_m)
                v0 = f32(voltage)
                want = v0 + (v0 - f32(0.0) + f32(0.0)) * f32(0.1)
                want = f32(want) - (f32(0.0) + a * f32(flags[0]) + b * f32(flags[1])) * (f32(0.1) / f32(1.0)) \
                    if any(flags) else f32(want)
                if want < f32(-55.0):
                    assert net["current_voltage"][0] == want, (flags, voltage, t)


def test_metabotropic_variable_modulates_the_next_step():
    """shared_receptors' MixedReceptors: the Meta receptor writes the top-level m = s * r AFTER the Iono receptor has
    used the m of the step before (declaration order, lib.rs:7512-7543)."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    text = MIXED + LIF.format(name="MixedIntegrateAndFire", receptors="MixedReceptors")
    names = [n for n, _ in modelgen.parse_description(text).receptors.variables]
    d, net = single_neuron(ob, parity, modelgen_ref, text, (1, 1, 0))
    net["current_voltage"] = -60.0
    net["custom_vars"][2] = 1e9                        # v_th: never spikes
    prev_m = f32(0.0)
    for t in (0.5, 0.25, 1.0, 0.0, 0.75):
        net["st_nt_t"][0, :] = t
        v = net["current_voltage"][0]
        net.inputs()
        net.update_neurons()
        assert net["rx_vars"][names.index("Iono$current")][0] == (f32(1.0) * prev_m * f32(t)) * (v - f32(0.0))
        prev_m = f32(1.0) * f32(t)
        assert net["rx_vars"][names.index("m")][0] == prev_m


def test_restated_ionotropic_set_equals_the_built_in_receptors():
    """A generated set that spells out the AMPA and GABA currents, in a network of generated neurons with Poisson rows:
    bit-identical to the same neurons on the built-in ionotropic receptors (NMDA absent in both)."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    from test_modelgen_kinetics import chemical_network
    runs = []
    for text in (STEP_NEURON.format(name="PlainReceptors", receptors=""),
                 IONOTROPIC_LIKE + STEP_NEURON.format(name="OwnReceptors", receptors="receptors: AmpaGabaReceptors\n    ")):
        d = modelgen.parse_description(text)
        net = chemical_network(ob, parity, ob.NT_APPROX, ob.RC_APPROX, model=ob.CUSTOM)
        modelgen_ref.attach(net, d.neuron)
        if d.receptors is not None:
            modelgen_ref.attach_receptors(net, d.receptors)
        net["current_voltage"] = ob.uniform_array(80, net.n_neurons, -68.0, -52.0)
        net["do_plasticity"] = 0
        net["rc_flags"][:, 1] = 0                      # no NMDA receptor
        g = ob.uniform_array(81, net.n_neurons, 0.5, 2.0)
        if d.receptors is None:
            net["rc_g"][:, 0] = g
        else:
            names = [n for n, _ in d.receptors.variables]
            net["rx_vars"][names.index("AMPA$g")] = g
        net.run(500, voltage_history=True, spike_history=True)
        runs.append(net)
    a, b = runs
    assert a.spike_history.sum() > 20 and a["rc_r"].max() > 0.01
    assert np.array_equal(a.spike_history, b.spike_history)
    assert np.array_equal(a.voltage_history.view(np.uint32), b.voltage_history.view(np.uint32))
    names = [n for n, _ in modelgen.parse_description(IONOTROPIC_LIKE + STEP_NEURON.format(
        name="OwnReceptors", receptors="receptors: AmpaGabaReceptors\n    ")).receptors.variables]
    assert np.array_equal(b["rx_vars"][names.index("AMPA$current")].view(np.uint32), a["rc_current"][:, 0].view(np.uint32))
    assert np.array_equal(b["rx_vars"][names.index("GABA$current")].view(np.uint32), a["rc_current"][:, 2].view(np.uint32))


@pytest.mark.parametrize("text,needle", [
    (MULTIPLE, "belong to a generated [neuron]"),
    (MULTIPLE + LIF.format(name="N", receptors="Other"), "unknown receptors type 'Other'"),
    (MULTIPLE + LIF.format(name="N", receptors="MultipleReceptors").replace("    receptors: MultipleReceptors\n", ""),
     "is not used"),
    (MULTIPLE.replace("current = g * r * (v - e)", "current = g * r * (v - q)") + LIF.format(name="N", receptors="MultipleReceptors"),
     "unknown variable 'q'"),
    (MULTIPLE.replace("    neurotransmitter: B", "    receptors: r1, r2\n    neurotransmitter: B")
     + LIF.format(name="N", receptors="MultipleReceptors"), "A: unknown variable 'r'"),      # A's states are r1, r2 now
    (MULTIPLE.replace("    neurotransmitter: B", "    receptors: r1, g\n    neurotransmitter: B")
     + LIF.format(name="N", receptors="MultipleReceptors"), "shares its name with a variable"),
    (MULTIPLE.replace("    neurotransmitter: A", "    kinetics: Unknown\n    neurotransmitter: A").replace(
        "    neurotransmitter: B", "    receptors: r, r2\n    neurotransmitter: B")
     + LIF.format(name="N", receptors="MultipleReceptors"), "is not part of the description"),
    (MULTIPLE.replace("current = 2 * g * r * (v - e)", "dcurrent/dt = r") + LIF.format(name="N", receptors="MultipleReceptors"),
     "not differential equations"),
    (MULTIPLE.replace("neurotransmitter: B", "neurotransmitter: A") + LIF.format(name="N", receptors="MultipleReceptors"),
     "listed twice"),
    (MIXED.replace("m = s * r", "g = 2") + LIF.format(name="N", receptors="MixedReceptors"), "Meta: unknown variable 'g'"),
])
def test_receptor_errors_name_the_problem(text, needle):
    with pytest.raises(modelgen.ModelError) as e:
        modelgen.parse_description(text)
    assert needle in str(e.value), (needle, str(e.value))

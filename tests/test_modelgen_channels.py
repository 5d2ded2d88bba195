"""Ion channels, `^` and the function set of the neuron-description generator (CPU side): parsing / flattening, the
numpy interpreter and the C oracle's stack program against hand expansions of what nb_macro generates
(/root/reference/build_test/nb_macro/tests/{ion_channel_based_neuron,timestep_dependent_ion_channel,
gating_variables_ion_channel,function_usage}.rs)."""
import numpy as np
import pytest

from snn_amd import modelgen
from test_modelgen import CURRENTS, lif_reference_trace

f32 = np.float32

LEAK_NEURON = """
[ion_channel]
    type: TestLeak
    vars: e = 0, g = 1,
    on_iteration:
        current = g * (v - e)
[end]

[neuron]
    type: BasicIntegrateAndFire
    ion_channels: l = TestLeak
    vars: v_reset = -75, v_th = -55
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        l.update_current(v)
        dv/dt = l.current + i
[end]"""          # build_test/nb_macro/tests/ion_channel_based_neuron.rs:13-32, restated as data

CALCIUM_CLAMP = """
[ion_channel]
    type: CalciumIonChannel
    vars: e = 80, g = 0.025,
    gating_vars: s
    on_iteration:
        s.alpha = 1.6 / (1 + exp(-0.072 * (v - 5)))
        s.beta = (0.02 * (v + 8.9)) / ((exp(v + 8.9) / 5) - 1)

        s.update(dt)

        current = g * -(s.state ^ 2) * (v - e)
[end]

[neuron]
    type: VoltageClamp
    ion_channels: ca = CalciumIonChannel
    vars: dt = 0.01, v_th = 100000
    spike_detection: v >= v_th
    on_iteration:
        ca.update_current(v, dt)
[end]"""          # the channel of timestep_dependent_ion_channel.rs:50-64 inside a neuron that holds its voltage

MORRIS_LECAR = """
[ion_channel]
    type: ReducedCalciumChannel
    vars: m_ss = 0, e_ca = 120, g_ca = 4, v1 = -1.2, v2 = 18
    on_iteration:
        m_ss = 0.5 * (1. + tanh((v - v1) / v2))
        current = g_ca * m_ss * (v - e_ca)
[end]

[ion_channel]
    type: KSteadyStateChannel
    vars: g_k = 8, v_k = -84, n = 0, n_ss = 0, t_n = 0, phi = 0.067, v_3 = 12, v_4 = 17.4
    on_iteration:
        n_ss = 0.5 * (1. + tanh((v - v_3) / v_4))
        t_n = 1. / (phi * cosh((v - v_3) / (2. * v_4)))
        dn/dt = (n_ss - n) / t_n
        current = g_k * n * (v - v_k)
[end]

[ion_channel]
    type: LeakIonChannel
    vars: e = -55, g = 0.3
    on_iteration:
        current = g * (v - e)
[end]

[neuron]
    type: MorrisLecarNeuron
    ion_channels: ca_channel = ReducedCalciumChannel, k_channel = KSteadyStateChannel, leak_channel = LeakIonChannel
    vars: current_voltage = -70, c_m = 20, v_th = 25, peaks = 0, armed = 1
    on_spike:
        peaks += 1
        armed = 0
    spike_detection: v >= v_th && armed > 0.5
    on_iteration:
        [if] v < 0 [then]
            armed = 1
        [end]
        ca_channel.update_current(v)
        k_channel.update_current(v, dt)
        leak_channel.update_current(v)

        dv/dt = (-ca_channel.current - k_channel.current - leak_channel.current + i) / c_m
[end]"""          # after build_test/nb_macro/tests/morris_lecar.rs (one spike per upward threshold crossing written
                  # with a flag, in place of continuous())


# HodgkinHuxleyNeuron (hodgkin_huxley/mod.rs:49-241 with the channels of ion_channels/mod.rs:219-316) written in the
# DSL, operation for operation: three ion channels with gating variables, continuous() spike detection
HODGKIN_HUXLEY = """
[ion_channel]
    type: NaIonChannel
    vars: g_na = 120, e_na = 50
    gating_vars: m, h
    on_iteration:
        m.alpha = 0.1 * ((v + 40) / (1 - exp(-(v + 40) / 10)))
        m.beta = 4 * exp(-(v + 65) / 18)
        h.alpha = 0.07 * exp(-(v + 65) / 20)
        h.beta = 1 / (exp(-(v + 35) / 10) + 1)
        m.update(dt)
        h.update(dt)
        current = m.state ^ 3 * h.state * g_na * (v - e_na)
[end]

[ion_channel]
    type: KIonChannel
    vars: g_k = 36, e_k = -77
    gating_vars: n
    on_iteration:
        n.alpha = 0.01 * (v + 55) / (1 - exp(-(v + 55) / 10))
        n.beta = 0.125 * exp(-(v + 65) / 80)
        n.update(dt)
        current = n.state ^ 4 * g_k * (v - e_k)
[end]

[ion_channel]
    type: KLeakChannel
    vars: g_k_leak = 0.3, e_k_leak = -55
    on_iteration:
        current = g_k_leak * (v - e_k_leak)
[end]

[neuron]
    type: DslHodgkinHuxley
    ion_channels: na_channel = NaIonChannel, k_channel = KIonChannel, k_leak_channel = KLeakChannel
    vars: current_voltage = -65, dt = 0.01, c_m = 1, gap_conductance = 7, v_th = 0
    spike_detection: continuous()
    on_iteration:
        na_channel.update_current(v, dt)
        k_channel.update_current(v, dt)
        k_leak_channel.update_current(v)
        v = v + dt * (i - (na_channel.current + k_channel.current + k_leak_channel.current)) / c_m
[end]"""


def test_hodgkin_huxley_in_the_dsl_equals_the_built_in_neuron():
    """Every channel variable of the description has the built-in model's attribute name (na_channel$m$state ...), and
    a gap-junction lattice of the generated neuron runs bit-identically to the built-in HodgkinHuxleyNeuron: voltages,
    gates, currents, and the raster of the continuous() peak detector."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    m = modelgen.parse(HODGKIN_HUXLEY)
    names = [n for n, _ in m.variables]
    assert {"na_channel$m$state", "k_channel$n$alpha", "k_leak_channel$current", "was_increasing", "last_voltage"} <= set(names)
    assert m.bools == {"was_increasing"} and m.after_detection
    runs = []
    for generated in (False, True):
        net = parity.make_oracle(parity.Layout([(0, 3, 4)]), model=ob.CUSTOM if generated else ob.HH)
        if generated:
            modelgen_ref.attach(net, m)
        net["current_voltage"] = ob.uniform_array(90, 12, -70.0, -40.0)
        net["gap_conductance"] = 0.5
        gates = {k: ob.uniform_array(91 + j, 12, 0.05, 0.6) for j, k in enumerate(("m", "h", "n"))}
        for k, arr in gates.items():
            if generated:
                channel = "k_channel" if k == "n" else "na_channel"
                net["custom_vars"][names.index(f"{channel}${k}$state")] = arr
            else:
                net[f"{k}_state"] = arr
        net.fill_graph(94, 0.5, 1.5)
        net.run(6000, voltage_history=True, spike_history=True)
        runs.append(net)
    built_in, gen = runs
    assert built_in.spike_history.sum() > 5 and built_in.voltage_history.max() > 20.0
    assert np.array_equal(gen.spike_history, built_in.spike_history)
    assert np.array_equal(gen.voltage_history.view(np.uint32), built_in.voltage_history.view(np.uint32))
    for dsl, own in (("na_channel$m$state", "m_state"), ("na_channel$h$state", "h_state"), ("k_channel$n$state", "n_state"),
                     ("na_channel$current", "na_current"), ("k_channel$current", "k_current"),
                     ("k_leak_channel$current", "k_leak_current"), ("na_channel$m$alpha", "m_alpha"),
                     ("k_channel$n$beta", "n_beta")):
        assert np.array_equal(gen["custom_vars"][names.index(dsl)].view(np.uint32), built_in[own].view(np.uint32)), dsl
    assert np.array_equal(gen["custom_vars"][names.index("was_increasing")] != 0, built_in["was_increasing"] != 0)


def _state(model, n):
    st = {"current_voltage": np.full(n, model.mandatory["current_voltage"], f32),
          "dt": np.full(n, model.mandatory["dt"], f32), "c_m": np.full(n, model.mandatory["c_m"], f32),
          "gap_conductance": np.full(n, model.mandatory["gap_conductance"], f32)}
    for name, default in model.variables:
        st[name] = np.full(n, default, f32)
    return st


def test_channels_become_neuron_variables_in_struct_order():
    m = modelgen.parse(CALCIUM_CLAMP)
    assert m.ion_channels == [("ca", "CalciumIonChannel")]
    # lib.rs:4006-4035: vars, gating variables (alpha, beta, state), current; attribute names use `$` like the
    # reference's (gpu_ion_channel_with_gating_vars.rs:44-51)
    assert m.variables == [("v_th", 100000.0), ("ca$e", 80.0), ("ca$g", 0.025), ("ca$s$alpha", 0.0), ("ca$s$beta", 0.0),
                           ("ca$s$state", 0.0), ("ca$current", 0.0)]
    assert m.mandatory["dt"] == 0.01
    ml = modelgen.parse(MORRIS_LECAR)
    assert len(ml.variables) == 3 + 6 + 9 + 3 and ml.variables[-1] == ("leak_channel$current", 0.0)
    src = modelgen.hip_source(ml)
    assert "tanhf_portable(" in src and "coshf_portable(" in src
    # the channel's own differential equation is applied at the end of ITS body, before the neuron's dv is formed
    assert src.index("x[11] += d_x11;") < src.index("const float d_v =")


def test_leak_channel_neuron_is_the_plain_lif_on_both_cpu_evaluators():
    """ion_channel_based_neuron.rs:34-58: 1000 iterations per input current, voltages and spikes equal to
    ReferenceIntegrateAndFire's."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    m = modelgen.parse(LEAK_NEURON)
    n = CURRENTS.size
    ref = lif_reference_trace(1000)
    step = modelgen_ref.make_step(m)
    st = _state(m, n)
    with np.errstate(over="ignore", invalid="ignore"):
        for t in range(1000):
            step(st, CURRENTS)
            assert np.array_equal(st["current_voltage"].view(np.uint32), ref[t].view(np.uint32))
    lay = parity.Layout([(1, 1, n)], [(0, 1, n)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_RATE)
    modelgen_ref.attach(net, m)
    net["st_v_resting"] = CURRENTS
    net["connections"][n + np.arange(n), np.arange(n)] = 1
    net["weights"][n + np.arange(n), np.arange(n)] = 1.0
    net.run(1000, voltage_history=True)
    assert np.array_equal(net.voltage_history.view(np.uint32), ref.view(np.uint32))


def calcium_reference(voltages, iterations, dt):
    """ReferenceCalciumIonChannel (timestep_dependent_ion_channel.rs:6-46) written out in float32, one channel carried
    through all the voltages as the test does; BasicGatingVariable::update is ion_channels/mod.rs:40-44."""
    import oracle_binding as ob
    alpha = beta = state = f32(0.0)
    g_ca, e_ca, dt = f32(0.025), f32(80.0), f32(dt)
    out = []
    with np.errstate(all="ignore"):
        for v in voltages:
            v = f32(v)
            for _ in range(iterations):
                alpha = f32(1.6) / (f32(1.0) + f32(ob.expf(f32(-0.072) * (v - f32(5.0)))))
                beta = (f32(0.02) * (v + f32(8.9))) / ((f32(ob.expf(v + f32(8.9))) / f32(5.0)) - f32(1.0))
                alpha_state = alpha * (f32(1.0) - state)
                beta_state = beta * state
                state = state + dt * (alpha_state - beta_state)
                out.append(-f32(ob.powif(state, 2)) * g_ca * (v - e_ca))
    return np.array(out, f32), (alpha, beta, state)


VOLTAGES = [-50., -40., -30., -20., -10., 0., 10., 20., 30.]      # timestep_dependent_ion_channel.rs:71


def test_gated_channel_equals_the_reference_hand_expansion_on_both_cpu_evaluators():
    import modelgen_ref
    import oracle_binding as ob
    import parity
    m = modelgen.parse(CALCIUM_CLAMP)
    want, (alpha, beta, state) = calcium_reference(VOLTAGES, 1000, 0.01)
    assert np.isfinite(want).all() and np.abs(want).max() > 0.01
    # numpy interpreter
    step = modelgen_ref.make_step(m)
    st = _state(m, 1)
    got = []
    for v in VOLTAGES:
        st["current_voltage"] = np.full(1, v, f32)
        for _ in range(1000):
            assert not step(st, np.zeros(1, f32)).any()
            got.append(st["ca$current"][0])
    assert np.array_equal(np.array(got, f32).view(np.uint32), want.view(np.uint32))
    assert (st["ca$s$alpha"][0], st["ca$s$beta"][0], st["ca$s$state"][0]) == (alpha, beta, state)
    # C oracle (stack program): the final current of every voltage segment
    net = parity.make_oracle(parity.Layout([(0, 1, 1)]), model=ob.CUSTOM)
    modelgen_ref.attach(net, m)
    names = [name for name, _ in m.variables]
    for k, v in enumerate(VOLTAGES):
        net["current_voltage"] = v
        net.run(1000)
        assert net["custom_vars"][names.index("ca$current")][0] == want[1000 * (k + 1) - 1]
    assert net["custom_vars"][names.index("ca$s$state")][0] == state


def test_gating_variable_fields_enter_the_current_as_written():
    """gating_variables_ion_channel.rs: current = g * n.alpha * n.beta * n.state * (v - e) with the fields set by hand"""
    import modelgen_ref
    text = """
[ion_channel]
    type: TestChannel
    vars: e = 0, g = 1
    gating_vars: n
    on_iteration:
        current = g * n.alpha * n.beta * n.state * (v - e)
[end]
[neuron]
    type: Holder
    ion_channels: c = TestChannel
    vars: v_th = 100000
    spike_detection: v >= v_th
    on_iteration:
        c.update_current(v)
[end]"""
    m = modelgen.parse(text)
    step = modelgen_ref.make_step(m)
    volts = np.array([-50., -40., -30., -20., -10., 0., 10., 20., 30.], f32)
    st = _state(m, volts.size)
    st["current_voltage"] = volts.copy()
    for name in ("c$n$alpha", "c$n$beta", "c$n$state"):
        st[name][:] = 1.0
    step(st, np.zeros_like(volts))
    assert np.array_equal(st["c$current"], volts)
    st["c$g"][:] = 2.0
    st["c$e"][:] = -10.0
    st["c$n$state"][:] = 2.0
    st["c$n$alpha"][:] = 4.0
    st["c$n$beta"][:] = 3.0
    step(st, np.zeros_like(volts))
    assert np.array_equal(st["c$current"], f32(2.0 * 4.0 * 3.0 * 2.0) * (volts + f32(10.0)))


FUNCTION_CASES = [("exp", "exp(i)", np.exp), ("tanh", "tanh(i)", np.tanh), ("sinh", "sinh(i)", np.sinh),
                  ("cosh", "cosh(i)", np.cosh), ("sin", "sin(i)", np.sin), ("cos", "cos(i)", np.cos), ("tan", "tan(i)", np.tan),
                  ("min", "min(0, i)", lambda x: np.minimum(0, x)),
                  ("max", "max(0, i)", lambda x: np.maximum(0, x)),
                  ("heaviside", "heaviside(i)", lambda x: np.where(x < 0, 0, x)),
                  ("cube", "i ^ 3", lambda x: x ** 3), ("inverse square", "i ^ -2", lambda x: x ** -2.0),
                  ("minus square", "-i ^ 2", lambda x: -(x ** 2))]


@pytest.mark.parametrize("name,expr,ref", FUNCTION_CASES, ids=[c[0] for c in FUNCTION_CASES])
def test_functions_and_powers_known_answers(name, expr, ref):
    """function_usage.rs: `v = f(i)` per model; the result is the correctly rounded float32 of the exact value (the
    reference's libm promises <= 1 ULP), on the numpy interpreter and on the C oracle's stack program."""
    import modelgen_ref
    import oracle_binding as ob
    import parity
    text = f"""
[neuron]
    type: FunctionTest
    vars: v_reset = -75, v_th = 50000000
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        v = {expr}
[end]"""
    m = modelgen.parse(text)
    x = np.array([-9.5, -3.0, -1.0, -0.3, -0.01, 0.02, 0.5, 1.0, 2.5, 7.25, 11.0], f32)
    want = ref(x.astype(np.float64)).astype(f32)
    step = modelgen_ref.make_step(m)
    st = _state(m, x.size)
    step(st, x)
    ulp = np.abs(st["current_voltage"].view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, (name, ulp)
    lay = parity.Layout([(1, 1, x.size)], [(0, 1, x.size)])
    net = parity.make_oracle(lay, model=ob.CUSTOM, st_kind=ob.ST_RATE)
    modelgen_ref.attach(net, m)
    net["st_v_resting"] = x
    net["connections"][x.size + np.arange(x.size), np.arange(x.size)] = 1
    net["weights"][x.size + np.arange(x.size), np.arange(x.size)] = 1.0
    net.run(1)
    assert np.array_equal(net["current_voltage"].view(np.uint32), st["current_voltage"].view(np.uint32))


def test_morris_lecar_oscillates_and_both_cpu_evaluators_agree():
    """Three channels (tanh, cosh, a channel-local differential equation) in a small gap-junction lattice: the C
    oracle's stack program against the numpy interpreter, bit for bit, and the model actually oscillates."""
    import modelgen_ref
    import numpy_ref as nr
    import oracle_binding as ob
    import parity
    m = modelgen.parse(MORRIS_LECAR)
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]), model=ob.CUSTOM)
    modelgen_ref.attach(net, m)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(5, n, -70.0, -20.0)
    net["gap_conductance"] = 6.0
    net.fill_graph(6, 0.5, 1.5)
    names = [name for name, _ in m.variables]
    st = {k: net[k].copy() for k in ("current_voltage", "dt", "c_m", "gap_conductance")}
    for k, name in enumerate(names):
        st[name] = net["custom_vars"][k].copy()
    steps = 3000
    vh, sh, _ = nr.run_lattice(modelgen_ref.make_step(m), st, st["gap_conductance"].copy(), net["weights"].copy(),
                               net["connections"].copy(), steps)
    net.run(steps, voltage_history=True, spike_history=True)
    assert np.isfinite(vh).all() and vh.max() > 0.0 and vh.min() < -30.0 and sh.sum() > 3
    assert np.array_equal(sh, net.spike_history)
    assert np.array_equal(vh.view(np.uint32), net.voltage_history.view(np.uint32))
    for k, name in enumerate(names):
        assert np.array_equal(st[name].view(np.uint32), net["custom_vars"][k].view(np.uint32)), name


@pytest.mark.parametrize("text,needle", [
    (LEAK_NEURON.replace("l.update_current(v)", "l.update_current(v, dt)"), "takes (v)"),
    (CALCIUM_CLAMP.replace("ca.update_current(v, dt)", "ca.update_current(v)"), "takes (v, dt)"),
    (LEAK_NEURON.replace("current = g * (v - e)", "current = g * (v - e) * dt"), "reads dt"),
    (LEAK_NEURON.replace("current = g * (v - e)", "current = g * (v - q)"), "unknown variable 'q'"),
    (LEAK_NEURON.replace("current = g * (v - e)", "current = g * (v - e) + i"), "unknown variable 'i'"),
    (LEAK_NEURON.replace("l.update_current(v)", "l.update_current(l.e)"), "own fields"),
    (LEAK_NEURON.replace("dv/dt = l.current + i", "dv/dt = l.conductance + i"), "unknown variable 'l.conductance'"),
    (LEAK_NEURON.replace("ion_channels: l = TestLeak", "ion_channels: l = TestLeak, l = TestLeak"), "already taken"),
    (CALCIUM_CLAMP.replace("s.update(dt)", "s.advance(dt)"), "update(dt) and init_state()"),
    (CALCIUM_CLAMP.replace("s.update(dt)", "q.update(dt)"), "not a gating variable"),
    (MORRIS_LECAR.replace("k_channel.update_current(v, dt)", "k_channel.update_current(v, 0.5)"), "neuron's dt"),
    (LEAK_NEURON.replace("[neuron]", "[ion_channel]\n    type: TestLeak\n    on_iteration:\n        current = v\n[end]\n[neuron]"),
     "defined twice"),
])
def test_channel_errors_name_the_problem(text, needle):
    with pytest.raises(modelgen.ModelError) as e:
        modelgen.parse(text)
    assert needle in str(e.value), (needle, str(e.value))

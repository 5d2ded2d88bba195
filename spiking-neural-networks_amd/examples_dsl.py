"""Model descriptions in the reference's `neuron_builder!` DSL (build_test/nb_macro) that the package ships: the two
libraries `__graft_entry__.build()` generates to check that every hook of the description generator compiles
(csrc/snn_custom_model.hpp), shared with the tests that hold the generated code to the oracle."""

IZH_DSL = """
[neuron]
    type: DslIzhikevich
    vars: a = 0.02, b = 0.2, c = -55, d = 8, w = 30, v_th = 30, tau_m = 1, c_m = 100, current_voltage = -65
    on_spike:
        v = c
        w += d
    spike_detection: v >= v_th
    on_iteration:
        dv/dt = (0.04 * v * v + 5 * v + 140 - w + i + 0.5 * exp((v - v_th) / 20)) / c_m
        dw/dt = (a * (b * v - w)) / tau_m
[end]"""

DESTEXHE_PAIR = """
[neurotransmitter_kinetics]
    type: DslDestexheNeurotransmitter
    vars: t_max = 1, v_p = 2, k_p = 5
    on_iteration:
        t = t_max / (1 + exp(-(v - v_p) / k_p))
[end]

[receptor_kinetics]
    type: DslDestexheReceptor
    vars: alpha = 1, beta = 1
    on_iteration:
        dr/dt = alpha * t * (1 - r) - beta * r
[end]"""

MIXED = """
[receptors]
    type: MixedReceptors
    vars: m = 0
    neurotransmitter: Iono
    vars: current = 0, g = 1, e = 0
    on_iteration:
        current = g * m * r * (v - e)
    neurotransmitter: Meta
    vars: s = 1
    on_iteration:
        m = s * r
[end]"""          # shared_receptors.rs:17-28

STEP_NEURON = """
[neuron]
    type: {name}
    {receptors}vars: e = -48, v_reset = -70, v_th = -50, current_voltage = -65, c_m = 2, gap_conductance = 1
    on_spike:
        v = v_reset
    spike_detection: v >= v_th
    on_iteration:
        v = v + (-(v - e) + i) * dt
[end]"""

BURST_DSL = """
[spike_train]
    type: BurstSpikeTrain
    vars: phase = 0, freq = 0.02, envelope = 0, tau = 40, bursting = false, v_th = 25, v_resting = -5
    on_iteration:
        dphase/dt = freq
        [if] phase >= 1 [then]
            phase = phase - 1
            bursting = true
        [end]
        envelope = exp(-phase * tau / 10)
        [if] bursting && envelope < 0.5 [then]
            bursting = false
        [end]
        [if] bursting [then]
            is_spiking = !is_spiking
        [else]
            is_spiking = false
        [end]
        [if] is_spiking [then]
            v = v_th
        [else]
            v = v_resting + envelope
        [end]
[end]

[neural_refractoriness]
    type: PlateauRefractoriness
    vars: decay = 2000, plateau = 3
    effect: (v_th - v_resting) * exp((-1 / (decay / dt)) * max(time_difference - plateau, 0)) + v_resting
[end]"""

# The neuron of the reference's GPU Python module: interface_gpu/lixirnet/src/lib.rs:22-79 hands this description to
# `neuron_builder!` -- an Izhikevich neuron with bounded neurotransmitter / receptor kinetics and the DopaGluGABA
# receptor set (glutamate with an AMPA and an NMDA state, GABA, dopamine with D1 and D2 states that modulate the two
# glutamate conductances).  lattice.IzhikevichNeuron and the classes around it are built from it.
LIXIRNET = """
[neurotransmitter_kinetics]
    type: BoundedNeurotransmitterKinetics
    vars: t_max = 1, clearance_constant = 0.001, conc = 0
    on_iteration:
        [if] is_spiking [then]
            conc = t_max
        [else]
            conc = 0
        [end]

        t = t + dt * -clearance_constant * t + conc

        t = min(max(t, 0), t_max)
[end]

[receptor_kinetics]
    type: BoundedReceptorKinetics
    vars: r_max = 1
    on_iteration:
        r = min(max(t, 0), r_max)
[end]

[receptors]
    type: DopaGluGABA
    kinetics: BoundedReceptorKinetics
    vars: inh_modifier = 1, nmda_modifier = 1
    neurotransmitter: Glutamate
    receptors: ampa_r, nmda_r
    vars: current = 0, g_ampa = 1, g_nmda = 0.6, e_ampa = 0, e_nmda = 0, mg = 0.3
    on_iteration:
        current = inh_modifier * g_ampa * ampa_r * (v - e_ampa) + (1 / (1 + (exp(-0.062 * v) * mg / 3.57))) * inh_modifier * g_nmda * (nmda_r r^ nmda_modifier) * (v - e_nmda)
    neurotransmitter: GABA
    vars: current = 0, g = 1.2, e = -80
    on_iteration:
        current = g * r * (v - e)
    neurotransmitter: Dopamine
    receptors: r_d1, r_d2
    vars: s_d2 = 0, s_d1 = 0
    on_iteration:
        inh_modifier = 1 - (r_d2 * s_d2)
        nmda_modifier = 1 - (r_d1 * s_d1)
[end]

[neuron]
    type: IzhikevichNeuron
    kinetics: BoundedNeurotransmitterKinetics, BoundedReceptorKinetics
    receptors: DopaGluGABA
    vars: u = 30, a = 0.02, b = 0.2, c = -55, d = 8, v_th = 30, tau_m = 1, c_m = 100
    on_spike:
        v = c
        u += d
    spike_detection: v >= v_th
    on_iteration:
        du/dt = (a * (b * v - u)) / tau_m
        dv/dt = (0.04 * v * v + 5 * v + 140 - u + i) / c_m
[end]"""

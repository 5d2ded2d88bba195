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

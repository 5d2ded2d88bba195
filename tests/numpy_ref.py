"""A second, independent restatement of parts of the hot path in vectorised numpy float32 -- used only to
cross-check the C oracle on small cases (tests/test_oracle_vs_numpy.py).  numpy float32 arithmetic is
IEEE binary32 with one rounding per operation, so expressions written in the reference's operation
order reproduce its results exactly.  Citations as in oracle/snn_oracle.c."""
import numpy as np

f32 = np.float32
CHUNK = 256


def gap_junction_inputs(v, g, weights, conn):
    """neuron/mod.rs:702-730 with the canonical chunked ascending order (neurons only)."""
    n_tot, nn = weights.shape
    total = np.zeros(nn, f32)
    for c0 in range(0, n_tot, CHUNK):
        part = np.zeros(nn, f32)
        for p in range(c0, min(c0 + CHUNK, n_tot)):
            term = (g * (v[p] - v[:nn])).astype(f32)
            contrib = (term * weights[p]).astype(f32)
            part = np.where(conn[p] != 0, (part + contrib).astype(f32), part)
        total = (total + part).astype(f32)
    n_in = conn.sum(axis=0).astype(f32)
    return (total / np.where(n_in == 0, f32(1), n_in)).astype(f32)


def izhikevich_step(s, i_in):
    """integrate_and_fire/mod.rs:1222-1267 (electrical only); s: dict of float32 arrays, updated in place."""
    v, w = s["current_voltage"], s["w_value"]
    dt_cm = (s["dt"] / s["c_m"]).astype(f32)
    dv = ((((f32(0.04) * (v * v).astype(f32)).astype(f32) + (f32(5.0) * v).astype(f32)).astype(f32) + f32(140.0)
           ).astype(f32) - w).astype(f32)
    dv = ((dv + i_in).astype(f32) * dt_cm).astype(f32)
    dw = ((s["a"] * ((s["b"] * v).astype(f32) - w).astype(f32)).astype(f32) * (s["dt"] / s["tau_m"]).astype(f32)).astype(f32)
    v_new = (v + dv).astype(f32)
    w_new = (w + dw).astype(f32)
    spike = v_new >= s["v_th"]
    s["current_voltage"] = np.where(spike, s["c"], v_new).astype(f32)
    s["w_value"] = np.where(spike, (w_new + s["d"]).astype(f32), w_new).astype(f32)
    return spike


def lif_step(s, i_in):
    """integrate_and_fire/mod.rs:87-102, 173-190 (electrical only)."""
    v = s["current_voltage"]
    dv = (((s["leak_constant"] * (v - s["e_l"]).astype(f32)).astype(f32) +
           (s["integration_constant"] * (i_in / s["g_l"]).astype(f32)).astype(f32)).astype(f32) *
          (s["dt"] / s["tau_m"]).astype(f32)).astype(f32)
    v_new = (v + dv).astype(f32)
    rc = s["refractory_count"]
    refr = rc > 0
    spike = (~refr) & (v_new >= s["v_th"])
    s["current_voltage"] = np.where(refr | spike, s["v_reset"], v_new).astype(f32)
    s["refractory_count"] = np.where(refr, (rc - f32(1)).astype(f32),
                                     np.where(spike, (s["tref"] / s["dt"]).astype(f32), rc)).astype(f32)
    return spike


def qif_step(s, i_in):
    """integrate_and_fire/mod.rs:324-345 + handle_spiking :87-102 (electrical only)."""
    v = s["current_voltage"]
    dv = ((((s["qif_alpha"] * (v - s["v_reset"]).astype(f32)).astype(f32) * (v - s["qif_v_c"]).astype(f32)).astype(f32) +
           (s["integration_constant"] * i_in).astype(f32)).astype(f32) * (s["dt"] / s["tau_m"]).astype(f32)).astype(f32)
    v_new = (v + dv).astype(f32)
    rc = s["refractory_count"]
    refr = rc > 0
    spike = (~refr) & (v_new >= s["v_th"])
    s["current_voltage"] = np.where(refr | spike, s["v_reset"], v_new).astype(f32)
    s["refractory_count"] = np.where(refr, (rc - f32(1)).astype(f32),
                                     np.where(spike, (s["tref"] / s["dt"]).astype(f32), rc)).astype(f32)
    return spike


def simple_lif_step(s, i_in):
    """integrate_and_fire/mod.rs:1577-1605 (electrical only)."""
    v = s["current_voltage"]
    dv = (((s["slif_g"] * (v - s["slif_e"]).astype(f32)).astype(f32) + i_in).astype(f32) * s["dt"]).astype(f32)
    v_new = (v + dv).astype(f32)
    spike = v_new >= s["v_th"]
    s["current_voltage"] = np.where(spike, s["v_reset"], v_new).astype(f32)
    return spike


def adaptive_step(s, i_in, expf=None):
    """AdaptiveLeakyIntegrateAndFireNeuron integrate_and_fire/mod.rs:1001-1049; with `expf` (elementwise float32
    exponential) the exponential variant :1132-1155 (electrical only)."""
    v, w = s["current_voltage"], s["w_value"]
    acc = (s["leak_constant"] * (v - s["e_l"]).astype(f32)).astype(f32)
    if expf is not None:
        e = expf(((v - s["v_th"]).astype(f32) / s["slope_factor"]).astype(f32)).astype(f32)
        acc = (acc + (s["slope_factor"] * e).astype(f32)).astype(f32)
    acc = (acc + (s["integration_constant"] * (i_in / s["g_l"]).astype(f32)).astype(f32)).astype(f32)
    acc = (acc - (w / s["g_l"]).astype(f32)).astype(f32)
    dv = (acc * (s["dt"] / s["c_m"]).astype(f32)).astype(f32)
    dw = (((s["adp_alpha"] * (v - s["e_l"]).astype(f32)).astype(f32) - w).astype(f32) *
          (s["dt"] / s["tau_m"]).astype(f32)).astype(f32)
    v_new = (v + dv).astype(f32)
    w_new = (w + dw).astype(f32)
    rc = s["refractory_count"]
    refr = rc > 0
    spike = (~refr) & (v_new >= s["v_th"])
    s["current_voltage"] = np.where(refr | spike, s["v_reset"], v_new).astype(f32)
    s["w_value"] = np.where(spike, (w_new + s["adp_beta"]).astype(f32), w_new).astype(f32)
    s["refractory_count"] = np.where(refr, (rc - f32(1)).astype(f32),
                                     np.where(spike, (s["tref"] / s["dt"]).astype(f32), rc)).astype(f32)
    return spike


def leaky_izhikevich_step(s, i_in):
    """LeakyIzhikevichNeuron integrate_and_fire/mod.rs:1336-1356 (electrical only)."""
    v, w = s["current_voltage"], s["w_value"]
    dv = (((f32(0.04) * (v * v).astype(f32)).astype(f32) + (f32(5.0) * v).astype(f32)).astype(f32) + f32(140.0)).astype(f32)
    dv = (dv - (w * (v - s["e_l"]).astype(f32)).astype(f32)).astype(f32)
    dv = ((dv + i_in).astype(f32) * (s["dt"] / s["c_m"]).astype(f32)).astype(f32)
    dw = ((s["a"] * ((s["b"] * v).astype(f32) - w).astype(f32)).astype(f32) * (s["dt"] / s["tau_m"]).astype(f32)).astype(f32)
    v_new = (v + dv).astype(f32)
    w_new = (w + dw).astype(f32)
    spike = v_new >= s["v_th"]
    s["current_voltage"] = np.where(spike, s["c"], v_new).astype(f32)
    s["w_value"] = np.where(spike, (w_new + s["d"]).astype(f32), w_new).astype(f32)
    return spike


def run_lattice(step_fn, s, g, weights, conn, steps):
    """run_lattice_electrical_synapses_only, neuron/mod.rs:1073-1088: returns (V history, raster)."""
    vh, sh = [], []
    lft = np.full(len(s["current_voltage"]), -1, np.int32)
    for t in range(steps):
        i_in = gap_junction_inputs(s["current_voltage"], g, weights, conn)
        spike = step_fn(s, i_in)
        lft[spike] = t
        vh.append(s["current_voltage"].copy())
        sh.append(spike.astype(np.uint8))
    return np.array(vh), np.array(sh), lft

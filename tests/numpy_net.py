"""The SECOND witness of the golden fixtures: a whole-network stepper in vectorised numpy float32, written from the
reference's formulas (citations relative to /root/reference/backend/src/neuron) and sharing no code with the C oracle.
tests/golden/make_golden.py writes tests/golden/*.npz with THIS stepper; the C oracle (tests/test_golden_oracle.py) and
the HIP stepper (tests/test_gpu_golden.py) are then both held to those files.

Where the two restatements could agree by construction they do not share the piece either:
  * exp / powf are the HOST libm's expf / powf called through ctypes -- the library the Rust crate links -- not the
    restated algorithm of oracle/snn_oracle_math.h;
  * the chunked canonical sum is a float32 np.add.accumulate down the rows of a chunk (strictly sequential, one rounding
    per add) over a whole [rows, columns] term matrix, not a loop over 16-column blocks;
  * plasticity walks the spiking neurons with whole-column / whole-row vector updates.
numpy float32 arithmetic is IEEE binary32 with one rounding per operation, so an expression written in the reference's
operation order reproduces its value exactly.  Test infrastructure only.
"""
import ctypes

import numpy as np

f32 = np.float32
CHUNK = 256                      # canonical reduction chunk
K = 3                            # AMPA, NMDA, GABA  iterate_and_spike/mod.rs:1323-1333

(IZHIKEVICH, LIF, HH, QIF, SIMPLE_LIF, ADAPTIVE_LIF, ADAPTIVE_EXP_LIF, LEAKY_IZHIKEVICH) = range(8)
CUSTOM = 100
NT_APPROX, NT_DESTEXHE, NT_DISCRETE_SPIKE, NT_EXPONENTIAL_DECAY = range(4)
RC_APPROX, RC_DESTEXHE, RC_EXPONENTIAL_DECAY = range(3)
ST_NONE, ST_POISSON, ST_RATE, ST_PRESET = range(4)

_libm = ctypes.CDLL("libm.so.6")
for _name, _nargs in (("expf", 1), ("powf", 2), ("tanhf", 1), ("coshf", 1), ("sinhf", 1)):
    getattr(_libm, _name).restype = ctypes.c_float
    getattr(_libm, _name).argtypes = [ctypes.c_float] * _nargs


def _elementwise(fn, *arrays):
    arrays = np.broadcast_arrays(*[np.asarray(a, f32) for a in arrays])
    out = np.empty(arrays[0].shape, f32)
    flat = [a.ravel() for a in arrays]
    o = out.reshape(-1)
    for i in range(o.size):
        o[i] = fn(*[float(a[i]) for a in flat])
    return out


def expf(x):
    """f32::exp = the platform libm's expf"""
    return _elementwise(_libm.expf, x)


def powf(x, y):
    """f32::powf = the platform libm's powf"""
    return _elementwise(_libm.powf, x, y)


def rust_min(a, b):
    """f32::min: a NaN operand yields the other one"""
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(a < b, a, b))).astype(f32)


def rust_max(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(a > b, a, b))).astype(f32)


def sequential_column_sums(terms):
    """For a [rows, columns] float32 matrix: per column, 0.0f + t[0] + t[1] + ... strictly in row order."""
    if terms.shape[0] == 0:
        return np.zeros(terms.shape[1], f32)
    with np.errstate(all="ignore"):
        return np.add.accumulate(np.vstack([np.zeros((1, terms.shape[1]), f32), terms]), axis=0, dtype=f32)[-1]


def chunked_column_sums(terms):
    """chunks of 256 presynaptic rows, sequential inside, partials added in chunk order."""
    total = np.zeros(terms.shape[1], f32)
    with np.errstate(all="ignore"):
        for c0 in range(0, terms.shape[0], CHUNK):
            total = (total + sequential_column_sums(terms[c0:c0 + CHUNK])).astype(f32)
    return total


def stdp_delta(tp, tq, a_plus, a_minus, tau_plus, tau_minus, dt):
    """STDP::update_weight, plasticity/mod.rs:45-66, as the delta it adds; tp / tq int arrays, -1 = None"""
    tp, tq = np.broadcast_arrays(np.asarray(tp, np.int64), np.asarray(tq, np.int64))
    out = np.zeros(tp.shape, f32)
    both = (tp >= 0) & (tq >= 0)
    fp, fq = tp.astype(f32), tq.astype(f32)
    pot = both & (fp < fq)
    dep = both & (fp > fq)
    if pot.any():
        x = (f32(-1.0) * np.abs(((fp[pot] - fq[pot]).astype(f32) * f32(dt)).astype(f32))).astype(f32) / f32(tau_plus)
        out[pot] = (f32(a_plus) * expf(x.astype(f32))).astype(f32)
    if dep.any():
        x = (f32(-1.0) * np.abs(((fq[dep] - fp[dep]).astype(f32) * f32(dt)).astype(f32))).astype(f32) / f32(tau_minus)
        out[dep] = ((f32(-1.0) * f32(a_minus)).astype(f32) * expf(x.astype(f32))).astype(f32)
    return out


def stdp_delta_each(tp, tq, a_plus, a_minus, tau_plus, tau_minus, dt):
    """stdp_delta with one rule PER ELEMENT (f32 arrays of the shape of tp / tq)"""
    tp, tq = np.asarray(tp, np.int64), np.asarray(tq, np.int64)
    fp, fq = tp.astype(f32), tq.astype(f32)
    both = (tp >= 0) & (tq >= 0)
    with np.errstate(all="ignore"):
        x = (f32(-1.0) * np.abs(((fp - fq).astype(f32) * dt).astype(f32))).astype(f32)
        pot = (a_plus * expf((x / tau_plus).astype(f32))).astype(f32)
        y = (f32(-1.0) * np.abs(((fq - fp).astype(f32) * dt).astype(f32))).astype(f32)
        dep = ((f32(-1.0) * a_minus).astype(f32) * expf((y / tau_minus).astype(f32))).astype(f32)
    return np.where(both & (fp < fq), pot, np.where(both & (fp > fq), dep, f32(0))).astype(f32)


class NumpyNet:
    """State = float32 / integer arrays under the oracle binding's names (so a golden-case builder's INPUT arrays can be
    taken over as they are); nothing of the oracle's code runs here."""

    def __init__(self, src):
        self.a = {k: np.array(v, copy=True) for k, v in src.arr.items()}
        self.nn, self.nc = src.n_neurons, src.n_cells
        self.n_neurons, self.n_cells = self.nn, self.nc
        self.model, self.nt_kind, self.rc_kind, self.st_kind = src.model, src.nt_kind, src.rc_kind, src.st_kind
        self.electrical, self.chemical = bool(src.electrical), bool(src.chemical)
        self.clock = int(src.clock)
        self.custom_model = getattr(src, "custom_model", None)
        self.n_tot = self.nn + self.nc
        self.voltage_history = self.spike_history = self.st_voltage_history = None

    def __getitem__(self, name):
        return self.a[name]

    # ---- step 1: inputs, neuron/mod.rs:702-754, 2086-2210 -------------------------------------------------
    def presynaptic_cell_values(self):
        """spike_train_gap_junction, neuron/mod.rs:119-137: (value, carries the conductance factor) per cell"""
        a = self.a
        lft = a["st_last_firing_time"]
        fired = lft >= 0
        val = a["st_v_resting"].copy()
        if fired.any():
            amp = (a["st_v_th"][fired] - a["st_v_resting"][fired]).astype(f32)
            td = (self.clock - lft[fired].astype(np.int64)).astype(f32)
            rate = (f32(-1.0) / (a["st_k"][fired] / a["st_dt"][fired]).astype(f32)).astype(f32)
            kind = a["st_refractoriness"][fired] if "st_refractoriness" in a else np.zeros(fired.sum(), np.uint32)
            # DeltaDirac spike_train/mod.rs:79-88: powf(2.) is x * x; ExponentialDecay :164-178
            arg = np.where(kind == 0, (rate * (td * td).astype(f32)).astype(f32), (rate * td).astype(f32)).astype(f32)
            val[fired] = ((amp * expf(arg)).astype(f32) + a["st_v_resting"][fired]).astype(f32)
        return val, fired

    def inputs(self):
        a, nn = self.a, self.nn
        w, conn = a["weights"], a["connections"] != 0
        v, g = a["current_voltage"], a["gap_conductance"]
        with np.errstate(all="ignore"):
            if self.electrical:
                term = np.empty((self.n_tot, nn), f32)
                term[:nn] = (g[None, :] * (v[:, None] - v[None, :]).astype(f32)).astype(f32)      # gap_junction :54-60
                if self.nc:
                    val, fired = self.presynaptic_cell_values()
                    term[nn:] = np.where(fired[:, None], (g[None, :] * val[:, None]).astype(f32), val[:, None])
                term = np.where(conn, (term * w).astype(f32), f32(0.0)).astype(f32)
                n_in = conn.sum(axis=0).astype(f32)
                i_in = (chunked_column_sums(term) / np.where(n_in == 0, f32(1.0), n_in)).astype(f32)  # :722-729
            else:
                i_in = np.zeros(nn, f32)                                                           # :929-931
            t_in = np.zeros((nn, K), f32)
            t_cnt = np.zeros((nn, K), f32)
            if self.chemical:
                for k in range(K):
                    t_all = np.concatenate([a["nt_t"][:, k], a["st_nt_t"][:, k]])
                    has = np.concatenate([a["nt_flags"][:, k], a["st_nt_flags"][:, k]]) != 0
                    use = conn & has[:, None]
                    term = np.where(use, (t_all[:, None] * w).astype(f32), f32(0.0)).astype(f32)
                    cnt = use.sum(axis=0).astype(f32)
                    tot = chunked_column_sums(term)
                    t_in[:, k] = np.where(cnt > 0, (tot / np.where(cnt > 0, cnt, f32(1.0))).astype(f32), f32(0.0))
                    t_cnt[:, k] = cnt
        return i_in, t_in, t_cnt

    # ---- neurotransmitter / receptor kinetics ---------------------------------------------------------------
    def nt_apply(self, prefix, voltage, spiking, dt):
        """NeurotransmitterKinetics::apply_t_change for every type a cell carries: Approximate
        iterate_and_spike/mod.rs:193-196, Destexhe :148-150, DiscreteSpike :300-302, ExponentialDecay :350-354"""
        a = self.a
        for k in range(K):
            on = a[prefix + "nt_flags"][:, k] != 0
            if not on.any():
                continue
            t, t_max = a[prefix + "nt_t"][on, k], a[prefix + "nt_t_max"][on, k]
            s = spiking[on].astype(f32)
            d = dt[on]
            with np.errstate(all="ignore"):
                if self.nt_kind == NT_DESTEXHE:
                    x = ((-(voltage[on] - a[prefix + "nt_v_p"][on, k]).astype(f32)) / a[prefix + "nt_k_p"][on, k]).astype(f32)
                    new = (t_max / (f32(1.0) + expf(x)).astype(f32)).astype(f32)
                elif self.nt_kind == NT_DISCRETE_SPIKE:
                    new = (t_max * s).astype(f32)
                else:
                    if self.nt_kind == NT_EXPONENTIAL_DECAY:
                        change = ((-t) * expf((d / (-a[prefix + "nt_clearance"][on, k])).astype(f32))).astype(f32)
                    else:
                        change = (((d * (-a[prefix + "nt_clearance"][on, k])).astype(f32)) * t).astype(f32)
                    t = (t + (change + (s * t_max).astype(f32)).astype(f32)).astype(f32)
                    new = rust_min(t_max, rust_max(t, f32(0.0)))
            a[prefix + "nt_t"][on, k] = new

    def receptors_update(self, t_in, t_cnt, v_old):
        """Ionotropic::update_receptor_kinetics + set_receptor_currents, iterate_and_spike/mod.rs:1186-1284"""
        a = self.a
        dt = a["dt"]
        with np.errstate(all="ignore"):
            for k in range(K):
                have = a["rc_flags"][:, k] != 0
                upd = have & (t_cnt[:, k] != 0)               # a type absent from the input leaves r untouched
                if upd.any():
                    r, t = a["rc_r"][upd, k], t_in[upd, k]
                    if self.rc_kind == RC_DESTEXHE:           # :404-406
                        grow = ((a["rc_alpha"][upd, k] * t).astype(f32) * (f32(1.0) - r).astype(f32)).astype(f32)
                        r = (r + ((grow - (a["rc_beta"][upd, k] * r).astype(f32)).astype(f32) * dt[upd]).astype(f32)).astype(f32)
                    elif self.rc_kind == RC_EXPONENTIAL_DECAY:  # :510-513 (rc_alpha = r_max, rc_beta = decay_constant)
                        change = ((-r) * expf((dt[upd] / (-a["rc_beta"][upd, k])).astype(f32))).astype(f32)
                        r = (r + (change + t).astype(f32)).astype(f32)
                        r = rust_min(a["rc_alpha"][upd, k], rust_max(r, f32(0.0)))
                    else:                                     # :435-437
                        r = t
                    a["rc_r"][upd, k] = r
                if have.any():
                    r, g, e, v = a["rc_r"][have, k], a["rc_g"][have, k], a["rc_e"][have, k], v_old[have]
                    if k == 1:                                # NMDA :1132-1137
                        block = ((expf((f32(-0.062) * v).astype(f32)) * a["rc_mg"][have, k]).astype(f32) / f32(3.75)).astype(f32)
                        gate = (f32(1.0) / (f32(1.0) + block).astype(f32)).astype(f32)
                        cur = (((gate * g).astype(f32) * r).astype(f32) * (v - e).astype(f32)).astype(f32)
                    else:                                     # AMPA :1103-1105, GABA :1164-1166
                        cur = ((g * r).astype(f32) * (v - e).astype(f32)).astype(f32)
                    a["rc_current"][have, k] = cur

    def receptor_currents(self):
        """Ionotropic::get_receptor_currents :1286-1304: (I_AMPA + I_NMDA + I_GABA) * (dt / c_m), present types only"""
        a = self.a
        total = np.zeros(self.nn, f32)
        with np.errstate(all="ignore"):
            for k in range(K):
                have = a["rc_flags"][:, k] != 0
                total = np.where(have, (total + a["rc_current"][:, k]).astype(f32), total).astype(f32)
            return (total * (a["dt"] / a["c_m"]).astype(f32)).astype(f32)

    # ---- step 2: the neuron models --------------------------------------------------------------------------
    def _advance(self, v, dv):
        """impl_iterate_and_spike! integrate_and_fire/mod.rs:217-255: v + dv, or v + (dv + (-currents)) when chemical"""
        if self.chemical:
            return (v + (dv + (-self.receptor_currents())).astype(f32)).astype(f32)
        return (v + dv).astype(f32)

    def _refractory(self, v_new, on_spike=None):
        """handle_spiking integrate_and_fire/mod.rs:87-102"""
        a = self.a
        rc = a["refractory_count"]
        refr = rc > 0
        spike = (~refr) & (v_new >= a["v_th"])
        a["current_voltage"] = np.where(refr | spike, a["v_reset"], v_new).astype(f32)
        a["refractory_count"] = np.where(refr, (rc - f32(1.0)).astype(f32),
                                         np.where(spike, (a["tref"] / a["dt"]).astype(f32), rc)).astype(f32)
        return spike

    def update_neurons(self, i_in, t_in, t_cnt):
        a = self.a
        v = a["current_voltage"].copy()
        dt = a["dt"]
        prev_spiking = a["is_spiking"] != 0
        m = self.model
        with np.errstate(all="ignore"):
            if self.chemical and m != CUSTOM:
                self.receptors_update(t_in, t_cnt, v)
            if m in (IZHIKEVICH, LEAKY_IZHIKEVICH):
                # IzhikevichNeuron :1222-1267; LeakyIzhikevichNeuron :1336-1356
                w = a["w_value"]
                dv = (((f32(0.04) * (v * v).astype(f32)).astype(f32) + (f32(5.0) * v).astype(f32)).astype(f32) + f32(140.0)).astype(f32)
                if m == LEAKY_IZHIKEVICH:
                    dv = (dv - (w * (v - a["e_l"]).astype(f32)).astype(f32)).astype(f32)
                else:
                    dv = (dv - w).astype(f32)
                dv = ((dv + i_in).astype(f32) * (dt / a["c_m"]).astype(f32)).astype(f32)
                dw = ((a["a"] * ((a["b"] * v).astype(f32) - w).astype(f32)).astype(f32) * (dt / a["tau_m"]).astype(f32)).astype(f32)
                v_new = self._advance(v, dv)
                w_new = (w + dw).astype(f32)
                self.nt_apply("", v_new, prev_spiking, dt)
                spike = v_new >= a["v_th"]
                a["current_voltage"] = np.where(spike, a["c"], v_new).astype(f32)
                a["w_value"] = np.where(spike, (w_new + a["d"]).astype(f32), w_new).astype(f32)
            elif m == LIF:                                    # :173-215
                dv = (((a["leak_constant"] * (v - a["e_l"]).astype(f32)).astype(f32) +
                       (a["integration_constant"] * (i_in / a["g_l"]).astype(f32)).astype(f32)).astype(f32) *
                      (dt / a["tau_m"]).astype(f32)).astype(f32)
                v_new = self._advance(v, dv)
                self.nt_apply("", v_new, prev_spiking, dt)
                spike = self._refractory(v_new)
            elif m == QIF:                                    # :324-365
                dv = ((((a["qif_alpha"] * (v - a["v_reset"]).astype(f32)).astype(f32) * (v - a["qif_v_c"]).astype(f32)).astype(f32) +
                       (a["integration_constant"] * i_in).astype(f32)).astype(f32) * (dt / a["tau_m"]).astype(f32)).astype(f32)
                v_new = self._advance(v, dv)
                self.nt_apply("", v_new, prev_spiking, dt)
                spike = self._refractory(v_new)
            elif m == SIMPLE_LIF:                             # :1577-1630
                dv = (((a["slif_g"] * (v - a["slif_e"]).astype(f32)).astype(f32) + i_in).astype(f32) * dt).astype(f32)
                v_new = self._advance(v, dv)
                self.nt_apply("", v_new, prev_spiking, dt)
                spike = v_new >= a["v_th"]
                a["current_voltage"] = np.where(spike, a["v_reset"], v_new).astype(f32)
            elif m in (ADAPTIVE_LIF, ADAPTIVE_EXP_LIF):       # :1001-1049, :1132-1155
                w = a["w_value"]
                acc = (a["leak_constant"] * (v - a["e_l"]).astype(f32)).astype(f32)
                if m == ADAPTIVE_EXP_LIF:
                    e = expf(((v - a["v_th"]).astype(f32) / a["slope_factor"]).astype(f32))
                    acc = (acc + (a["slope_factor"] * e).astype(f32)).astype(f32)
                acc = (acc + (a["integration_constant"] * (i_in / a["g_l"]).astype(f32)).astype(f32)).astype(f32)
                acc = (acc - (w / a["g_l"]).astype(f32)).astype(f32)
                dv = (acc * (dt / a["c_m"]).astype(f32)).astype(f32)
                dw = (((a["adp_alpha"] * (v - a["e_l"]).astype(f32)).astype(f32) - w).astype(f32) *
                      (dt / a["tau_m"]).astype(f32)).astype(f32)
                v_new = self._advance(v, dv)
                w_new = (w + dw).astype(f32)
                self.nt_apply("", v_new, prev_spiking, dt)
                spike = self._refractory(v_new)
                a["w_value"] = np.where(spike, (w_new + a["adp_beta"]).astype(f32), w_new).astype(f32)
            elif m == HH:
                spike = self._hodgkin_huxley(v, i_in, prev_spiking)
            elif m == CUSTOM:
                spike = self._generated(v, i_in)
            else:
                raise NotImplementedError(m)
        a["is_spiking"] = spike.astype(np.uint32)
        a["last_firing_time"] = np.where(spike, np.int32(self.clock), a["last_firing_time"]).astype(np.int32)   # mod.rs:964-966
        return spike

    def _hodgkin_huxley(self, v, i_in, prev_spiking):
        """HodgkinHuxleyNeuron hodgkin_huxley/mod.rs:156-241; gates ion_channels/mod.rs:40-44, 219-312"""
        a = self.a
        dt = a["dt"]

        def gate(state, alpha, beta):                           # BasicGatingVariable::update
            return (state + (dt * ((alpha * (f32(1.0) - state).astype(f32)).astype(f32) - (beta * state).astype(f32)).astype(f32)
                             ).astype(f32)).astype(f32)

        def e(x):
            return expf(x.astype(f32))

        v40, v65, v35, v55 = (v + f32(40.0)).astype(f32), (v + f32(65.0)).astype(f32), (v + f32(35.0)).astype(f32), (v + f32(55.0)).astype(f32)
        m_a = (f32(0.1) * (v40 / (f32(1.0) - e((-v40) / f32(10.0))).astype(f32)).astype(f32)).astype(f32)
        m_b = (f32(4.0) * e((-v65) / f32(18.0))).astype(f32)
        h_a = (f32(0.07) * e((-v65) / f32(20.0))).astype(f32)
        h_b = (f32(1.0) / (e((-v35) / f32(10.0)) + f32(1.0)).astype(f32)).astype(f32)
        n_a = ((f32(0.01) * v55).astype(f32) / (f32(1.0) - e((-v55) / f32(10.0))).astype(f32)).astype(f32)
        n_b = (f32(0.125) * e((-v65) / f32(80.0))).astype(f32)
        mg, hg, ng = gate(a["m_state"], m_a, m_b), gate(a["h_state"], h_a, h_b), gate(a["n_state"], n_a, n_b)
        i_na = (((powf(mg, f32(3.0)) * hg).astype(f32) * a["g_na"]).astype(f32) * (v - a["e_na"]).astype(f32)).astype(f32)
        i_k = ((powf(ng, f32(4.0)) * a["g_k"]).astype(f32) * (v - a["e_k"]).astype(f32)).astype(f32)
        i_kl = (a["g_k_leak"] * (v - a["e_k_leak"]).astype(f32)).astype(f32)
        for name, val in (("m_alpha", m_a), ("m_beta", m_b), ("h_alpha", h_a), ("h_beta", h_b), ("n_alpha", n_a),
                          ("n_beta", n_b), ("m_state", mg), ("h_state", hg), ("n_state", ng), ("na_current", i_na),
                          ("k_current", i_k), ("k_leak_current", i_kl)):
            a[name] = val
        i_ligand = self.receptor_currents()                     # the stored currents count even with chemical off, :156-166
        i_sum = (i_in - ((i_na + i_k).astype(f32) + i_kl).astype(f32)).astype(f32)
        v_new = (v + (((dt * i_sum).astype(f32) / a["c_m"]).astype(f32) - i_ligand).astype(f32)).astype(f32)
        self.nt_apply("", v_new, prev_spiking, dt)
        increasing = v < v_new                                  # :207-220
        spike = (v_new > a["v_th"]) & (a["was_increasing"] != 0) & ~increasing
        a["was_increasing"] = increasing.astype(np.uint32)
        a["current_voltage"] = v_new
        return spike

    def _generated(self, v, i_in):
        """a neuron_builder! model through the numpy interpreter of tests/modelgen_ref.py, with tanh / cosh / sinh / exp
        taken from the host libm's binary64 routines and rounded once to binary32"""
        import modelgen_ref
        a = self.a
        model = self.custom_model
        if not hasattr(self, "_step"):
            self._step = modelgen_ref.make_step(model)
        names = [n for n, _ in model.variables]
        state = {"current_voltage": v, "dt": a["dt"], "c_m": a["c_m"], "gap_conductance": a["gap_conductance"]}
        for k, name in enumerate(names):
            state[name] = a["custom_vars"][k].copy()
        saved = dict(modelgen_ref._FUNCTIONS)
        modelgen_ref._FUNCTIONS.update({name: (lambda x, fn=fn: fn(np.asarray(x, np.float64)).astype(f32))
                                        for name, fn in (("exp", np.exp), ("tanh", np.tanh), ("sinh", np.sinh),
                                                         ("cosh", np.cosh))})
        try:
            spike = self._step(state, i_in)
        finally:
            modelgen_ref._FUNCTIONS.update(saved)
        a["current_voltage"] = state["current_voltage"].astype(f32)
        for k, name in enumerate(names):
            a["custom_vars"][k] = state[name]
        return spike

    # ---- step 3: plasticity, deferred form  neuron/mod.rs:2308-2417, 2573-2576 -------------------------------
    def plasticity(self, spike):
        a, nn = self.a, self.nn
        if not a["do_plasticity"].any():
            return
        w, conn = a["weights"], a["connections"] != 0
        lft_all = np.concatenate([a["last_firing_time"], a["st_last_firing_time"]])
        lat = a["lattice"]

        def params(l):
            return tuple(float(a[k][l]) for k in ("stdp_a_plus", "stdp_a_minus", "stdp_tau_plus", "stdp_tau_minus", "stdp_dt"))

        kinds = self.connection_kinds()                         # [n_tot, n_lattices]: connections of a reward-modulated network are not ours
        is_mod = self.modulated()
        for j in np.nonzero(spike)[0]:
            if not a["do_plasticity"][lat[j]] or is_mod[lat[j]]:          # (a RewardModulatedLattice has no STDP rule of its own)
                continue
            rows = conn[:, j] & (kinds[:, lat[j]] == 0)         # incoming edges: the plasticity of j's lattice
            d = stdp_delta(lft_all[rows], lft_all[j], *params(lat[j]))
            w[rows, j] = (w[rows, j] + d).astype(f32)
            for l in np.unique(lat):                            # outgoing edges: the plasticity of the target's lattice
                cols = conn[j, :] & (lat == l) & (kinds[j, l] == 0)
                if cols.any():
                    d = stdp_delta(lft_all[j], a["last_firing_time"][cols], *params(l))
                    w[j, cols] = (w[j, cols] + d).astype(f32)

    def reward_modulation(self):
        """RewardModulatedLattice::update_weights_from_neurons neuron/mod.rs:3022-3054 with RewardModulatedSTDP /
        TraceRSTDP plasticity/mod.rs:126-242, deferred: both visits of an internal edge see the same delta"""
        a, nn = self.a, self.nn
        if "rm_do_modulation" not in a or not a["rm_do_modulation"].any():
            return
        lat, lft = a["lattice"], a["last_firing_time"]
        conn = a["connections"][:nn] != 0
        for l in np.nonzero(a["rm_do_modulation"])[0]:
            members = lat == l
            edge = conn & members[:, None] & members[None, :]
            p, q = np.nonzero(edge)
            dop, dt, tau_c = f32(a["rm_dopamine"][l]), f32(a["rm_dt"][l]), f32(a["rm_tau_c"][l])
            delta = stdp_delta(lft[p], lft[q], float(a["rm_a_plus"][l]), float(a["rm_a_minus"][l]),
                               float(a["rm_tau_plus"][l]), float(a["rm_tau_minus"][l]), float(dt))
            decay = expf(np.array([(-dt) / tau_c], f32))[0]
            w, c = a["weights"][p, q], a["traces"][p, q]
            with np.errstate(all="ignore"):
                dw = (f32(0.0) + delta).astype(f32)
                w = (w + (c * dop).astype(f32)).astype(f32)
                dw = (dw + delta).astype(f32)
                c = ((c * decay).astype(f32) + (tau_c * dw).astype(f32)).astype(f32)
                w = (w + (c * dop).astype(f32)).astype(f32)
            a["weights"][p, q] = w
            a["traces"][p, q] = c

    def modulated(self):
        """per lattice: held by the network's reward_modulated_lattices map -- do_modulation set, or marked so while paused
        (rm_is_modulated; neuron/mod.rs:2744, 3419-3453)"""
        a = self.a
        on = a["rm_do_modulation"].astype(bool) if "rm_do_modulation" in a else np.zeros(int(a["lattice_count"].size), bool)
        return on | a["rm_is_modulated"].astype(bool) if "rm_is_modulated" in a else on

    def connection_kinds(self):
        """per presynaptic row and post lattice: 0 a plain network's edge, 1 RewardModulatedConnection::RewardModulatedWeight,
        2 RewardModulatedConnection::Weight (conn_kind is indexed by the SOURCE lattice: neuron lattices, then spike-train ones)"""
        a = self.a
        nl = int(a["lattice_count"].size)
        if "conn_kind" not in a or not a["conn_kind"].any():
            return np.zeros((self.nn + self.nc, nl), np.uint8)
        source = np.concatenate([a["lattice"], nl + a["st_lattice"]]).astype(np.int64)
        return a["conn_kind"][source]

    def reward_cross(self):
        """The connections between lattices of a RewardModulatedLatticeNetwork (update_weights_from_neurons_across_lattices,
        neuron/mod.rs:4707-4802, and _across_reward_lattices, :4855-4977), PAIR by pair: the two connections x -> y and y -> x of
        neurons in different lattices are touched by the visits of x and of y only, so each pair is a sequence of at most two
        visits -- the spiking neurons of plastic plain lattices first, then the neurons of modulated lattices, by index.  A visit
        of z with partner o: the connection o -> z takes its rule (incoming half), then z -> o is REPLACED by a copy of o -> z that
        took the rule once more with (pre = z, post = o) (outgoing half: the reference looks up the reverse connection)."""
        a, nn, nc = self.a, self.nn, self.nc
        if "conn_kind" not in a or not a["conn_kind"].any():
            return
        kinds = self.connection_kinds()                           # [n_tot][n_lattices]
        lat = a["lattice"].astype(np.int64)
        mod = self.modulated()                                    # which map holds the lattice ...
        mod_visits = mod & a["rm_do_modulation"].astype(bool)     # ... and whether its neurons are visited (:5113)
        plastic = a["do_plasticity"].astype(bool) & ~mod
        conn = a["connections"] != 0
        lft = np.concatenate([a["last_firing_time"], a["st_last_firing_time"]]).astype(np.int64)
        spiking = a["is_spiking"].astype(bool)
        # the pairs: x a neuron, y a later neuron of another lattice or a cell (cells are never visited and have no incoming edge)
        xs, ys = np.nonzero(np.triu(np.ones((nn, nn + nc), bool), 1))
        keep = (ys >= nn) | (lat[xs] != lat[np.minimum(ys, nn - 1)])
        xs, ys = xs[keep], ys[keep]
        y_cell = ys >= nn
        yn = np.minimum(ys, nn - 1)
        ex = {"yx": conn[ys, xs], "xy": np.where(y_cell, False, conn[xs, yn])}
        kind = {"yx": kinds[ys, lat[xs]], "xy": np.where(y_cell, 0, kinds[xs, lat[yn]])}
        st = {}
        for name in ("weights", "traces", "pending", "edge_counter"):
            st[name, "yx"] = a[name][ys, xs].copy()
            st[name, "xy"] = np.where(y_cell, 0, a[name][xs, yn]).astype(a[name].dtype)

        def rule(l, which):
            return a[which][l].astype(f32)

        def stdp(l, tp, tq):          # the STDP rule of lattice l (arrays), pre / post firing times
            return stdp_delta_each(tp, tq, rule(l, "stdp_a_plus"), rule(l, "stdp_a_minus"), rule(l, "stdp_tau_plus"),
                                   rule(l, "stdp_tau_minus"), rule(l, "stdp_dt"))

        def trace_visit(sel, m, tp, tq, w, c, dw, cnt):
            """RewardModulatedSTDP::update_weight (plasticity/mod.rs:203-237) of lattice m's modulator, where sel"""
            dt, tau_c = rule(m, "rm_dt"), rule(m, "rm_tau_c")
            delta = stdp_delta_each(tp, tq, rule(m, "rm_a_plus"), rule(m, "rm_a_minus"), rule(m, "rm_tau_plus"), rule(m, "rm_tau_minus"), dt)
            with np.errstate(all="ignore"):
                dw2 = (dw + delta).astype(f32)
                second = cnt != 0
                decay = expf(((-dt) / tau_c).astype(f32))
                c2 = np.where(second, ((c * decay).astype(f32) + (tau_c * dw2).astype(f32)).astype(f32), c)
                dw2 = np.where(second, f32(0), dw2).astype(f32)
                w2 = (w + (c2 * rule(m, "rm_dopamine")).astype(f32)).astype(f32)
            return (np.where(sel, w2, w), np.where(sel, c2, c), np.where(sel, dw2, dw), np.where(sel, cnt ^ 1, cnt).astype(cnt.dtype))

        def visit(visited, z, o, o_is_neuron, inn, out):
            lz, lo = lat[z], lat[np.minimum(o, nn - 1)]
            mod_z, mod_o = mod[lz], mod[lo] & o_is_neuron
            tz, to = lft[z], lft[o]
            k_in = np.where(visited & ex[inn], kind[inn], 0)
            with np.errstate(all="ignore"):
                # incoming o -> z
                plain_rule = np.where(mod_z, lo, lz)
                do2 = (k_in == 2) & (~mod_z | (o_is_neuron & ~mod_o))
                st["weights", inn] = np.where(do2, (st["weights", inn] + stdp(plain_rule, to, tz)).astype(f32), st["weights", inn])
                m = np.where(mod_z, lz, lo)
                st["weights", inn], st["traces", inn], st["pending", inn], st["edge_counter", inn] = trace_visit(
                    k_in == 1, m, to, tz, st["weights", inn], st["traces", inn], st["pending", inn], st["edge_counter", inn])
                # outgoing z -> o: the reverse connection, updated once more, replaces it
                k_out = np.where(visited & ex[out] & ex[inn] & o_is_neuron, kind[out], 0)
                do2 = (k_out == 2) & (~mod_z | ~mod_o)
                st["weights", out] = np.where(do2, (st["weights", inn] + stdp(plain_rule, tz, to)).astype(f32), st["weights", out])
                w, c, dw, cnt = trace_visit(k_out == 1, m, tz, to, st["weights", inn], st["traces", inn], st["pending", inn], st["edge_counter", inn])
                for name, v in (("weights", w), ("traces", c), ("pending", dw), ("edge_counter", cnt)):
                    st[name, out] = np.where(k_out == 1, v, st[name, out]).astype(st[name, out].dtype)

        x_plain = plastic[lat[xs]] & spiking[xs]
        y_plain = ~y_cell & plastic[lat[yn]] & spiking[yn]
        visit(x_plain, xs, ys, ~y_cell, "yx", "xy")
        visit(y_plain, yn, xs, np.ones(xs.size, bool), "xy", "yx")
        visit(mod_visits[lat[xs]], xs, ys, ~y_cell, "yx", "xy")
        visit(~y_cell & mod_visits[lat[yn]], yn, xs, np.ones(xs.size, bool), "xy", "yx")
        for name in ("weights", "traces", "pending", "edge_counter"):
            sel = ex["yx"]
            a[name][ys[sel], xs[sel]] = st[name, "yx"][sel]
            sel = ex["xy"]
            a[name][xs[sel], yn[sel]] = st[name, "xy"][sel]

    # ---- step 6: spike trains  neuron/mod.rs:1377-1393 --------------------------------------------------------
    def spike_trains(self):
        a = self.a
        if self.st_kind == ST_POISSON:                          # spike_train/mod.rs:380-388, 411-435
            s = a["st_seed"].astype(np.uint32)
            s ^= s << np.uint32(13)
            s ^= s >> np.uint32(17)
            s ^= s << np.uint32(5)
            a["st_seed"] = s
            spike = (s.astype(f32) / f32(4294967296.0)).astype(f32) < a["st_chance_of_firing"]
        elif self.st_kind == ST_RATE:                           # :1016-1031
            step = (a["st_step"] + a["st_dt"]).astype(f32)
            spike = (a["st_rate"] != 0) & (step >= a["st_rate"])
            a["st_step"] = np.where(spike, f32(0.0), step).astype(f32)
        elif self.st_kind == ST_PRESET:                         # :803-827; an empty list never fires
            clock = (a["st_step"] + a["st_dt"]).astype(f32)
            ptr, times, counter = a["st_firing_ptr"], a["st_firing_times"], a["st_counter"]
            length = np.diff(ptr.astype(np.int64))
            spike = np.zeros(self.nc, bool)
            for s in range(self.nc):
                if length[s] and clock[s] > times[int(ptr[s]) + int(counter[s])]:
                    spike[s] = True
                    counter[s] = (int(counter[s]) + 1) % int(length[s])
            a["st_step"] = np.where(spike, f32(0.0), clock).astype(f32)
        else:
            raise NotImplementedError(self.st_kind)
        v = np.where(spike, a["st_v_th"], a["st_v_resting"]).astype(f32)
        a["st_current_voltage"] = v
        a["st_is_spiking"] = spike.astype(np.uint32)
        self.nt_apply("st_", v, spike, a["st_dt"])             # the flag the cell has just set
        clocks = a["st_clock"][a["st_lattice"]] if self.nc else np.zeros(0, np.int64)
        a["st_last_firing_time"] = np.where(spike, clocks.astype(np.int32), a["st_last_firing_time"]).astype(np.int32)
        a["st_clock"] += 1

    # ---- the loop  run_lattice_* neuron/mod.rs:1035-1088, run_lattices_* :2598-2651 ----------------------------
    def apply_reward(self, reward):
        """RewardModulatedSTDP::update (plasticity/mod.rs:199-201) on every modulated lattice"""
        a = self.a
        on = self.modulated()                                     # (a paused modulator still takes the reward, :5287-5291)
        with np.errstate(all="ignore"):
            decay = expf(((-a["rm_dt"]) / a["rm_tau_d"]).astype(f32))
            new = ((a["rm_dopamine"] * decay).astype(f32) + (a["rm_tau_d"] * f32(reward)).astype(f32)).astype(f32)
        a["rm_dopamine"] = np.where(on, new, a["rm_dopamine"]).astype(f32)

    def run(self, steps, voltage_history=True, spike_history=True, st_voltage_history=False, rewards=None):
        vh, sh, ch = [], [], []
        if not (self.electrical or self.chemical):
            steps = 0
        for it in range(steps):
            if rewards is not None:
                self.apply_reward(rewards[it])
            if self.nn:
                i_in, t_in, t_cnt = self.inputs()
                spike = self.update_neurons(i_in, t_in, t_cnt)
                self.plasticity(spike)
                self.reward_modulation()
                self.reward_cross()
                vh.append(self.a["current_voltage"].copy())
                sh.append(spike.astype(np.uint8))
            self.clock += 1
            if self.nc:
                self.spike_trains()
                ch.append(self.a["st_current_voltage"].copy())
        self.voltage_history = np.array(vh, f32).reshape(steps, self.nn) if voltage_history else None
        self.spike_history = np.array(sh, np.uint8).reshape(steps, self.nn) if spike_history else None
        self.st_voltage_history = np.array(ch, f32).reshape(steps, self.nc) if st_voltage_history else None
        return self

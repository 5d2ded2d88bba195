"""ctypes binding of the CPU oracle (oracle/snn_oracle.h) -- test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
`Net` owns numpy arrays named after the reference's struct fields / GPU buffer
names and mirrors `snn_o_net` field for field.
"""
import ctypes as C
import os
import subprocess

import numpy as np

K = 3          # AMPA, NMDA, GABA
CHUNK = 256
IZHIKEVICH, LIF, HH, QIF, SIMPLE_LIF, ADAPTIVE_LIF, ADAPTIVE_EXP_LIF, LEAKY_IZHIKEVICH, BCM_IZHIKEVICH = range(9)
CUSTOM = 100
NT_APPROX, NT_DESTEXHE, NT_DISCRETE_SPIKE, NT_EXPONENTIAL_DECAY = 0, 1, 2, 3
RC_APPROX, RC_DESTEXHE, RC_EXPONENTIAL_DECAY = 0, 1, 2
ST_NONE, ST_POISSON, ST_RATE, ST_PRESET, ST_BCM_POISSON = 0, 1, 2, 3, 4
ST_CUSTOM = 100
NT_CUSTOM = RC_CUSTOM = 100
REFRACTORINESS_CUSTOM = 2

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "_build", "libsnn_oracle.so")

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)

# (name, ctype) in the exact order of `struct snn_o_net`
_FIELDS = [
    ("n_neurons", C.c_uint32), ("n_cells", C.c_uint32),
    ("model", C.c_int32), ("nt_kind", C.c_int32), ("rc_kind", C.c_int32), ("st_kind", C.c_int32),
    ("electrical", C.c_int32), ("chemical", C.c_int32),
    ("clock", C.c_int64),
    ("current_voltage", f32p), ("gap_conductance", f32p), ("dt", f32p), ("c_m", f32p), ("v_th", f32p),
    ("is_spiking", u32p), ("last_firing_time", i32p),
    ("w_value", f32p), ("a", f32p), ("b", f32p), ("c", f32p), ("d", f32p), ("tau_m", f32p),
    ("v_reset", f32p), ("refractory_count", f32p), ("tref", f32p), ("leak_constant", f32p),
    ("integration_constant", f32p), ("e_l", f32p), ("g_l", f32p),
    ("m_state", f32p), ("h_state", f32p), ("n_state", f32p),
    ("m_alpha", f32p), ("m_beta", f32p), ("h_alpha", f32p), ("h_beta", f32p), ("n_alpha", f32p), ("n_beta", f32p),
    ("g_na", f32p), ("e_na", f32p), ("g_k", f32p), ("e_k", f32p), ("g_k_leak", f32p), ("e_k_leak", f32p),
    ("na_current", f32p), ("k_current", f32p), ("k_leak_current", f32p),
    ("was_increasing", u32p),
    ("nt_t", f32p), ("nt_t_max", f32p), ("nt_clearance", f32p), ("nt_v_p", f32p), ("nt_k_p", f32p),
    ("nt_flags", u32p),
    ("rc_g", f32p), ("rc_e", f32p), ("rc_mg", f32p), ("rc_r", f32p), ("rc_alpha", f32p), ("rc_beta", f32p),
    ("rc_current", f32p), ("rc_flags", u32p),
    ("st_current_voltage", f32p), ("st_v_th", f32p), ("st_v_resting", f32p), ("st_dt", f32p), ("st_k", f32p),
    ("st_chance_of_firing", f32p), ("st_rate", f32p), ("st_step", f32p),
    ("st_seed", u32p), ("st_is_spiking", u32p), ("st_last_firing_time", i32p),
    ("st_nt_t", f32p), ("st_nt_t_max", f32p), ("st_nt_clearance", f32p), ("st_nt_v_p", f32p), ("st_nt_k_p", f32p),
    ("st_nt_flags", u32p), ("st_lattice", u32p),
    ("n_st_lattices", C.c_uint32), ("st_clock", i64p),
    ("weights", f32p), ("connections", u8p),
    ("lattice", u32p), ("n_lattices", C.c_uint32),
    ("stdp_a_plus", f32p), ("stdp_a_minus", f32p), ("stdp_tau_plus", f32p), ("stdp_tau_minus", f32p),
    ("stdp_dt", f32p), ("do_plasticity", u32p),
    ("voltage_history", f32p), ("spike_history", u8p), ("st_voltage_history", f32p),
    ("input_current", f32p), ("input_t", f32p), ("input_count", f32p),
    ("n_threads", C.c_int32),
    ("w_col0", C.c_uint32), ("w_ld", C.c_uint32),
    ("qif_alpha", f32p), ("qif_v_c", f32p), ("slif_g", f32p), ("slif_e", f32p),
    ("lattice_first", u32p), ("lattice_count", u32p), ("avg_history", f32p), ("eeg_history", f32p),
    ("eeg_reference_voltage", C.c_float), ("eeg_distance", C.c_float), ("eeg_conductivity", C.c_float),
    ("spike_counts", u32p),
    ("adp_alpha", f32p), ("adp_beta", f32p), ("slope_factor", f32p),
    ("st_firing_ptr", u32p), ("st_firing_times", f32p), ("st_counter", u32p),
    ("traces", f32p), ("rm_do_modulation", u32p), ("rm_dopamine", f32p), ("rm_tau_d", f32p), ("rm_tau_c", f32p),
    ("rm_a_plus", f32p), ("rm_a_minus", f32p), ("rm_tau_plus", f32p), ("rm_tau_minus", f32p), ("rm_dt", f32p),
    ("rewards", f32p), ("st_refractoriness", u32p),
    ("plasticity_kind", u32p), ("bcm_decay", f32p), ("bcm_average_scalar", f32p), ("bcm_dt", f32p),
    ("bcm_average_activity", f32p), ("bcm_current_activity", f32p), ("bcm_clock", f32p), ("bcm_window", f32p),
    ("bcm_period", u32p), ("bcm_num_spikes", u32p),
    ("st_bcm_average_activity", f32p), ("st_bcm_current_activity", f32p), ("st_bcm_clock", f32p), ("st_bcm_window", f32p),
    ("st_bcm_period", u32p), ("st_bcm_num_spikes", u32p),
    ("custom_code", C.POINTER(C.c_int32)), ("custom_consts", f32p), ("custom_section", C.c_uint32 * 3),
    ("custom_nvars", C.c_uint32), ("custom_vars", f32p),
    ("st_custom_code", C.POINTER(C.c_int32)), ("st_custom_consts", f32p), ("st_custom_nvars", C.c_uint32),
    ("st_custom_vars", f32p),
    ("refr_code", C.POINTER(C.c_int32)), ("refr_consts", f32p), ("refr_nvars", C.c_uint32), ("refr_vars", f32p),
    ("nt_code", C.POINTER(C.c_int32)), ("nt_consts", f32p), ("nt_nvars", C.c_uint32), ("nt_custom_vars", f32p),
    ("st_nt_custom_vars", f32p),
    ("rc_code", C.POINTER(C.c_int32)), ("rc_consts", f32p), ("rc_nvars", C.c_uint32), ("rc_custom_vars", f32p),
    ("custom_has_chem", C.c_uint32), ("custom_chem_section", C.c_uint32),
    ("rx_code", C.POINTER(C.c_int32)), ("rx_consts", f32p), ("rx_ntypes", C.c_uint32), ("rx_nvars", C.c_uint32),
    ("rx_section", C.c_uint32 * 3), ("rx_current_index", C.c_int32 * 3), ("rx_vars", f32p),
    ("rx_multi", C.c_uint32), ("rx_kin_section", C.c_uint32 * 3),
    ("conn_kind", u8p), ("pending", f32p), ("edge_counter", u8p),
    ("rm_is_modulated", u32p),
]


class _CNet(C.Structure):
    _fields_ = _FIELDS


_PTR_DTYPE = {f32p: np.float32, u32p: np.uint32, i32p: np.int32, u8p: np.uint8, i64p: np.int64}

_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build_oracle()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER(_CNet)
        for fn in ("snn_o_inputs", "snn_o_update_neurons", "snn_o_plasticity", "snn_o_spike_trains",
                   "snn_o_reward_modulation"):
            getattr(L, fn).argtypes = [P]
            getattr(L, fn).restype = None
        for fn in ("snn_o_inputs_range", "snn_o_update_neurons_range", "snn_o_plasticity_cols"):
            getattr(L, fn).argtypes = [P, C.c_uint32, C.c_uint32]
            getattr(L, fn).restype = None
        L.snn_o_apply_reward.argtypes = [P, C.c_float]
        L.snn_o_apply_reward.restype = None
        L.snn_o_reward_modulation_cols.argtypes = [P, C.c_uint32, C.c_uint32]
        L.snn_o_reward_modulation_cols.restype = None
        L.snn_o_run.argtypes = [P, C.c_uint64]
        L.snn_o_reward_cross_check.argtypes = [P]
        L.snn_o_reward_cross_check.restype = C.c_int
        L.snn_o_run.restype = None
        for fn in ("snn_o_expf_export", "snn_o_pow3f_export", "snn_o_pow4f_export", "snn_o_tanhf_export",
                   "snn_o_sinhf_export", "snn_o_coshf_export", "snn_o_sinf_export", "snn_o_cosf_export",
                   "snn_o_tanf_export"):
            getattr(L, fn).argtypes = [C.c_float]
            getattr(L, fn).restype = C.c_float
        L.snn_o_powif_export.argtypes = [C.c_float, C.c_int]
        L.snn_o_powif_export.restype = C.c_float
        L.snn_o_powf_export.argtypes = [C.c_float, C.c_float]
        L.snn_o_powf_export.restype = C.c_float
        L.snn_o_math_bits.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_float, C.POINTER(C.c_float)]
        L.snn_o_math_bits.restype = None
        L.snn_o_stdp_delta.argtypes = [C.c_int32, C.c_int32] + [C.c_float] * 5
        L.snn_o_stdp_delta.restype = C.c_float
        L.snn_o_delta_dirac_effect.argtypes = [C.c_int64, C.c_int32] + [C.c_float] * 4
        L.snn_o_delta_dirac_effect.restype = C.c_float
        L.snn_o_exponential_decay_effect.argtypes = [C.c_int64, C.c_int32] + [C.c_float] * 4
        L.snn_o_exponential_decay_effect.restype = C.c_float
        L.snn_o_xorshift32.argtypes = [C.c_uint32]
        L.snn_o_xorshift32.restype = C.c_uint32
        L.snn_o_hash32.argtypes = [C.c_uint64, C.c_uint64]
        L.snn_o_hash32.restype = C.c_uint32
        L.snn_o_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_float, C.c_float]
        L.snn_o_uniform.restype = C.c_float
        L.snn_o_fill_graph.argtypes = [f32p, u8p, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_float, C.c_float, C.c_int]
        L.snn_o_fill_graph.restype = None
        L.snn_o_fill_graph_window.argtypes = [f32p, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint64, C.c_float, C.c_float, C.c_int]
        L.snn_o_fill_graph_window.restype = None
        L.snn_o_fill_graph_window_blocked.argtypes = [f32p, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_int]
        L.snn_o_fill_graph_window_blocked.restype = None
        L.snn_o_inputs_tiled.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32]
        L.snn_o_inputs_tiled.restype = None
        L.snn_o_inputs_csr.argtypes = [P, C.POINTER(C.c_uint64), u32p, f32p, C.c_uint32, C.c_uint32]
        L.snn_o_inputs_csr.restype = None
        L.snn_o_run_csr.argtypes = [P, C.POINTER(C.c_uint64), u32p, f32p, C.c_uint64]
        L.snn_o_run_csr.restype = None
        _lib = L
    return _lib


# ---------------------------------------------------------------------------
# Reference defaults (file:line relative to /root/reference/backend/src/neuron)
# ---------------------------------------------------------------------------
NEURON_DEFAULTS = {
    # integrate_and_fire/mod.rs:1198-1220
    IZHIKEVICH: dict(current_voltage=-65.0, gap_conductance=7.0, w_value=30.0, a=0.02, b=0.2, c=-55.0,
                     d=8.0, v_th=30.0, tau_m=1.0, c_m=100.0, dt=0.1),
    # integrate_and_fire/mod.rs:149-171
    LIF: dict(current_voltage=-75.0, refractory_count=0.0, leak_constant=-1.0, integration_constant=1.0,
              gap_conductance=7.0, v_th=-55.0, v_reset=-75.0, tau_m=10.0, c_m=100.0, g_l=10.0,
              e_l=-75.0, tref=10.0, dt=0.1),
    # hodgkin_huxley/mod.rs:80-98, ion_channels/mod.rs:23-31,205-215,255-264,299-307
    HH: dict(current_voltage=-65.0, gap_conductance=7.0, dt=0.01, c_m=1.0, v_th=0.0,
             g_na=120.0, e_na=50.0, g_k=36.0, e_k=-77.0, g_k_leak=0.3, e_k_leak=-55.0),
    # integrate_and_fire/mod.rs:299-322
    QIF: dict(current_voltage=-75.0, refractory_count=0.0, integration_constant=1.0, gap_conductance=7.0,
              qif_alpha=1.0, v_th=-55.0, v_reset=-75.0, qif_v_c=-60.0, tau_m=100.0, c_m=100.0, tref=10.0, dt=0.1),
    # integrate_and_fire/mod.rs:1552-1570
    SIMPLE_LIF: dict(current_voltage=-75.0, gap_conductance=10.0, v_th=-55.0, v_reset=-75.0, c_m=100.0,
                     slif_g=-0.1, slif_e=0.0, dt=0.1),
    # integrate_and_fire/mod.rs:969-996
    ADAPTIVE_LIF: dict(current_voltage=-75.0, refractory_count=0.0, leak_constant=-1.0, integration_constant=1.0,
                       gap_conductance=7.0, w_value=0.0, adp_alpha=6.0, adp_beta=10.0, v_th=-55.0, v_reset=-75.0,
                       tau_m=10.0, c_m=100.0, g_l=10.0, e_l=-75.0, tref=10.0, dt=0.1),
    # integrate_and_fire/mod.rs:1105-1130
    ADAPTIVE_EXP_LIF: dict(current_voltage=-75.0, refractory_count=0.0, leak_constant=-1.0, integration_constant=1.0,
                           gap_conductance=7.0, w_value=0.0, adp_alpha=6.0, adp_beta=10.0, slope_factor=1.0,
                           v_th=-55.0, v_reset=-75.0, tau_m=10.0, c_m=100.0, g_l=10.0, e_l=-75.0, tref=10.0, dt=0.1),
    # integrate_and_fire/mod.rs:1310-1331
    LEAKY_IZHIKEVICH: dict(current_voltage=-65.0, gap_conductance=7.0, w_value=30.0, a=0.02, b=0.2, c=-55.0, d=8.0,
                           v_th=30.0, tau_m=10.0, c_m=100.0, e_l=-65.0, dt=0.1),
    # integrate_and_fire/mod.rs:1409-1436
    BCM_IZHIKEVICH: dict(current_voltage=-65.0, gap_conductance=7.0, w_value=30.0, a=0.02, b=0.2, c=-55.0, d=8.0,
                         v_th=30.0, tau_m=1.0, c_m=100.0, dt=0.1),
    # generated model: the DSL's mandatory defaults (build_test/nb_macro/src/lib.rs:2200-2211); modelgen_ref.attach
    CUSTOM: dict(current_voltage=0.0, gap_conductance=10.0, c_m=1.0, dt=0.1),
}
# activity bookkeeping of BCMIzhikevichNeuron / BCMPoissonNeuron; BCM rule plasticity/mod.rs:91-95
BCM_CELL_DEFAULTS = dict(average_activity=0.0, current_activity=0.0, clock=0.0, window=500.0, period=3, num_spikes=0)
BCM_DEFAULTS = dict(bcm_decay=0.1, bcm_average_scalar=0.1, bcm_dt=0.1)
# iterate_and_spike/mod.rs:174-182 (Approximate), :136-145 (Destexhe)
NT_DEFAULTS = dict(nt_t=0.0, nt_t_max=1.0, nt_clearance=0.01, nt_v_p=2.0, nt_k_p=5.0)
# iterate_and_spike/mod.rs:1085-1094, 1115-1125, 1148-1157; Destexhe receptor :417-425
RC_DEFAULTS = dict(rc_g=(1.0, 0.6, 1.2), rc_e=(0.0, 0.0, -80.0), rc_mg=(0.0, 0.3, 0.0),
                   rc_r=0.0, rc_alpha=1.0, rc_beta=1.0, rc_current=0.0)
# spike_train/mod.rs:299-313 (Poisson), :998-1013 (Rate), :50-56 (k)
ST_DEFAULTS = dict(st_current_voltage=0.0, st_v_th=30.0, st_v_resting=0.0, st_dt=0.1, st_k=10000.0,
                   st_chance_of_firing=0.0, st_rate=0.0, st_step=0.0)
# plasticity/mod.rs:29-39
STDP_DEFAULTS = dict(stdp_a_plus=2.0, stdp_a_minus=2.0, stdp_tau_plus=4.5, stdp_tau_minus=4.5, stdp_dt=0.1)
# RewardModulatedSTDP, plasticity/mod.rs:176-189
RM_DEFAULTS = dict(rm_dopamine=0.0, rm_tau_d=20.0, rm_tau_c=0.0001, rm_a_plus=2.0, rm_a_minus=2.0, rm_tau_plus=4.5,
                   rm_tau_minus=4.5, rm_dt=0.1)

_NAMES = [n for n, _ in _FIELDS]
_PER_NEURON = set(_NAMES[_NAMES.index("current_voltage"):_NAMES.index("was_increasing") + 1]) | {
    "qif_alpha", "qif_v_c", "slif_g", "slif_e", "adp_alpha", "adp_beta", "slope_factor",
    "bcm_average_activity", "bcm_current_activity", "bcm_clock", "bcm_window", "bcm_period", "bcm_num_spikes"}
_PER_NEURON_K = {"nt_t", "nt_t_max", "nt_clearance", "nt_v_p", "nt_k_p", "nt_flags",
                 "rc_g", "rc_e", "rc_mg", "rc_r", "rc_alpha", "rc_beta", "rc_current", "rc_flags",
                 "input_t", "input_count"}
_PER_CELL = {"st_current_voltage", "st_v_th", "st_v_resting", "st_dt", "st_k", "st_chance_of_firing",
             "st_rate", "st_step", "st_seed", "st_is_spiking", "st_last_firing_time", "st_lattice", "st_counter",
             "st_refractoriness", "st_bcm_average_activity", "st_bcm_current_activity", "st_bcm_clock", "st_bcm_window",
             "st_bcm_period", "st_bcm_num_spikes"}
_PER_CELL_K = {"st_nt_t", "st_nt_t_max", "st_nt_clearance", "st_nt_v_p", "st_nt_k_p", "st_nt_flags"}
_PER_LATTICE = {"stdp_a_plus", "stdp_a_minus", "stdp_tau_plus", "stdp_tau_minus", "stdp_dt", "do_plasticity",
                "lattice_first", "lattice_count", "rm_do_modulation", "rm_is_modulated", *RM_DEFAULTS, "plasticity_kind", *BCM_DEFAULTS}


class Net:
    """A network in the oracle's flat SoA form (neurons first, spike-train cells after)."""

    def __init__(self, n_neurons, model=IZHIKEVICH, n_cells=0, st_kind=ST_NONE, n_lattices=1,
                 n_st_lattices=None, nt_kind=NT_APPROX, rc_kind=RC_APPROX,
                 electrical=True, chemical=False, dense=True):
        """dense=False leaves the [n_tot][n_neurons] graph arrays out (sparse networks: inputs_csr / run_csr)"""
        self.n_neurons, self.n_cells = int(n_neurons), int(n_cells)
        self.n_tot = self.n_neurons + self.n_cells
        self.model, self.nt_kind, self.rc_kind, self.st_kind = model, nt_kind, rc_kind, st_kind
        self.electrical, self.chemical = bool(electrical), bool(chemical)
        self.clock = 0
        self.n_lattices = int(n_lattices)
        self.n_st_lattices = int(n_st_lattices if n_st_lattices is not None else (1 if n_cells else 0))
        self.n_threads = 1
        self.w_col0 = 0
        self.w_ld = 0
        # EEGHistory defaults, neuron/mod.rs:246-255
        self.eeg_reference_voltage, self.eeg_distance, self.eeg_conductivity = 0.007, 0.8, 251.0
        self.avg_history = self.eeg_history = self.spike_counts = None
        self.arr = {}
        nn, nc = self.n_neurons, self.n_cells
        for name, ct in _FIELDS:
            if ct not in _PTR_DTYPE:
                continue
            dt = _PTR_DTYPE[ct]
            if name in _PER_NEURON:
                shape = (nn,)
            elif name in _PER_NEURON_K:
                shape = (nn, K)
            elif name in _PER_CELL:
                shape = (nc,)
            elif name in _PER_CELL_K:
                shape = (nc, K)
            elif name in _PER_LATTICE:
                shape = (self.n_lattices,)
            elif name == "lattice":
                shape = (nn,)
            elif name == "st_clock":
                shape = (self.n_st_lattices,)
            elif name in ("weights", "connections", "traces", "pending", "edge_counter"):
                if not dense:
                    continue
                shape = (self.n_tot, nn)
            elif name == "conn_kind":
                shape = (self.n_lattices + self.n_st_lattices, self.n_lattices)
            elif name == "input_current":
                shape = (nn,)
            else:   # histories: allocated per run
                continue
            self.arr[name] = np.zeros(shape, dtype=dt)
        a = self.arr
        for k, v in NEURON_DEFAULTS[model].items():
            a[k][...] = v
        a["last_firing_time"][...] = -1
        a["st_last_firing_time"][...] = -1
        for k, v in NT_DEFAULTS.items():
            a[k][...] = v
            a["st_" + k][...] = v
        for k, v in RC_DEFAULTS.items():
            a[k][...] = np.asarray(v, dtype=np.float32)
        for k, v in ST_DEFAULTS.items():
            a[k][...] = v
        for k, v in STDP_DEFAULTS.items():
            a[k][...] = v
        for k, v in RM_DEFAULTS.items():
            a[k][...] = v
        for k, v in BCM_DEFAULTS.items():
            a[k][...] = v
        for k, v in BCM_CELL_DEFAULTS.items():
            a["bcm_" + k][...] = v
            a["st_bcm_" + k][...] = v
        self.rewards = None
        self.custom_section = np.zeros(3, np.uint32)
        self.custom_nvars = 0
        self.st_custom_nvars = 0
        self.refr_nvars = 0
        self.nt_nvars = 0
        self.custom_has_chem = 0
        self.rx_ntypes = 0
        self.rx_nvars = 0
        self.rx_section = np.zeros(3, np.uint32)
        self.rx_multi = 0
        self.rx_kin_section = np.zeros(3, np.uint32)
        self.rx_current_index = np.full(3, -1, np.int32)
        self.custom_chem_section = 0
        self.rc_nvars = 0
        if nt_kind == NT_EXPONENTIAL_DECAY:       # decay_constant, iterate_and_spike/mod.rs:336-343
            a["nt_clearance"][...] = 2.0
            a["st_nt_clearance"][...] = 2.0
        if rc_kind == RC_EXPONENTIAL_DECAY:       # r_max (alpha array) 1, decay_constant (beta array) 2, :525-533
            a["rc_beta"][...] = 2.0
        a["st_firing_ptr"] = np.zeros(nc + 1, np.uint32)
        a["st_firing_times"] = np.zeros(0, np.float32)
        a["st_seed"][...] = np.arange(1, nc + 1, dtype=np.uint32)
        a["lattice_count"][...] = 0
        a["lattice_count"][0] = nn              # single lattice by default; parity.make_oracle sets real ranges
        self.voltage_history = self.spike_history = self.st_voltage_history = None

    def __getitem__(self, name):
        return self.arr[name]

    def __setitem__(self, name, value):
        self.arr[name][...] = value

    def set_firing_times(self, per_cell):
        """PresetSpikeTrain::firing_times for every cell (list of sequences, one per cell)."""
        assert len(per_cell) == self.n_cells
        lens = np.array([len(x) for x in per_cell], np.uint32)
        self.arr["st_firing_ptr"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        self.arr["st_firing_times"] = np.ascontiguousarray(
            np.concatenate([np.asarray(x, np.float32) for x in per_cell]) if lens.sum() else np.zeros(0, np.float32))

    def connect_all_to_all(self, weight=1.0, with_diagonal=False):
        self.arr["connections"][...] = 1
        if not with_diagonal:
            idx = np.arange(self.n_neurons)
            self.arr["connections"][idx, idx] = 0
        self.arr["weights"][...] = np.float32(weight) * self.arr["connections"]

    def fill_graph(self, seed, lo, hi, with_diagonal=False):
        lib().snn_o_fill_graph(self.arr["weights"].ctypes.data_as(f32p),
                               self.arr["connections"].ctypes.data_as(u8p),
                               self.n_tot, self.n_neurons, seed, lo, hi, int(with_diagonal))

    def _cnet(self):
        c = _CNet()
        for name, ct in _FIELDS:
            if ct in _PTR_DTYPE:
                arr = self.arr.get(name)
                if arr is None:
                    arr = getattr(self, name, None)
                if arr is None or arr.size == 0:
                    setattr(c, name, C.cast(None, ct))
                else:
                    assert arr.flags["C_CONTIGUOUS"] and arr.dtype == _PTR_DTYPE[ct], name
                    setattr(c, name, arr.ctypes.data_as(ct))
            elif ct is C.c_float:
                setattr(c, name, float(getattr(self, name)))
            elif name in ("custom_section", "rx_section", "rx_kin_section"):
                setattr(c, name, (C.c_uint32 * 3)(*[int(x) for x in getattr(self, name)]))
            elif name == "rx_current_index":
                setattr(c, name, (C.c_int32 * 3)(*[int(x) for x in self.rx_current_index]))
            else:
                setattr(c, name, int(getattr(self, name)))
        return c

    def _sync_back(self, c):
        self.clock = int(c.clock)

    def inputs(self, q0=None, q1=None):
        c = self._cnet()
        if q0 is None:
            lib().snn_o_inputs(C.byref(c))
        else:
            lib().snn_o_inputs_range(C.byref(c), q0, q1)

    def inputs_tiled(self, q0, q1, block=1024):
        """step 1 for [q0, q1) with the all-core tiling of bench.py's cpu_baseline (same results bit for bit)"""
        c = self._cnet()
        lib().snn_o_inputs_tiled(C.byref(c), q0, q1, block)

    def _csr_args(self, row_ptr, pre, w):
        assert row_ptr.dtype == np.uint64 and pre.dtype == np.uint32 and w.dtype == np.float32
        assert row_ptr.size == self.n_neurons + 1 and pre.size == w.size == int(row_ptr[-1])
        return (row_ptr.ctypes.data_as(C.POINTER(C.c_uint64)), pre.ctypes.data_as(u32p), w.ctypes.data_as(f32p))

    def inputs_csr(self, row_ptr, pre, w, q0=0, q1=None):
        """step 1 over a sparse graph (CSR by post, ascending presynaptic indices per row)"""
        c = self._cnet()
        lib().snn_o_inputs_csr(C.byref(c), *self._csr_args(row_ptr, pre, w), q0, self.n_neurons if q1 is None else q1)

    def run_csr(self, row_ptr, pre, w, iterations, voltage_history=False, spike_history=False):
        it = int(iterations)
        self.voltage_history = np.zeros((it, self.n_neurons), np.float32) if voltage_history else None
        self.spike_history = np.zeros((it, self.n_neurons), np.uint8) if spike_history else None
        self.st_voltage_history = None
        c = self._cnet()
        lib().snn_o_run_csr(C.byref(c), *self._csr_args(row_ptr, pre, w), it)
        self._sync_back(c)
        return self

    def update_neurons(self, q0=None, q1=None):
        c = self._cnet()
        if q0 is None:
            lib().snn_o_update_neurons(C.byref(c))
        else:
            lib().snn_o_update_neurons_range(C.byref(c), q0, q1)

    def plasticity(self, c0=None, c1=None):
        c = self._cnet()
        if c0 is None:
            lib().snn_o_plasticity(C.byref(c))
        else:
            lib().snn_o_plasticity_cols(C.byref(c), c0, c1)

    def apply_reward(self, reward):
        c = self._cnet()
        lib().snn_o_apply_reward(C.byref(c), C.c_float(reward))

    def reward_modulation(self, c0=None, c1=None):
        c = self._cnet()
        if c0 is None:
            lib().snn_o_reward_modulation(C.byref(c))
        else:
            lib().snn_o_reward_modulation_cols(C.byref(c), c0, c1)

    def spike_trains(self):
        c = self._cnet()
        lib().snn_o_spike_trains(C.byref(c))

    def reward_cross_check(self):
        """0 when every connection of kind 1 / 2 lies where the reference defines its update (snn_o_reward_cross_check)"""
        return int(lib().snn_o_reward_cross_check(C.byref(self._cnet())))

    def run(self, iterations, voltage_history=False, spike_history=False, st_voltage_history=False,
            summaries=False, spike_counts=False, rewards=None):
        it = int(iterations)
        self.rewards = None if rewards is None else np.ascontiguousarray(rewards, np.float32)
        assert self.rewards is None or self.rewards.size == it
        self.avg_history = np.zeros((it, self.n_lattices), np.float32) if summaries else None
        self.eeg_history = np.zeros((it, self.n_lattices), np.float32) if summaries else None
        if spike_counts and self.spike_counts is None:
            self.spike_counts = np.zeros(self.n_neurons, np.uint32)
        if not spike_counts:
            self.spike_counts = None
        self.voltage_history = np.zeros((it, self.n_neurons), np.float32) if voltage_history else None
        self.spike_history = np.zeros((it, self.n_neurons), np.uint8) if spike_history else None
        self.st_voltage_history = np.zeros((it, self.n_cells), np.float32) if st_voltage_history else None
        c = self._cnet()
        lib().snn_o_run(C.byref(c), it)
        self._sync_back(c)
        return self


def usable_cpus(limit=64):
    """threads worth starting: the visible CPUs capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, min(n, limit))


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16 << 30


def expf(x):
    return lib().snn_o_expf_export(float(np.float32(x)))


def tanhf(x):
    return lib().snn_o_tanhf_export(float(np.float32(x)))


def sinhf(x):
    return lib().snn_o_sinhf_export(float(np.float32(x)))


def coshf(x):
    return lib().snn_o_coshf_export(float(np.float32(x)))


def sinf(x):
    return lib().snn_o_sinf_export(float(np.float32(x)))


def cosf(x):
    return lib().snn_o_cosf_export(float(np.float32(x)))


def tanf(x):
    return lib().snn_o_tanf_export(float(np.float32(x)))


def powif(x, n):
    return lib().snn_o_powif_export(float(np.float32(x)), int(n))


def powf(x, y):
    return lib().snn_o_powf_export(float(np.float32(x)), float(np.float32(y)))


def math_bits(which, first, count, stride=1, y=0.0):
    """f(float with bit pattern first + i * stride), i < count: which = 0 expf, 1 powf(x, 3), 2 powf(x, 4), 3 powf(x, y)"""
    out = np.empty(count, np.float32)
    lib().snn_o_math_bits(which, first & 0xFFFFFFFF, stride, count, float(y), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def uniform(seed, index, lo, hi):
    return lib().snn_o_uniform(seed, index, lo, hi)


def uniform_array(seed, n, lo, hi, offset=0):
    """Vectorised numpy twin of snn_o_uniform for indices offset..offset+n-1."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    h = (x >> np.uint64(32)).astype(np.uint32)
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)

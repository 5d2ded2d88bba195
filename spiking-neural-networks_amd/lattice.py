"""Lixirnet-style classes over the HIP stepper.

Same class / method names and argument meaning as the reference's Python interface
(interface_gpu/lixirnet/src/lattices/mod.rs:314-568 `impl_lattice_gpu!`, :1450-2117 `impl_network_gpu!`;
class list interface_gpu/lixirnet/src/lib.rs:463-482) for the containers on the hot path, with the
backend crate's own model types (backend/src/neuron/integrate_and_fire/mod.rs, hodgkin_huxley/mod.rs,
iterate_and_spike/mod.rs, spike_train/mod.rs, plasticity/mod.rs).

The host-side containers (`*Lattice`, `*Network`) only BUILD a lattice (populate / connect / apply ...);
stepping exists on the `*GPU` classes alone -- there is no CPU stepper in this package.
"""
import copy
import enum

import numpy as np

from .network import (CUSTOM, NT_CUSTOM, RC_CUSTOM, REFRACTORINESS_CUSTOM, ST_CUSTOM, NT_DISCRETE_SPIKE, NT_EXPONENTIAL_DECAY, RC_EXPONENTIAL_DECAY, ST_PRESET, BCM_IZHIKEVICH, ST_BCM_POISSON,
                      DeviceNetwork, HODGKIN_HUXLEY, IZHIKEVICH, LIF, QUADRATIC_INTEGRATE_AND_FIRE, SIMPLE_LIF,
                      ADAPTIVE_LIF, ADAPTIVE_EXP_LIF, LEAKY_IZHIKEVICH,
                      NT_APPROXIMATE, NT_DESTEXHE,
                      RC_APPROXIMATE, RC_DESTEXHE, ST_NONE, ST_POISSON, ST_RATE)


class IonotropicNeurotransmitterType(enum.IntEnum):      # iterate_and_spike/mod.rs:1068-1073
    AMPA = 0
    NMDA = 1
    GABA = 2


class _Record:
    _defaults = {}

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, copy.deepcopy(v))
        for k, v in kw.items():
            if k not in self._defaults:
                raise AttributeError(f"{type(self).__name__} has no field {k}")
            setattr(self, k, v)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={getattr(self, k)!r}' for k in self._defaults)})"


class ApproximateNeurotransmitter(_Record):              # iterate_and_spike/mod.rs:161-182
    _defaults = dict(t_max=1.0, t=0.0, clearance_constant=0.01)
    kinetics = NT_APPROXIMATE


class DestexheNeurotransmitter(_Record):                 # iterate_and_spike/mod.rs:122-145
    _defaults = dict(t_max=1.0, t=0.0, v_p=2.0, k_p=5.0)
    kinetics = NT_DESTEXHE


class DiscreteSpikeNeurotransmitter(_Record):            # iterate_and_spike/mod.rs:287-317
    _defaults = dict(t_max=1.0, t=0.0)
    kinetics = NT_DISCRETE_SPIKE


class ExponentialDecayNeurotransmitter(_Record):         # iterate_and_spike/mod.rs:323-366
    _defaults = dict(t_max=1.0, t=0.0, decay_constant=2.0)
    kinetics = NT_EXPONENTIAL_DECAY


class ApproximateReceptor(_Record):                      # iterate_and_spike/mod.rs:427-446
    _defaults = dict(r=0.0)
    kinetics = RC_APPROXIMATE


class DestexheReceptor(_Record):                         # iterate_and_spike/mod.rs:394-425
    _defaults = dict(r=0.0, alpha=1.0, beta=1.0)
    kinetics = RC_DESTEXHE


class ExponentialDecayReceptor(_Record):                 # iterate_and_spike/mod.rs:497-533
    _defaults = dict(r_max=1.0, r=0.0, decay_constant=2.0)
    kinetics = RC_EXPONENTIAL_DECAY


class AMPAReceptor(_Record):                             # iterate_and_spike/mod.rs:1078-1094
    _defaults = dict(current=0.0, g=1.0, e=0.0, r=ApproximateReceptor())
    type = IonotropicNeurotransmitterType.AMPA


class NMDAReceptor(_Record):                             # iterate_and_spike/mod.rs:1107-1125
    _defaults = dict(current=0.0, g=0.6, mg=0.3, e=0.0, r=ApproximateReceptor())
    type = IonotropicNeurotransmitterType.NMDA


class GABAReceptor(_Record):                             # iterate_and_spike/mod.rs:1140-1157
    _defaults = dict(current=0.0, g=1.2, e=-80.0, r=ApproximateReceptor())
    type = IonotropicNeurotransmitterType.GABA


class Ionotropic(dict):                                  # iterate_and_spike/mod.rs:1177-1257
    def insert(self, neurotransmitter_type, receptor):
        if IonotropicNeurotransmitterType(neurotransmitter_type) != receptor.type:
            raise ValueError("ReceptorNeurotransmitterError::MismatchedTypes")
        self[IonotropicNeurotransmitterType(neurotransmitter_type)] = receptor


class STDP(_Record):                                     # plasticity/mod.rs:16-39
    _defaults = dict(a_plus=2.0, a_minus=2.0, tau_plus=4.5, tau_minus=4.5, dt=0.1)


class TraceRSTDP(_Record):                               # plasticity/mod.rs:126-146
    _defaults = dict(counter=0, dw=0.0, weight=0.0, c=0.0)


class RewardModulatedSTDP(_Record):                      # plasticity/mod.rs:158-190
    _defaults = dict(dopamine=0.0, tau_d=20.0, tau_c=0.0001, a_plus=2.0, a_minus=2.0, tau_plus=4.5, tau_minus=4.5, dt=0.1)


class BCM(_Record):                                      # plasticity/mod.rs:80-94
    _defaults = dict(decay=0.1, average_scalar=0.1, dt=0.1)


class GraphPosition:                                     # graph/mod.rs:24-30
    def __init__(self, id, pos):
        self.id, self.pos = id, tuple(pos)

    def __eq__(self, o):
        return (self.id, self.pos) == (o.id, o.pos)

    def __hash__(self):
        return hash((self.id, self.pos))


class ConnectingWeights(np.ndarray):
    """The connecting graph's weight matrix as the reference hands it out (`connecting_weights`: rows / columns in
    `connecting_position_to_index` order, absent edges 0), which can also be asked by position:
    m[GraphPosition(pre), GraphPosition(post)]."""

    def __new__(cls, matrix, index):
        obj = np.asarray(matrix, np.float32).view(cls)
        obj.index = index
        return obj

    def __array_finalize__(self, obj):
        self.index = getattr(obj, "index", {})

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 2 and all(isinstance(k, GraphPosition) for k in key):
            key = (self.index[key[0]], self.index[key[1]])
        return super().__getitem__(key)


class _Neuron(_Record):
    _common = dict(is_spiking=False, last_firing_time=None)
    model = None
    abi_names = {}          # python field -> C-ABI attribute (when they differ)

    receptors_type = None       # the receptor container of the class (Ionotropic unless the model brings its own set)
    receptor_set = None         # modelgen.ReceptorsModel of a generated neuron with its own [receptors]

    def __init__(self, **kw):
        self.synaptic_neurotransmitters = {}
        self.receptors = (self.receptors_type or Ionotropic)()
        super().__init__(**kw)

    def set_synaptic_neurotransmitters(self, d):
        self.synaptic_neurotransmitters = {IonotropicNeurotransmitterType(int(k)): v for k, v in d.items()}

    def set_receptors(self, r):
        self.receptors = r


class IzhikevichNeuron(_Neuron):                         # integrate_and_fire/mod.rs:1159-1220
    model = IZHIKEVICH
    _defaults = dict(current_voltage=-65.0, v_th=30.0, v_init=-65.0, a=0.02, b=0.2, c=-55.0, d=8.0, w_value=30.0,
                     w_init=30.0, gap_conductance=7.0, tau_m=1.0, c_m=100.0, dt=0.1, **_Neuron._common)
    state_fields = ("w_value", "a", "b", "c", "d", "tau_m")


class LeakyIntegrateAndFireNeuron(_Neuron):              # integrate_and_fire/mod.rs:108-171
    model = LIF
    _defaults = dict(current_voltage=-75.0, v_th=-55.0, v_reset=-75.0, v_init=-75.0, refractory_count=0.0, tref=10.0,
                     leak_constant=-1.0, integration_constant=1.0, gap_conductance=7.0, e_l=-75.0, g_l=10.0,
                     tau_m=10.0, c_m=100.0, dt=0.1, **_Neuron._common)
    state_fields = ("v_reset", "refractory_count", "tref", "leak_constant", "integration_constant", "e_l", "g_l", "tau_m")


class QuadraticIntegrateAndFireNeuron(_Neuron):          # integrate_and_fire/mod.rs:259-322
    model = QUADRATIC_INTEGRATE_AND_FIRE
    _defaults = dict(current_voltage=-75.0, v_th=-55.0, v_reset=-75.0, v_init=-75.0, refractory_count=0.0, tref=10.0,
                     alpha=1.0, v_c=-60.0, integration_constant=1.0, gap_conductance=7.0, tau_m=100.0, c_m=100.0,
                     dt=0.1, **_Neuron._common)
    state_fields = ("v_reset", "refractory_count", "tref", "alpha", "v_c", "integration_constant", "tau_m")


class SimpleLeakyIntegrateAndFire(_Neuron):              # integrate_and_fire/mod.rs:1523-1570
    model = SIMPLE_LIF
    _defaults = dict(current_voltage=-75.0, g=-0.1, e=0.0, v_th=-55.0, v_reset=-75.0, v_init=-75.0,
                     gap_conductance=10.0, c_m=100.0, dt=0.1, **_Neuron._common)
    state_fields = ("g", "e", "v_reset")


class AdaptiveLeakyIntegrateAndFireNeuron(_Neuron):      # integrate_and_fire/mod.rs:918-996
    model = ADAPTIVE_LIF
    _defaults = dict(current_voltage=-75.0, v_th=-55.0, v_reset=-75.0, v_init=-75.0, refractory_count=0.0, tref=10.0,
                     alpha=6.0, beta=10.0, w_value=0.0, w_init=0.0, leak_constant=-1.0, integration_constant=1.0,
                     gap_conductance=7.0, e_l=-75.0, g_l=10.0, tau_m=10.0, c_m=100.0, dt=0.1, **_Neuron._common)
    state_fields = ("v_reset", "refractory_count", "tref", "alpha", "beta", "w_value", "leak_constant",
                    "integration_constant", "e_l", "g_l", "tau_m")


class AdaptiveExpLeakyIntegrateAndFireNeuron(_Neuron):   # integrate_and_fire/mod.rs:1051-1130
    model = ADAPTIVE_EXP_LIF
    _defaults = dict(AdaptiveLeakyIntegrateAndFireNeuron._defaults, slope_factor=1.0)
    state_fields = AdaptiveLeakyIntegrateAndFireNeuron.state_fields + ("slope_factor",)


class LeakyIzhikevichNeuron(_Neuron):                    # integrate_and_fire/mod.rs:1270-1331
    model = LEAKY_IZHIKEVICH
    _defaults = dict(current_voltage=-65.0, v_th=30.0, v_init=-65.0, a=0.02, b=0.2, c=-55.0, d=8.0, w_value=30.0,
                     w_init=30.0, e_l=-65.0, gap_conductance=7.0, tau_m=10.0, c_m=100.0, dt=0.1, **_Neuron._common)
    state_fields = ("w_value", "a", "b", "c", "d", "tau_m", "e_l")


class BCMIzhikevichNeuron(_Neuron):                      # integrate_and_fire/mod.rs:1358-1436
    model = BCM_IZHIKEVICH
    _defaults = dict(IzhikevichNeuron._defaults, average_activity=0.0, current_activity=0.0, period=3, num_spikes=0,
                     firing_rate_clock=0.0, firing_rate_window=500.0)
    state_fields = IzhikevichNeuron.state_fields + ("average_activity", "current_activity", "firing_rate_clock",
                                                    "firing_rate_window")
    counter_fields = ("period", "num_spikes")


class HodgkinHuxleyNeuron(_Neuron):                      # hodgkin_huxley/mod.rs:49-98, ion_channels/mod.rs
    model = HODGKIN_HUXLEY
    _defaults = dict(current_voltage=-65.0, gap_conductance=7.0, dt=0.01, c_m=1.0, v_th=0.0,
                     m=0.0, h=0.0, n=0.0, g_na=120.0, e_na=50.0, g_k=36.0, e_k=-77.0, g_k_leak=0.3, e_k_leak=-55.0,
                     na_current=0.0, k_current=0.0, k_leak_current=0.0, was_increasing=False, **_Neuron._common)
    state_fields = ("m", "h", "n", "g_na", "e_na", "g_k", "e_k", "g_k_leak", "e_k_leak", "na_current", "k_current",
                    "k_leak_current")
    abi_names = dict(m="na_channel$m$state", h="na_channel$h$state", n="k_channel$n$state", g_na="na_channel$g_na",
                     e_na="na_channel$e_na", g_k="k_channel$g_k", e_k="k_channel$e_k",
                     g_k_leak="k_leak_channel$g_k_leak", e_k_leak="k_leak_channel$e_k_leak",
                     na_current="na_channel$current", k_current="k_channel$current",
                     k_leak_current="k_leak_channel$current")


def neuron_builder(text):
    """The reference's `neuron_builder!("[neuron] ...")` (build_test/nb_macro/src/lib.rs) for this façade: parse the
    description (ion channels included), compile its library (modelgen -> hipcc, cached under csrc/generated) and
    return (NeuronClass, LatticeClass, LatticeGPUClass) -- the neuron's fields are `current_voltage`, `dt`, `c_m`,
    `gap_conductance` and the description's variables under their DSL names (`getattr(n, "k$current")` for an ion
    channel field).  Receptors are the ionotropic AMPA / NMDA / GABA set.  See description_builder for texts that also
    hold a spike train or a refractoriness."""
    from . import modelgen
    modelgen.parse(text)                      # a neuron and nothing else
    out = description_builder(text)
    return out.Neuron, out.Lattice, out.LatticeGPU


def neuron_builder_from_file(path):
    """`neuron_builder_from_file!("model.nb")` (build_test/nb_macro/src/lib.rs:9360): neuron_builder on a file's text"""
    with open(path) as f:
        return neuron_builder(f.read())


def description_builder_from_file(path):
    with open(path) as f:
        return description_builder(f.read())


class GeneratedReceptors(dict):
    """The receptor set of a generated neuron (a `[receptors]` block): receptors keyed by the set's own
    neurotransmitter type (slot 0.. of the exchange), the block's top-level variables as attributes."""
    model = None
    NeurotransmitterType = None

    def __init__(self):
        super().__init__()
        for name, default in self.model.variables:
            if "$" not in name:
                setattr(self, name, bool(default) if name in self.model.bools else default)

    def insert(self, neurotransmitter_type, receptor):
        if self.NeurotransmitterType(neurotransmitter_type) != receptor.type:
            raise ValueError("ReceptorNeurotransmitterError::MismatchedTypes")
        self[self.NeurotransmitterType(neurotransmitter_type)] = receptor


class GeneratedDescription:
    """What `description_builder` returns: the classes of the blocks the description has (None for the others), all
    living in ONE compiled library (`library`)."""
    Neuron = Lattice = LatticeGPU = SpikeTrain = SpikeTrainLattice = Refractoriness = None
    Neurotransmitter = ReceptorKinetics = Receptors = NeurotransmitterType = None
    receptor_types = None                       # {neurotransmitter name: receptor record class} of a generated receptor set
    library = description = None


def description_builder(text):
    """`neuron_builder!` over a text with several blocks: a [neuron] (with its [ion_channel]s), a [spike_train] and a
    [neural_refractoriness], a [neurotransmitter_kinetics] and a [receptor_kinetics] become façade classes that share
    one compiled library, so that they can meet in one LatticeNetwork (the kinetics classes stand where
    ApproximateNeurotransmitter / ApproximateReceptor do: `AMPAReceptor(r=g.ReceptorKinetics())`).  A [receptors] block
    gives the neuron its own container `g.Receptors` (top-level variables as attributes) with one record class per
    neurotransmitter in `g.receptor_types`, keyed by `g.NeurotransmitterType`."""
    from . import _lib, modelgen
    desc = modelgen.parse_description(text)
    out = GeneratedDescription()
    out.description, out.library = desc, _lib.build_custom(desc)
    receptors_type = None
    if desc.receptors is not None:
        rx = desc.receptors
        out.NeurotransmitterType = enum.IntEnum(rx.name + "NeurotransmitterType", {t[0]: k for k, t in enumerate(rx.types)})
        out.receptor_types = {}
        multi = getattr(rx, "multi", False)
        state_kinetics = ApproximateReceptor         # the kinetics value of every receptor state (`kinetics:` of the block)
        if multi and desc.receptor_kinetics is not None:
            rk = desc.receptor_kinetics
            kf = {k: (bool(v) if k in rk.bools else v) for k, v in rk.variables}
            state_kinetics = type(rk.name, (_Record,), dict(kinetics=RC_CUSTOM, description=rk, lib_path=out.library,
                                                            _defaults={"r": 0.0, **kf}, state_fields=tuple(kf)))
        for k, (nt, _, _) in enumerate(rx.types):
            own = {n.split("$", 1)[1]: (bool(d) if n in rx.bools else d) for n, d in rx.variables
                   if n.startswith(nt + "$") and "$kinetics$" not in n}
            if multi:      # several states per type: each a kinetics value under its name (GlutamateReceptor.ampa_r.r ...)
                states = tuple(rx.states[k])
                out.receptor_types[nt] = type(nt + "Receptor", (_Record,), dict(
                    type=out.NeurotransmitterType(k), state_fields=tuple(own), state_names=states,
                    _defaults=dict(own, **{st: state_kinetics() for st in states})))
            else:
                out.receptor_types[nt] = type(nt + "Receptor", (_Record,), dict(
                    type=out.NeurotransmitterType(k), state_fields=tuple(own), _defaults=dict(own, r=ApproximateReceptor())))
        receptors_type = out.Receptors = type(rx.name, (GeneratedReceptors,), dict(
            model=rx, NeurotransmitterType=out.NeurotransmitterType))
    if desc.neuron is not None:
        model = desc.neuron
        fields = {k: (bool(v) if k in model.bools else v) for k, v in model.variables}
        defaults = dict(current_voltage=model.mandatory["current_voltage"], dt=model.mandatory["dt"],
                        c_m=model.mandatory["c_m"], gap_conductance=model.mandatory["gap_conductance"], **_Neuron._common)
        defaults.update(fields)
        defaults.setdefault("v_th", 0.0)        # the common attribute exists on the device whether the model reads it or not
        out.Neuron = type(model.name, (_Neuron,), dict(model=CUSTOM, _defaults=defaults, state_fields=tuple(fields),
                                                       lib_path=out.library, description=model,
                                                       receptors_type=receptors_type, receptor_set=desc.receptors,
                                                       __doc__=f"generated from a neuron description ({model.name})"))
        out.Lattice = type(model.name + "Lattice", (Lattice,), dict(neuron_type=out.Neuron))
        out.LatticeGPU = type(model.name + "LatticeGPU", (LatticeGPU,), dict(lattice_type=out.Lattice))
    if desc.spike_train is not None:
        st = desc.spike_train
        fields = {k: (bool(v) if k in st.bools else v) for k, v in st.variables}
        defaults = dict(current_voltage=st.mandatory["current_voltage"], v_th=st.mandatory["v_th"],
                        v_resting=st.mandatory["v_resting"], dt=st.mandatory["dt"], is_spiking=False,
                        last_firing_time=None, k=10000.0)
        defaults.update(fields)
        out.SpikeTrain = type(st.name, (_SpikeTrain,), dict(kind=ST_CUSTOM, _defaults=defaults, state_fields=tuple(fields),
                                                            lib_path=out.library, description=st))
        out.SpikeTrainLattice = type(st.name + "Lattice", (SpikeTrainLattice,), dict(spike_train_type=out.SpikeTrain))
    if desc.refractoriness is not None:
        rf = desc.refractoriness
        out.Refractoriness = type(rf.name, (_Record,), dict(kind=REFRACTORINESS_CUSTOM, description=rf,
                                                            _defaults=dict(k=rf.decay, **dict(rf.variables)),
                                                            state_fields=tuple(n for n, _ in rf.variables)))
    if desc.receptors is not None and getattr(desc.receptors, "multi", False):
        out.ReceptorKinetics = state_kinetics
    for attr, model, selector in (("Neurotransmitter", desc.nt_kinetics, NT_CUSTOM),
                                  ("ReceptorKinetics", desc.receptor_kinetics, RC_CUSTOM)):
        if model is not None and getattr(out, attr) is None:
            fields = {k: (bool(v) if k in model.bools else v) for k, v in model.variables}
            setattr(out, attr, type(model.name, (_Record,), dict(kinetics=selector, description=model, lib_path=out.library,
                                                                 _defaults={model.state: 0.0, **fields},
                                                                 state_fields=tuple(fields))))
    return out


class DeltaDiracRefractoriness(_Record):                 # spike_train/mod.rs:79-88
    _defaults = dict(k=10000.0)
    kind = 0


class ExponentialDecayRefractoriness(_Record):           # spike_train/mod.rs:164-178
    _defaults = dict(k=10000.0)
    kind = 1


class _SpikeTrain(_Record):
    kind = ST_NONE
    neural_refractoriness = None            # a DeltaDiracRefractoriness / ExponentialDecayRefractoriness; None: the `k` field, delta dirac

    def __init__(self, **kw):
        self.synaptic_neurotransmitters = {}
        super().__init__(**kw)

    def set_synaptic_neurotransmitters(self, d):
        self.synaptic_neurotransmitters = {IonotropicNeurotransmitterType(k): v for k, v in d.items()}


class PoissonNeuron(_SpikeTrain):                        # spike_train/mod.rs:259-313 (+ GPU generator :380-435)
    kind = ST_POISSON
    _defaults = dict(current_voltage=0.0, v_th=30.0, v_resting=0.0, is_spiking=False, last_firing_time=None,
                     chance_of_firing=0.0, dt=0.1, k=10000.0, seed=1)

    @classmethod
    def from_firing_rate(cls, hertz, dt):                # spike_train/mod.rs:327-334
        return cls(dt=dt, chance_of_firing=1.0 / ((1000.0 / dt) / hertz))


class BCMPoissonNeuron(_SpikeTrain):                     # spike_train/mod.rs:835-884
    kind = ST_BCM_POISSON
    _defaults = dict(PoissonNeuron._defaults, average_activity=0.0, current_activity=0.0, period=3, num_spikes=0,
                     firing_rate_clock=0.0, firing_rate_window=500.0)

    @classmethod
    def from_firing_rate(cls, hertz, dt):                # spike_train/mod.rs:903-910
        return cls(dt=dt, chance_of_firing=1.0 / ((1000.0 / dt) / hertz))


class RateSpikeTrain(_SpikeTrain):                       # spike_train/mod.rs:975-1013
    kind = ST_RATE
    _defaults = dict(current_voltage=0.0, v_th=30.0, v_resting=0.0, rate=0.0, step=0.0, is_spiking=False,
                     last_firing_time=None, dt=0.1, k=10000.0)


class PresetSpikeTrain(_SpikeTrain):                     # spike_train/mod.rs:753-800
    kind = ST_PRESET
    _defaults = dict(current_voltage=0.0, v_th=30.0, v_resting=0.0, is_spiking=False, last_firing_time=None,
                     firing_times=(), internal_clock=0.0, counter=0, dt=0.1, k=10000.0)


# ---------------------------------------------------------------------------------------------------
# host-side containers
# ---------------------------------------------------------------------------------------------------
class Lattice:
    """Lattice<T, AdjacencyMatrix, GridVoltageHistory, STDP> builder (backend/src/neuron/mod.rs:556-1157)."""
    neuron_type = IzhikevichNeuron

    def __init__(self, id=0):
        self.id = id
        self.cell_grid = []
        self.weights = np.zeros((0, 0), np.float32)            # AdjacencyMatrix, index = row*cols + col
        self.connections = np.zeros((0, 0), np.uint32)
        self.update_grid_history = False
        self.update_graph_history = False
        self.electrical_synapse = True
        self.chemical_synapse = False
        self.do_plasticity = False
        self.plasticity = STDP()
        self.internal_clock = 0
        self.parallel = False                                  # neuron/mod.rs:582 (the device path has no serial form)
        self.history = np.zeros((0, 0, 0), np.float32)         # GridVoltageHistory: [steps][rows][cols], filled by the GPU classes
        self.weights_history = np.zeros((0, 0, 0), np.float32)  # AdjacencyMatrix::history when update_graph_history

    # -- shape ------------------------------------------------------------------------------------
    @property
    def rows(self):
        return len(self.cell_grid)

    @property
    def cols(self):
        return len(self.cell_grid[0]) if self.cell_grid else 0

    def _index(self, pos):
        r, c = pos
        if not (0 <= r < self.rows and 0 <= c < self.cols):
            raise KeyError(f"GraphError::PositionNotFound({pos})")
        return r * self.cols + c

    @property
    def position_to_index(self):
        return {(r, c): r * self.cols + c for r in range(self.rows) for c in range(self.cols)}

    def get_every_node(self):
        return set(self.position_to_index)

    # -- building ---------------------------------------------------------------------------------
    def populate(self, neuron, num_rows, num_cols):             # neuron/mod.rs:1105-1126
        self.cell_grid = [[copy.deepcopy(neuron) for _ in range(num_cols)] for _ in range(num_rows)]
        n = num_rows * num_cols
        self.weights = np.zeros((n, n), np.float32)
        self.connections = np.zeros((n, n), np.uint32)

    def connect(self, connection_conditional, weight_logic=None):   # neuron/mod.rs:1134-1157
        pos = [(r, c) for r in range(self.rows) for c in range(self.cols)]
        for i, a in enumerate(pos):
            for j, b in enumerate(pos):
                if connection_conditional(a, b):
                    self.connections[i, j] = 1
                    self.weights[i, j] = 1.0 if weight_logic is None else weight_logic(a, b)
                else:
                    self.connections[i, j] = 0
                    self.weights[i, j] = 0.0

    def apply(self, function):
        for row in self.cell_grid:
            for n in row:
                function(n)

    def apply_given_position(self, function):
        for r, row in enumerate(self.cell_grid):
            for c, n in enumerate(row):
                function((r, c), n)

    def get_neuron(self, row, col):
        self._index((row, col))
        return copy.deepcopy(self.cell_grid[row][col])

    def set_neuron(self, row, col, neuron):
        self._index((row, col))
        self.cell_grid[row][col] = copy.deepcopy(neuron)

    def get_weight(self, presynaptic, postsynaptic):
        """lookup_weight(...).unwrap_or(0.) as the reference's Python class hands it out (interface
        lattices/mod.rs:114-121): an absent edge reads 0, a position outside the lattice is a KeyError"""
        i, j = self._index(presynaptic), self._index(postsynaptic)
        return float(self.weights[i, j]) if self.connections[i, j] else 0.0

    def get_incoming_connections(self, position):
        j = self._index(position)
        return {(i // self.cols, i % self.cols) for i in np.nonzero(self.connections[:, j])[0]}

    def get_outgoing_connections(self, position):
        i = self._index(position)
        return {(j // self.cols, j % self.cols) for j in np.nonzero(self.connections[i])[0]}

    def set_dt(self, dt):                                       # neuron/mod.rs:649-652
        self.apply(lambda n: setattr(n, "dt", dt))
        self.plasticity.dt = dt

    def reset_timing(self):                                     # neuron/mod.rs:405-420
        self.internal_clock = 0
        self.apply(lambda n: setattr(n, "last_firing_time", None))

    def reset_history(self):                                    # neuron/mod.rs:398-403
        self.history = np.zeros((0, self.rows, self.cols), np.float32)
        self.weights_history = np.zeros((0, self.rows * self.cols, self.rows * self.cols), np.float32)

    def get_weights(self):                                      # interface lattices/mod.rs:275-285: None -> 0
        return np.where(self.connections != 0, self.weights, np.float32(0)).astype(np.float32)

    def get_position_to_index_for_weights(self):
        return self.position_to_index

    def run_lattice(self, iterations):
        raise NotImplementedError("this package steps lattices on the GPU only: use "
                                  f"{type(self).__name__}GPU.from_lattice(lattice).run_lattice(iterations)")


class RewardModulatedLattice(Lattice):
    """RewardModulatedLattice<TraceRSTDP, T, AdjacencyMatrix, GridVoltageHistory, RewardModulatedSTDP> builder
    (backend/src/neuron/mod.rs:2719-3417): every edge carries a TraceRSTDP, `weights` holds TraceRSTDP::weight and
    `traces` TraceRSTDP::c."""

    def __init__(self, id=0):
        super().__init__(id)
        self.traces = np.zeros((0, 0), np.float32)
        self.do_modulation = True
        self.reward_modulator = RewardModulatedSTDP()

    def populate(self, neuron, num_rows, num_cols):
        super().populate(neuron, num_rows, num_cols)
        self.traces = np.zeros_like(self.weights)

    def connect(self, connection_conditional, weight_logic=None):      # neuron/mod.rs:3301-3321
        pos = [(r, c) for r in range(self.rows) for c in range(self.cols)]
        for i, a in enumerate(pos):
            for j, b in enumerate(pos):
                on = bool(connection_conditional(a, b))
                t = (weight_logic(a, b) if weight_logic is not None else TraceRSTDP(weight=1.0)) if on else TraceRSTDP()
                if not isinstance(t, TraceRSTDP):
                    t = TraceRSTDP(weight=float(t))
                if t.counter != 0 or t.dw != 0.0:
                    raise ValueError("a TraceRSTDP enters the stepper between steps: counter and dw must be 0")
                self.connections[i, j] = int(on)
                self.weights[i, j], self.traces[i, j] = t.weight, t.c

    def get_weight(self, presynaptic, postsynaptic):
        i, j = self._index(presynaptic), self._index(postsynaptic)
        if not self.connections[i, j]:
            raise KeyError("no connection")
        return TraceRSTDP(weight=float(self.weights[i, j]), c=float(self.traces[i, j]))

    def set_dt(self, dt):
        self.apply(lambda n: setattr(n, "dt", dt))
        self.reward_modulator.dt = dt


class SpikeTrainLattice:
    """SpikeTrainLattice builder (backend/src/neuron/mod.rs:1292-1436)."""
    spike_train_type = RateSpikeTrain

    def __init__(self, id=0):
        self.id = id
        self.cell_grid = []
        self.update_grid_history = False
        self.internal_clock = 0
        self.history = np.zeros((0, 0, 0), np.float32)         # SpikeTrainGridHistory: [steps][rows][cols]

    rows = Lattice.rows
    cols = Lattice.cols

    def reset_timing(self):                                     # neuron/mod.rs:1344-1356
        self.internal_clock = 0
        self.apply(lambda n: setattr(n, "last_firing_time", None))

    def reset_history(self):
        self.history = np.zeros((0, self.rows, self.cols), np.float32)

    def populate(self, spike_train, num_rows, num_cols):
        self.cell_grid = [[copy.deepcopy(spike_train) for _ in range(num_cols)] for _ in range(num_rows)]

    apply = Lattice.apply
    apply_given_position = Lattice.apply_given_position

    def get_neuron(self, row, col):
        return copy.deepcopy(self.cell_grid[row][col])

    def set_neuron(self, row, col, neuron):
        self.cell_grid[row][col] = copy.deepcopy(neuron)

    def set_dt(self, dt):
        self.apply(lambda n: setattr(n, "dt", dt))


class LatticeNetwork:
    """LatticeNetwork builder (backend/src/neuron/mod.rs:1538-2075)."""

    def __init__(self):
        self.lattices, self.spike_train_lattices = {}, {}
        self.connecting = {}              # (GraphPosition pre, GraphPosition post) -> weight
        self.connecting_nodes = []        # nodes of the connecting graph in insertion order (its matrix index)
        self.electrical_synapse, self.chemical_synapse = True, False
        self.internal_clock = 0
        self.parallel = False
        self.update_connecting_graph_history = False
        self.connecting_graph_history = []

    def clear(self):                                            # neuron/mod.rs:1681-1686
        self.lattices, self.spike_train_lattices = {}, {}
        self.connecting, self.connecting_nodes = {}, []

    @classmethod
    def generate_network(cls, lattices=(), spike_train_lattices=()):
        net = cls()
        for l in lattices:
            net.add_lattice(l)
        for l in spike_train_lattices:
            net.add_spike_train_lattice(l)
        return net

    def get_all_ids(self):
        return set(self.lattices) | set(self.spike_train_lattices)

    def add_lattice(self, lattice):                             # neuron/mod.rs:1663-1679
        if lattice.id in self.get_all_ids():
            raise KeyError(f"LatticeNetworkError::GraphIDAlreadyPresent({lattice.id})")
        self.lattices[lattice.id] = lattice

    def add_spike_train_lattice(self, lattice):
        if lattice.id in self.get_all_ids():
            raise KeyError(f"LatticeNetworkError::GraphIDAlreadyPresent({lattice.id})")
        self.spike_train_lattices[lattice.id] = lattice

    def get_lattice(self, id):
        return self.lattices[id]

    def get_spike_train_lattice(self, id):
        return self.spike_train_lattices[id]

    def connect_internally(self, id, connection_conditional, weight_logic=None):
        self.lattices[id].connect(connection_conditional, weight_logic)

    def connect(self, presynaptic_id, postsynaptic_id, connection_conditional, weight_logic=None):   # mod.rs:1845-1935
        if postsynaptic_id in self.spike_train_lattices:
            raise KeyError("LatticeNetworkError::PostsynapticLatticeCannotBeSpikeTrain")
        if postsynaptic_id not in self.lattices:
            raise KeyError(f"LatticeNetworkError::PostsynapticIDNotFound({postsynaptic_id})")
        if presynaptic_id not in self.get_all_ids():
            raise KeyError(f"LatticeNetworkError::PresynapticIDNotFound({presynaptic_id})")
        if presynaptic_id == postsynaptic_id:
            return self.connect_internally(presynaptic_id, connection_conditional, weight_logic)
        pre = self.lattices.get(presynaptic_id) or self.spike_train_lattices[presynaptic_id]
        post = self.lattices[postsynaptic_id]
        for a in ((r, c) for r in range(pre.rows) for c in range(pre.cols)):
            for b in ((r, c) for r in range(post.rows) for c in range(post.cols)):
                key = (GraphPosition(presynaptic_id, a), GraphPosition(postsynaptic_id, b))
                for node in key:                                 # add_node: both ends join the connecting graph
                    if node not in self._node_set():
                        self.connecting_nodes.append(node)
                        self._nodes.add(node)
                if connection_conditional(a, b):
                    self.connecting[key] = 1.0 if weight_logic is None else float(weight_logic(a, b))
                else:
                    self.connecting.pop(key, None)

    def set_dt(self, dt):
        for l in list(self.lattices.values()) + list(self.spike_train_lattices.values()):
            l.set_dt(dt)

    def _node_set(self):
        if getattr(self, "_nodes", None) is None or len(self._nodes) != len(self.connecting_nodes):
            self._nodes = set(self.connecting_nodes)
        return self._nodes

    # -- the rest of impl_network! / impl_network_gpu! (interface lattices/mod.rs:697-1448, 1450-2117) -------------
    def _lattice(self, id):
        if id not in self.lattices:
            raise KeyError(f"Lattice {id} not found in network")
        return self.lattices[id]

    def _spike_train_lattice(self, id):
        if id not in self.spike_train_lattices:
            raise KeyError(f"Spike train lattice {id} not found in network")
        return self.spike_train_lattices[id]

    def _any(self, id):
        if id in self.lattices:
            return self.lattices[id]
        if id in self.spike_train_lattices:
            return self.spike_train_lattices[id]
        raise KeyError(f"Id {id} not found in network")

    @property
    def connecting_position_to_index(self):
        return {node: k for k, node in enumerate(self.connecting_nodes)}

    def get_connecting_position_to_index(self):
        return self.connecting_position_to_index

    def get_connecting_weights(self):
        """the connecting graph's matrix, absent edges 0 (interface lattices/mod.rs:893-903)"""
        index = self.connecting_position_to_index
        m = np.zeros((len(index), len(index)), np.float32)
        for (pre, post), w in self.connecting.items():
            m[index[pre], index[post]] = w
        return ConnectingWeights(m, index)

    @property
    def connecting_weights(self):
        return self.get_connecting_weights()

    def get_weight(self, presynaptic, postsynaptic):
        """lookup inside a lattice (same id) or in the connecting graph; an absent edge reads 0 (:914-940)"""
        if presynaptic.id == postsynaptic.id:
            l = self._lattice(presynaptic.id)
            i, j = l._index(presynaptic.pos), l._index(postsynaptic.pos)
            return float(l.weights[i, j]) if l.connections[i, j] else 0.0
        nodes = self._node_set()
        if presynaptic not in nodes or postsynaptic not in nodes:
            raise KeyError("GraphError::PositionNotFound")
        return float(self.connecting.get((presynaptic, postsynaptic), 0.0))

    def get_incoming_connections_within_lattice(self, id, position):
        return self._lattice(id).get_incoming_connections(position)

    def get_outgoing_connections_within_lattice(self, id, position):
        return self._lattice(id).get_outgoing_connections(position)

    def get_incoming_connectings_across_lattices(self, id, position):
        self._lattice(id)
        gp = GraphPosition(id, position)
        if gp not in self._node_set():
            raise KeyError(f"Position {position} not found in lattice")
        return {pre for (pre, post) in self.connecting if post == gp}

    def get_outgoing_connectings_across_lattices(self, id, position):
        self._lattice(id)
        gp = GraphPosition(id, position)
        if gp not in self._node_set():
            raise KeyError(f"Position {position} not found in lattice")
        return {post for (pre, post) in self.connecting if pre == gp}

    def get_neuron(self, id, row, col):
        return self._lattice(id).get_neuron(row, col)

    def set_neuron(self, id, row, col, neuron):
        self._lattice(id).set_neuron(row, col, neuron)

    def get_spike_train(self, id, row, col):
        l = self._spike_train_lattice(id)
        if not (0 <= row < l.rows and 0 <= col < l.cols):
            raise KeyError(f"Position ({row}, {col}) not found")
        return l.get_neuron(row, col)

    def set_spike_train(self, id, row, col, neuron):
        l = self._spike_train_lattice(id)
        if not (0 <= row < l.rows and 0 <= col < l.cols):
            raise KeyError(f"Position ({row}, {col}) not found")
        l.set_neuron(row, col, neuron)

    def set_lattice(self, id, lattice):
        """replace lattice `id` (the replacement takes that id, interface lattices/mod.rs:1132-1140)"""
        self._lattice(id)
        lattice.id = id
        self.lattices[id] = lattice

    def set_spike_train_lattice(self, id, lattice):
        self._spike_train_lattice(id)
        lattice.id = id
        self.spike_train_lattices[id] = lattice

    def get_do_plasticity(self, id):
        return self._lattice(id).do_plasticity

    def set_do_plasticity(self, id, flag):
        self._lattice(id).do_plasticity = bool(flag)

    def get_plasticity(self, id):
        return copy.deepcopy(self._lattice(id).plasticity)

    def set_plasticity(self, id, plasticity):
        self._lattice(id).plasticity = copy.deepcopy(plasticity)

    def reset_timing(self, id):
        self._any(id).reset_timing()

    def reset_history(self, id):
        self._any(id).reset_history()

    def get_update_grid_history(self, id):
        return self._any(id).update_grid_history

    def set_update_grid_history(self, id, flag):
        self._any(id).update_grid_history = bool(flag)

    def get_update_graph_history(self, id):
        return self._lattice(id).update_graph_history

    def set_update_graph_history(self, id, flag):
        self._lattice(id).update_graph_history = bool(flag)

    def apply_lattice(self, id, function):
        self._lattice(id).apply(function)

    def apply_spike_train_lattice(self, id, function):
        self._spike_train_lattice(id).apply(function)

    def apply_lattice_given_position(self, id, function):
        self._lattice(id).apply_given_position(function)

    def apply_spike_train_lattice_given_position(self, id, function):
        self._spike_train_lattice(id).apply_given_position(function)

    def run_lattices(self, iterations):
        raise NotImplementedError("this package steps networks on the GPU only: use "
                                  f"{type(self).__name__}GPU.from_network(network).run_lattices(iterations)")


# ---------------------------------------------------------------------------------------------------
# AoS <-> named SoA buffers (IterateAndSpikeGPU::convert_to_gpu / convert_to_cpu)
# ---------------------------------------------------------------------------------------------------
def _receptor_states(receptor):
    """the receptor-kinetics values of a receptor record: its `r`, or the named states of a generated receptor with
    several of them (`receptors: ampa_r, nmda_r`)"""
    names = getattr(type(receptor), "state_names", None)
    return [receptor.r] if names is None else [getattr(receptor, n) for n in names]


def _kinetics_of(cells, default_nt=NT_APPROXIMATE, default_rc=RC_APPROXIMATE):
    nt, rc = None, None
    for c in cells:
        for v in c.synaptic_neurotransmitters.values():
            nt = v.kinetics if nt is None else nt
            if v.kinetics != nt:
                raise TypeError("one neurotransmitter kinetics type per network (a type parameter in the reference)")
        for v in getattr(c, "receptors", {}).values():
            for k in _receptor_states(v):
                rc = k.kinetics if rc is None else rc
                if k.kinetics != rc:
                    raise TypeError("one receptor kinetics type per network (a type parameter in the reference)")
    return (default_nt if nt is None else nt), (default_rc if rc is None else rc)


def _lft(cells):
    return np.array([-1 if c.last_firing_time is None else int(c.last_firing_time) for c in cells], np.int32)


def _upload_nt(dn, id, cells):
    n = len(cells)
    arr = {k: np.zeros((n, 3), np.float32) for k in ("t", "t_max", "clearance_constant", "v_p", "k_p")}
    arr["t_max"][...] = 1.0
    arr["clearance_constant"][...] = 0.01
    arr["v_p"][...] = 2.0
    arr["k_p"][...] = 5.0
    flags = np.zeros((n, 3), np.uint32)
    for i, c in enumerate(cells):
        for t, v in c.synaptic_neurotransmitters.items():
            flags[i, int(t)] = 1
            for k in arr:
                if hasattr(v, k):
                    arr[k][i, int(t)] = getattr(v, k)
            if hasattr(v, "decay_constant"):         # ExponentialDecay: same storage as the clearance constant
                arr["clearance_constant"][i, int(t)] = v.decay_constant
    dn.set_attr(id, "neurotransmitters$flags", flags)
    for k, a in arr.items():
        dn.set_attr(id, f"neurotransmitters${k}", a)
    generated = [type(v) for c in cells for v in c.synaptic_neurotransmitters.values() if v.kinetics == NT_CUSTOM]
    for k in generated[0].state_fields if generated else ():            # variables of generated kinetics, per type
        a = np.full((n, 3), generated[0]._defaults[k], np.float32)
        for i, c in enumerate(cells):
            for t, v in c.synaptic_neurotransmitters.items():
                a[i, int(t)] = getattr(v, k)
        dn.set_attr(id, f"neurotransmitters${k}", a)


def _upload_neurons(dn, id, cells):
    if not cells:
        return
    cls = type(cells[0])
    f32 = lambda k: np.array([getattr(c, k) for c in cells], np.float32)
    for k in ("current_voltage", "gap_conductance", "dt", "c_m", "v_th") + tuple(cls.state_fields):
        dn.set_attr(id, cls.abi_names.get(k, k), f32(k))
    dn.set_attr(id, "is_spiking", np.array([c.is_spiking for c in cells], np.uint32))
    dn.set_attr(id, "last_firing_time", _lft(cells))
    for k in getattr(cls, "counter_fields", ()):
        dn.set_attr(id, k, np.array([getattr(c, k) for c in cells], np.uint32))
    if cls.model == HODGKIN_HUXLEY:
        dn.set_attr(id, "was_increasing", np.array([c.was_increasing for c in cells], np.uint32))
    _upload_nt(dn, id, cells)
    n = len(cells)
    flags = np.zeros((n, 3), np.uint32)
    if cls.receptor_set is not None:                 # the generated neuron's own receptor set
        rx = cls.receptor_set
        for name, default in rx.variables:
            if "$" not in name:
                dn.set_attr(id, "receptors$" + name, np.array([getattr(c.receptors, name) for c in cells], np.float32))
        for k, (nt, _, _) in enumerate(rx.types):
            recs = [c.receptors.get(k) for c in cells]
            flags[:, k] = [r is not None for r in recs]
            for name, default in rx.variables:
                if name.startswith(nt + "$"):
                    parts = name.split("$")
                    if len(parts) == 4:             # <Type>$<state>$kinetics$<var>: a field of that state's kinetics value
                        value = lambda r, parts=parts: getattr(getattr(r, parts[1]), parts[3])
                    else:
                        value = lambda r, parts=parts: getattr(r, parts[1])
                    dn.set_attr(id, "receptors$" + name,
                                np.array([default if r is None else value(r) for r in recs], np.float32))
            if getattr(rx, "multi", False):
                continue                            # the states are among the set's variables
            kin = lambda field, d: np.array([d if r is None else getattr(r.r, field, d) for r in recs], np.float32)
            dn.set_attr(id, f"receptors${nt}$r$kinetics$r", kin("r", 0.0))
            dn.set_attr(id, f"receptors${nt}$r$kinetics$alpha", kin("alpha", 1.0))
            dn.set_attr(id, f"receptors${nt}$r$kinetics$beta", kin("beta", 1.0))
        dn.set_attr(id, "receptors$flags", flags)
        return
    for t in IonotropicNeurotransmitterType:
        vals = {k: np.zeros(n, np.float32) for k in ("g", "e", "current", "r", "alpha", "beta", "mg")}
        proto = {0: AMPAReceptor, 1: NMDAReceptor, 2: GABAReceptor}[int(t)]()
        for i, c in enumerate(cells):
            rec = c.receptors.get(t)
            flags[i, int(t)] = rec is not None
            rec = rec or proto
            vals["g"][i], vals["e"][i], vals["current"][i] = rec.g, rec.e, rec.current
            vals["mg"][i] = getattr(rec, "mg", 0.0)
            vals["r"][i] = rec.r.r
            # ExponentialDecayReceptor keeps r_max / decay_constant where Destexhe keeps alpha / beta
            vals["alpha"][i] = getattr(rec.r, "alpha", getattr(rec.r, "r_max", 1.0))
            vals["beta"][i] = getattr(rec.r, "beta", getattr(rec.r, "decay_constant", 1.0))
        p = f"receptors${t.name}"
        dn.set_attr(id, p + "_g", vals["g"])
        dn.set_attr(id, p + "_e", vals["e"])
        dn.set_attr(id, p + "_current", vals["current"])
        dn.set_attr(id, p + "$r$kinetics$r", vals["r"])
        dn.set_attr(id, p + "$r$kinetics$alpha", vals["alpha"])
        dn.set_attr(id, p + "$r$kinetics$beta", vals["beta"])
        if t == IonotropicNeurotransmitterType.NMDA:
            dn.set_attr(id, p + "_mg", vals["mg"])
        generated = [type(c.receptors[t].r) for c in cells if t in c.receptors and c.receptors[t].r.kinetics == RC_CUSTOM]
        for k in generated[0].state_fields if generated else ():        # variables of generated receptor kinetics
            dn.set_attr(id, f"{p}$r$kinetics${k}", np.array(
                [getattr(c.receptors[t].r, k) if t in c.receptors else generated[0]._defaults[k] for c in cells], np.float32))
    dn.set_attr(id, "receptors$flags", flags)


def _download_neurons(dn, id, cells):
    if not cells:
        return
    cls = type(cells[0])
    bools = getattr(getattr(cls, "description", None), "bools", ())       # bool variables of a generated model
    for k in ("current_voltage",) + tuple(cls.state_fields):
        for c, v in zip(cells, dn.get_attr(id, cls.abi_names.get(k, k))):
            setattr(c, k, bool(v) if k in bools else float(v))
    for k in getattr(cls, "counter_fields", ()):
        for c, v in zip(cells, dn.get_attr(id, k, dtype=np.uint32)):
            setattr(c, k, int(v))
    spk = dn.get_attr(id, "is_spiking", dtype=np.uint32)
    lft = dn.get_attr(id, "last_firing_time", dtype=np.int32)
    t = dn.get_attr(id, "neurotransmitters$t", per_type=True)
    for i, c in enumerate(cells):
        c.is_spiking = bool(spk[i])
        c.last_firing_time = None if lft[i] < 0 else int(lft[i])
        for ty, v in c.synaptic_neurotransmitters.items():
            v.t = float(t[i, int(ty)])
    if cls.model == HODGKIN_HUXLEY:
        for c, v in zip(cells, dn.get_attr(id, "was_increasing", dtype=np.uint32)):
            c.was_increasing = bool(v)
    if cls.receptor_set is not None:
        rx = cls.receptor_set
        for name, _ in rx.variables:
            values = dn.get_attr(id, "receptors$" + name)
            for c, v in zip(cells, values):
                value = bool(v) if name in rx.bools else float(v)
                if "$" not in name:
                    setattr(c.receptors, name, value)
                else:
                    parts = name.split("$")
                    rec = c.receptors.get([t[0] for t in rx.types].index(parts[0]))
                    if rec is not None and len(parts) == 4:
                        setattr(getattr(rec, parts[1]), parts[3], value)
                    elif rec is not None:
                        setattr(rec, parts[1], value)
        for k, (nt, _, _) in enumerate([] if getattr(rx, "multi", False) else rx.types):
            for c, v in zip(cells, dn.get_attr(id, f"receptors${nt}$r$kinetics$r")):
                if c.receptors.get(k) is not None:
                    c.receptors[k].r.r = float(v)
        return
    for ty in IonotropicNeurotransmitterType:
        r = dn.get_attr(id, f"receptors${ty.name}$r$kinetics$r")
        cur = dn.get_attr(id, f"receptors${ty.name}_current")
        for i, c in enumerate(cells):
            rec = c.receptors.get(ty)
            if rec is not None:
                rec.r.r, rec.current = float(r[i]), float(cur[i])
        generated = [type(c.receptors[ty].r) for c in cells if ty in c.receptors and c.receptors[ty].r.kinetics == RC_CUSTOM]
        for k in generated[0].state_fields if generated else ():
            for c, v in zip(cells, dn.get_attr(id, f"receptors${ty.name}$r$kinetics${k}")):
                if ty in c.receptors:
                    setattr(c.receptors[ty].r, k, float(v))
    generated = [type(v) for c in cells for v in c.synaptic_neurotransmitters.values() if v.kinetics == NT_CUSTOM]
    for k in generated[0].state_fields if generated else ():
        a = dn.get_attr(id, f"neurotransmitters${k}", per_type=True)
        for i, c in enumerate(cells):
            for ty, v in c.synaptic_neurotransmitters.items():
                setattr(v, k, float(a[i, int(ty)]))


def _upload_cells(dn, id, cells):
    if not cells:
        return
    f32 = lambda k: np.array([getattr(c, k) for c in cells], np.float32)
    for k, a in (("current_voltage", "current_voltage"), ("v_th", "v_th"), ("v_resting", "v_resting"), ("dt", "dt")):
        dn.set_attr(id, a, f32(k))
    refr = [c.neural_refractoriness for c in cells]
    dn.set_attr(id, "neural_refractoriness$k", np.array([c.k if r is None else r.k for c, r in zip(cells, refr)], np.float32))
    dn.set_attr(id, "neural_refractoriness$kind", np.array([0 if r is None else r.kind for r in refr], np.uint32))
    for k in getattr(type(refr[0]), "state_fields", ()) if refr[0] is not None else ():     # generated refractoriness
        dn.set_attr(id, "neural_refractoriness$" + k, np.array([getattr(r, k) for r in refr], np.float32))
    if cells[0].kind in (ST_POISSON, ST_BCM_POISSON):
        dn.set_attr(id, "chance_of_firing", f32("chance_of_firing"))
        dn.set_attr(id, "seed", np.array([c.seed for c in cells], np.uint32))
        if cells[0].kind == ST_BCM_POISSON:
            for k in ("average_activity", "current_activity", "firing_rate_clock", "firing_rate_window"):
                dn.set_attr(id, k, f32(k))
            for k in ("period", "num_spikes"):
                dn.set_attr(id, k, np.array([getattr(c, k) for c in cells], np.uint32))
    elif cells[0].kind == ST_CUSTOM:
        for k in type(cells[0]).state_fields:
            dn.set_attr(id, k, f32(k))
    elif cells[0].kind == ST_PRESET:
        dn.set_attr(id, "internal_clock", f32("internal_clock"))
        dn.set_attr(id, "counter", np.array([c.counter for c in cells], np.uint32))
        ptr = np.concatenate([[0], np.cumsum([len(c.firing_times) for c in cells])]).astype(np.uint32)
        times = np.array([t for c in cells for t in c.firing_times], np.float32)
        dn.set_firing_times(id, ptr, times)
    else:
        dn.set_attr(id, "rate", f32("rate"))
        dn.set_attr(id, "step", f32("step"))
    dn.set_attr(id, "is_spiking", np.array([c.is_spiking for c in cells], np.uint32))
    dn.set_attr(id, "last_firing_time", _lft(cells))
    _upload_nt(dn, id, cells)


def _flat(lattice):
    return [c for row in lattice.cell_grid for c in row]


class LatticeNetworkGPU:
    """LatticeNetworkGPU (backend/src/neuron/gpu_lattices/mod.rs:1517-3212) over one DeviceNetwork, with the method set
    of the reference's Python class (`impl_network_gpu!`, interface_gpu/lixirnet/src/lattices/mod.rs:1450-2117).

    The host container (`self.network`, a LatticeNetwork) holds the network between runs: building methods (connect,
    add_lattice, set_neuron, apply_* ...) edit it and drop the device copy, the next run_lattices uploads again;
    everything that only reads (get_weight, get_neuron, connecting_weights ...) is answered from it -- every
    run_lattices ends with the download of state, weights and histories."""
    network_type = None        # host container class (LatticeNetwork unless a subclass says otherwise)

    def __init__(self, network=None, device=0, graph_history_order=2):
        object.__setattr__(self, "network", network if network is not None else (self.network_type or LatticeNetwork)())
        self._device = device
        self._dn = None
        self._graph_hist = {}
        # when a lattice's weight snapshot is taken: 2 = before the step's weight updates (LatticeNetwork::iterate,
        # neuron/mod.rs:2450-2461), 1 = after them (a lone Lattice, neuron/mod.rs:904-910 -- what LatticeGPU passes)
        self._graph_hist_order = graph_history_order
        if network is not None:
            self._ensure()                      # from_network converts right away, as the reference does

    @classmethod
    def from_network(cls, network, device=0):                   # gpu_lattices/mod.rs:1636-1651
        return cls(copy.deepcopy(network), device=device)

    @classmethod
    def generate_network(cls, lattices=(), spike_train_lattices=(), device=0):   # interface lattices/mod.rs:1465-1500
        g = cls(device=device)
        for l in lattices:
            g.add_lattice(l)
        for l in spike_train_lattices:
            g.add_spike_train_lattice(l)
        return g

    # -- building: edits of the host container invalidate the device copy -----------------------------------------
    _BUILDERS = ("set_dt", "add_lattice", "add_spike_train_lattice", "clear", "connect_internally", "connect",
                 "set_neuron", "set_spike_train", "set_lattice", "set_spike_train_lattice", "set_do_plasticity",
                 "set_plasticity", "apply_lattice", "apply_spike_train_lattice", "apply_lattice_given_position",
                 "apply_spike_train_lattice_given_position")
    _FLAGS = ("electrical_synapse", "chemical_synapse", "parallel", "update_connecting_graph_history")

    def __getattr__(self, name):
        # get_all_ids, get_weight, get_*_connections_*, get_neuron, get_spike_train, get_lattice, connecting_weights,
        # connecting_position_to_index, flags ...: read from the host container; builders drop the device copy first
        if name.startswith("_") or name == "network":
            raise AttributeError(name)
        target = getattr(self.network, name)
        if name in self._BUILDERS:
            def builder(*args, **kw):
                if name in ("add_lattice", "add_spike_train_lattice", "set_lattice", "set_spike_train_lattice"):
                    args = tuple(copy.deepcopy(a) for a in args)          # the container owns its lattices
                self._dirty()
                return target(*args, **kw)
            return builder
        return target

    def __setattr__(self, name, value):
        if name in self._FLAGS:
            return setattr(self.network, name, value)
        object.__setattr__(self, name, value)

    def reset_timing(self, id=None):
        """one lattice (the reference's method) or, without an id, the whole network"""
        ids = self.network.get_all_ids() if id is None else [id]
        for i in ids:
            self.network.reset_timing(i)
        if id is None:
            self.network.internal_clock = 0
            if self._dn is not None:
                self._dn.reset_timing()      # snn_reset_timing: clocks and firing times of the whole handle
        else:
            self._dirty()                    # one lattice only: uploaded again with the next run

    def reset_history(self, id=None):
        for i in (self.network.get_all_ids() if id is None else [id]):
            self.network.reset_history(i)
        if self._dn is not None:
            self._dn.reset_history()

    def set_update_grid_history(self, id, flag):
        self.network.set_update_grid_history(id, flag)

    def set_update_graph_history(self, id, flag):
        self.network.set_update_graph_history(id, flag)

    def _dirty(self):
        if self._dn is not None:
            self._dn.close()
            self._dn = None
            self._graph_hist = {}

    def _ensure(self):
        if self._dn is not None:
            return self._dn
        network = self.network
        neurons = [c for l in network.lattices.values() for c in _flat(l)]
        cells = [c for l in network.spike_train_lattices.values() for c in _flat(l)]
        models = {type(c).model for c in neurons} or {IZHIKEVICH}
        kinds = {c.kind for c in cells} or {ST_NONE}
        if len(models) > 1 or len(kinds) > 1:
            raise TypeError("one neuron model and one spike-train model per network (type parameters in the reference)")
        nt, rc = _kinetics_of(neurons + cells)
        # generated models carry their library; everything generated in one network comes from ONE description
        libs = {getattr(type(c), "lib_path", None) for c in neurons + cells}
        libs |= {getattr(type(v), "lib_path", None) for c in neurons + cells for v in c.synaptic_neurotransmitters.values()}
        libs |= {getattr(type(k), "lib_path", None) for c in neurons for v in c.receptors.values() for k in _receptor_states(v)}
        libs -= {None}
        if len(libs) > 1:
            raise TypeError("the generated models of one network come from one description_builder call")
        self._dn = DeviceNetwork(model=models.pop(), nt_kinetics=nt, receptor_kinetics=rc, spike_train=kinds.pop(),
                                 device=self._device, lib_path=(libs.pop() if libs else None))
        for id, l in network.lattices.items():
            self._dn.add_lattice(id, l.rows, l.cols)
        for id, l in network.spike_train_lattices.items():
            self._dn.add_spike_train_lattice(id, l.rows, l.cols)
        self._dn.finalize()
        self._upload()
        return self._dn

    # InterleavingGraphGPU::convert_to_gpu (graph/mod.rs:644-807)
    def _upload(self):
        dn, net = self._dn, self.network
        for id, l in net.lattices.items():
            _upload_neurons(dn, id, _flat(l))
            p = l.plasticity
            if isinstance(p, BCM):
                dn.set_bcm(id, p.decay, p.average_scalar, p.dt, l.do_plasticity)
            else:
                dn.set_plasticity(id, p.a_plus, p.a_minus, p.tau_plus, p.tau_minus, p.dt, l.do_plasticity)
        for id, l in net.spike_train_lattices.items():
            _upload_cells(dn, id, _flat(l))
        nn, nt = dn.n_neurons, dn.n_tot
        if nn == 0 or nt == 0:
            return
        w = np.zeros((nt, nn), np.float32)
        c = np.zeros((nt, nn), np.uint32)
        for id, l in net.lattices.items():
            first, count = dn.lattice_range(id)
            w[first:first + count, first:first + count] = l.weights
            c[first:first + count, first:first + count] = l.connections
        for (pre, post), weight in net.connecting.items():
            i, j = self._global(pre), self._global(post)
            w[i, j], c[i, j] = weight, 1
        dn.set_graph_rows(0, w, c)
        for id, l in net.lattices.items():
            if isinstance(l, RewardModulatedLattice):
                m = l.reward_modulator
                dn.set_reward_modulator(id, m.dopamine, m.tau_d, m.tau_c, m.a_plus, m.a_minus, m.tau_plus, m.tau_minus,
                                        m.dt, l.do_modulation)
                first, count = dn.lattice_range(id)
                t = np.zeros((count, nn), np.float32)
                t[:, first:first + count] = l.traces
                dn.set_trace_rows(first, t)
        # internal_clock: lattice_network.internal_clock (gpu_lattices/mod.rs:1630): a network that has already run goes on at its
        # clock (last_firing_time values are absolute step numbers); every spike-train lattice keeps its own
        dn.set_clock(int(getattr(net, "internal_clock", 0) or 0))
        for id, l in net.spike_train_lattices.items():
            dn.set_spike_train_clock(id, int(getattr(l, "internal_clock", 0) or 0))

    def _global(self, gp):
        first, _ = self._dn.lattice_range(gp.id)
        l = self.network.lattices.get(gp.id) or self.network.spike_train_lattices[gp.id]
        return first + gp.pos[0] * l.cols + gp.pos[1]

    def _history_flags(self):
        dn, net = self._dn, self.network
        hist = any(l.update_grid_history for l in list(net.lattices.values()) + list(net.spike_train_lattices.values()))
        dn.set_history(voltage=hist, spikes=False)
        for id, l in net.lattices.items():                      # update_graph_history, neuron/mod.rs:572
            if getattr(l, "update_graph_history", False) != self._graph_hist.get(id, False):
                dn.set_graph_history(id, self._graph_hist_order if l.update_graph_history else 0)
                self._graph_hist[id] = l.update_graph_history

    def graph_history(self, id):
        """[steps][n][n] internal weights of lattice `id` after every step (AdjacencyMatrix::history)."""
        return self._ensure().graph_history(id)

    def run_lattices(self, iterations):                         # gpu_lattices/mod.rs:3183-3212
        dn, net = self._ensure(), self.network
        dn.set_synapses(net.electrical_synapse, net.chemical_synapse)
        self._history_flags()
        dn.run(iterations)
        self._download()

    def run_lattices_with_reward(self, reward, download=True):
        """One step preceded by RewardModulator::update(reward) on every reward-modulated lattice
        (RewardModulatedLatticeNetwork::run_lattices_with_reward, neuron/mod.rs:5385-5408).  `download=False` leaves
        the results on the device until the next downloading call (an agent loop steps thousands of times)."""
        dn, net = self._ensure(), self.network
        dn.set_synapses(net.electrical_synapse, net.chemical_synapse)
        self._history_flags()
        dn.run_with_reward(float(reward))
        if download:
            self._download()

    def _download(self):
        dn, net = self._dn, self.network
        net.internal_clock = dn.clock
        nn, nt = dn.n_neurons, dn.n_tot
        w = c = None
        if nn and nt:
            w, c = dn.get_graph_rows(0, nt)
        for id, l in net.lattices.items():
            _download_neurons(dn, id, _flat(l))
            l.internal_clock = net.internal_clock
            if w is not None:
                first, count = dn.lattice_range(id)
                l.weights = w[first:first + count, first:first + count].copy()
                l.connections = c[first:first + count, first:first + count].copy()
                if isinstance(l, RewardModulatedLattice):
                    l.traces = dn.get_trace_rows(first, count)[:, first:first + count].copy()
                    l.reward_modulator.dopamine = float(dn.dopamine(id))
        for id, l in net.spike_train_lattices.items():
            l.internal_clock = dn.spike_train_clock(id)             # SpikeTrainLattice::internal_clock, neuron/mod.rs:1318
            cells = _flat(l)
            if cells:
                for cell, v, s, t in zip(cells, dn.get_attr(id, "current_voltage"),
                                         dn.get_attr(id, "is_spiking", dtype=np.uint32),
                                         dn.get_attr(id, "last_firing_time", dtype=np.int32)):
                    cell.current_voltage, cell.is_spiking = float(v), bool(s)
                    cell.last_firing_time = None if t < 0 else int(t)
                if cells[0].kind == ST_CUSTOM:
                    bools = type(cells[0]).description.bools
                    for k in type(cells[0]).state_fields:
                        for cell, v in zip(cells, dn.get_attr(id, k)):
                            setattr(cell, k, bool(v) if k in bools else float(v))
                if cells[0].kind == ST_PRESET:
                    for cell, clk, cnt in zip(cells, dn.get_attr(id, "internal_clock"),
                                              dn.get_attr(id, "counter", dtype=np.uint32)):
                        cell.internal_clock, cell.counter = float(clk), int(cnt)
        if w is not None:
            for key in net.connecting:
                net.connecting[key] = float(w[self._global(key[0]), self._global(key[1])])
        # the histories the reference keeps in its lattices (GridVoltageHistory; AdjacencyMatrix::history)
        for id, l in list(net.lattices.items()) + list(net.spike_train_lattices.items()):
            if l.update_grid_history and l.rows * l.cols:
                l.history = dn.voltage_history(id).reshape(-1, l.rows, l.cols)
        for id, l in net.lattices.items():
            if getattr(l, "update_graph_history", False) and l.rows * l.cols:
                l.weights_history = dn.graph_history(id)

    def history(self, id):
        """[steps][rows][cols] voltages of lattice `id` (GridVoltageHistory, gpu_lattices/mod.rs:189-280)."""
        l = self.network.lattices.get(id) or self.network.spike_train_lattices[id]
        if self._dn is None:
            return np.zeros((0, l.rows, l.cols), np.float32)
        return self._dn.voltage_history(id).reshape(-1, l.rows, l.cols)

    # Reduced histories kept on the device (the CPU lattices' other LatticeHistory types, neuron/mod.rs:233-360;
    # the reference's GPU lattices only carry GridVoltageHistory).
    def set_reduced_history(self, average_voltage=False, eeg=False, spike_counts=False,
                            reference_voltage=0.007, distance=0.8, conductivity=251.0):
        self._ensure().set_reduced_history(average_voltage, eeg, spike_counts, reference_voltage, distance, conductivity)

    def average_voltage_history(self, id):
        """one value per step (AverageVoltageHistory, neuron/mod.rs:305-322)"""
        return self._dn.average_voltage_history(id)

    def eeg_history(self, id):
        """one value per step (EEGHistory, neuron/mod.rs:233-284)"""
        return self._dn.eeg_history(id)

    def spike_counts(self, id):
        """[rows][cols] spike totals since the last reset (SpikeHistory::aggregate, neuron/mod.rs:331-360)"""
        l = self.network.lattices[id]
        return self._dn.spike_counts(id).reshape(l.rows, l.cols)

    def close(self):
        self._dirty()


class LatticeGPU:
    """LatticeGPU (backend/src/neuron/gpu_lattices/mod.rs:327-1118): a network of one lattice."""
    lattice_type = Lattice

    def __init__(self, id=0, device=0):
        self._lattice = self.lattice_type(id)
        self._device = device
        self._net = None

    @classmethod
    def from_lattice(cls, lattice, device=0):                   # gpu_lattices/mod.rs:496-511
        g = cls(lattice.id, device=device)
        g._lattice = copy.deepcopy(lattice)
        return g

    def _ensure(self):
        if self._net is None:
            host = LatticeNetwork.generate_network([self._lattice])
            self._net = LatticeNetworkGPU(host, device=self._device, graph_history_order=1)
        return self._net

    def _dirty(self):
        if self._net is not None:
            self._net.close()
            self._net = None

    # building methods forward to the host lattice (and invalidate the device copy)
    def populate(self, neuron, num_rows, num_cols):
        self._dirty()
        self._lattice.populate(neuron, num_rows, num_cols)

    def connect(self, connection_conditional, weight_logic=None):
        self._dirty()
        self._lattice.connect(connection_conditional, weight_logic)

    def apply(self, function):
        self._dirty()
        self._lattice.apply(function)

    def apply_given_position(self, function):
        self._dirty()
        self._lattice.apply_given_position(function)

    def set_neuron(self, row, col, neuron):
        self._dirty()
        self._lattice.set_neuron(row, col, neuron)

    def set_dt(self, dt):
        self._dirty()
        self._lattice.set_dt(dt)

    def __getattr__(self, name):        # get_neuron, get_weight, get_*_connections, position_to_index, flags ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._lattice, name)

    def __setattr__(self, name, value):
        if name.startswith("_") or name in ("lattice_type",):
            return object.__setattr__(self, name, value)
        if name in ("update_grid_history", "update_graph_history", "electrical_synapse", "chemical_synapse",
                    "do_plasticity", "plasticity", "id"):
            if name in ("do_plasticity", "plasticity", "id"):
                self._dirty()
            return setattr(self._lattice, name, value)
        object.__setattr__(self, name, value)

    def run_lattice(self, iterations):                          # gpu_lattices/mod.rs:1081-1100
        net = self._ensure()
        net.network.electrical_synapse = self._lattice.electrical_synapse
        net.network.chemical_synapse = self._lattice.chemical_synapse
        net.run_lattices(iterations)

    @property
    def history(self):
        if self._net is None:
            return np.zeros((0, self._lattice.rows, self._lattice.cols), np.float32)
        return self._net.history(self._lattice.id)

    @property
    def weights(self):
        return self._lattice.weights

    @property
    def graph_history(self):
        n = self._lattice.rows * self._lattice.cols
        if self._net is None:
            return np.zeros((0, n, n), np.float32)
        return self._net.graph_history(self._lattice.id)

    def reset_history(self):
        if self._net is not None:
            self._net.reset_history()

    def reset_timing(self):
        if self._net is not None:
            self._net.reset_timing()
        else:
            self._lattice.reset_timing()

    def close(self):
        self._dirty()


class RewardModulatedLatticeGPU(LatticeGPU):
    """A RewardModulatedLattice stepped on the device (the reference has no GPU form of it).  Agent interface:
    update_and_apply_reward / update (neuron/mod.rs:3402-3415)."""
    lattice_type = RewardModulatedLattice

    def __setattr__(self, name, value):
        if name in ("do_modulation", "reward_modulator"):
            self._dirty()
            return setattr(self._lattice, name, value)
        super().__setattr__(name, value)

    def run_lattice_with_reward(self, reward, download=True):       # neuron/mod.rs:3250-3257
        net = self._ensure()
        net.network.electrical_synapse = self._lattice.electrical_synapse
        net.network.chemical_synapse = self._lattice.chemical_synapse
        net.run_lattices_with_reward(reward, download=download)

    def update_and_apply_reward(self, reward):
        self.run_lattice_with_reward(reward, download=False)

    def update(self):
        net = self._ensure()
        net._dn.set_synapses(self._lattice.electrical_synapse, self._lattice.chemical_synapse)
        net._dn.run(1)

    def sync(self):
        """Bring weights, traces, dopamine and neuron state back after a series of non-downloading steps."""
        if self._net is not None:
            self._net._download()

    @property
    def traces(self):
        return self._lattice.traces


def _named(base, name, **attrs):
    return type(name, (base,), attrs)


# class names as in Lixirnet (interface_gpu/lixirnet/src/lib.rs:463-482), one family per backend model
IzhikevichNeuronLattice = _named(Lattice, "IzhikevichNeuronLattice", neuron_type=IzhikevichNeuron)
LeakyIntegrateAndFireNeuronLattice = _named(Lattice, "LeakyIntegrateAndFireNeuronLattice",
                                            neuron_type=LeakyIntegrateAndFireNeuron)
HodgkinHuxleyNeuronLattice = _named(Lattice, "HodgkinHuxleyNeuronLattice", neuron_type=HodgkinHuxleyNeuron)
QuadraticIntegrateAndFireNeuronLattice = _named(Lattice, "QuadraticIntegrateAndFireNeuronLattice",
                                                neuron_type=QuadraticIntegrateAndFireNeuron)
SimpleLeakyIntegrateAndFireLattice = _named(Lattice, "SimpleLeakyIntegrateAndFireLattice",
                                            neuron_type=SimpleLeakyIntegrateAndFire)
QuadraticIntegrateAndFireNeuronLatticeGPU = _named(LatticeGPU, "QuadraticIntegrateAndFireNeuronLatticeGPU",
                                                   lattice_type=QuadraticIntegrateAndFireNeuronLattice)
SimpleLeakyIntegrateAndFireLatticeGPU = _named(LatticeGPU, "SimpleLeakyIntegrateAndFireLatticeGPU",
                                               lattice_type=SimpleLeakyIntegrateAndFireLattice)
AdaptiveLeakyIntegrateAndFireNeuronLattice = _named(Lattice, "AdaptiveLeakyIntegrateAndFireNeuronLattice",
                                                    neuron_type=AdaptiveLeakyIntegrateAndFireNeuron)
AdaptiveExpLeakyIntegrateAndFireNeuronLattice = _named(Lattice, "AdaptiveExpLeakyIntegrateAndFireNeuronLattice",
                                                       neuron_type=AdaptiveExpLeakyIntegrateAndFireNeuron)
LeakyIzhikevichNeuronLattice = _named(Lattice, "LeakyIzhikevichNeuronLattice", neuron_type=LeakyIzhikevichNeuron)
AdaptiveLeakyIntegrateAndFireNeuronLatticeGPU = _named(LatticeGPU, "AdaptiveLeakyIntegrateAndFireNeuronLatticeGPU",
                                                       lattice_type=AdaptiveLeakyIntegrateAndFireNeuronLattice)
AdaptiveExpLeakyIntegrateAndFireNeuronLatticeGPU = _named(LatticeGPU, "AdaptiveExpLeakyIntegrateAndFireNeuronLatticeGPU",
                                                          lattice_type=AdaptiveExpLeakyIntegrateAndFireNeuronLattice)
LeakyIzhikevichNeuronLatticeGPU = _named(LatticeGPU, "LeakyIzhikevichNeuronLatticeGPU",
                                         lattice_type=LeakyIzhikevichNeuronLattice)
IzhikevichNeuronLatticeGPU = _named(LatticeGPU, "IzhikevichNeuronLatticeGPU", lattice_type=IzhikevichNeuronLattice)
LeakyIntegrateAndFireNeuronLatticeGPU = _named(LatticeGPU, "LeakyIntegrateAndFireNeuronLatticeGPU",
                                               lattice_type=LeakyIntegrateAndFireNeuronLattice)
HodgkinHuxleyNeuronLatticeGPU = _named(LatticeGPU, "HodgkinHuxleyNeuronLatticeGPU",
                                       lattice_type=HodgkinHuxleyNeuronLattice)
RateSpikeTrainLattice = _named(SpikeTrainLattice, "RateSpikeTrainLattice", spike_train_type=RateSpikeTrain)
PoissonNeuronLattice = _named(SpikeTrainLattice, "PoissonNeuronLattice", spike_train_type=PoissonNeuron)
PresetSpikeTrainLattice = _named(SpikeTrainLattice, "PresetSpikeTrainLattice", spike_train_type=PresetSpikeTrain)
BCMPoissonNeuronLattice = _named(SpikeTrainLattice, "BCMPoissonNeuronLattice", spike_train_type=BCMPoissonNeuron)
BCMIzhikevichNeuronLattice = _named(Lattice, "BCMIzhikevichNeuronLattice", neuron_type=BCMIzhikevichNeuron)
BCMIzhikevichNeuronLatticeGPU = _named(LatticeGPU, "BCMIzhikevichNeuronLatticeGPU", lattice_type=BCMIzhikevichNeuronLattice)
IzhikevichNeuronNetwork = _named(LatticeNetwork, "IzhikevichNeuronNetwork")
IzhikevichNeuronNetworkGPU = _named(LatticeNetworkGPU, "IzhikevichNeuronNetworkGPU")

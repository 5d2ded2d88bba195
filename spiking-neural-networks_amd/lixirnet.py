"""`import lixirnet` of the reference's GPU Python interface, over the HIP stepper.

The reference's module (interface_gpu/lixirnet/src/lib.rs) hands the description at lib.rs:22-79 to `neuron_builder!`
and registers seventeen classes (lib.rs:463-482).  This module builds the SAME description with this package's
generator (modelgen -> one compiled library, cached under csrc/generated) and publishes the same seventeen names with
the same constructors, attributes and methods:

    IzhikevichNeuron, BoundedNeurotransmitterKinetics, BoundedReceptorKinetics, DopaGluGABANeurotransmitterType,
    GlutamateReceptor, GABAReceptor, DopamineReceptor, DopaGluGABA, STDP, IzhikevichNeuronLattice,
    IzhikevichNeuronLatticeGPU, DeltaDiracRefractoriness, RateSpikeTrain, RateSpikeTrainLattice, GraphPosition,
    IzhikevichNeuronNetwork, IzhikevichNeuronNetworkGPU

so that the procedure of the reference's own Python tests (interface_gpu/lixirnet/tests/lattices.py, networks.py) runs
against it unchanged apart from the import (`from snn_amd import lixirnet as ln`).  The `*Lattice` / `*Network`
classes only build; stepping lives on the `*GPU` classes (there is no CPU stepper in this package).

The library is compiled the first time one of the names is touched (hipcc, about a minute, once per checkout).
"""
import ctypes as _C

import numpy as _np

from . import lattice as _l
from .examples_dsl import LIXIRNET as DESCRIPTION

_NAMES = ("IzhikevichNeuron", "BoundedNeurotransmitterKinetics", "BoundedReceptorKinetics",
          "DopaGluGABANeurotransmitterType", "GlutamateReceptor", "GABAReceptor", "DopamineReceptor", "DopaGluGABA",
          "STDP", "IzhikevichNeuronLattice", "IzhikevichNeuronLatticeGPU", "DeltaDiracRefractoriness", "RateSpikeTrain",
          "RateSpikeTrainLattice", "GraphPosition", "IzhikevichNeuronNetwork", "IzhikevichNeuronNetworkGPU")
__all__ = list(_NAMES)
_built = None


class DeltaDiracRefractoriness(_l.DeltaDiracRefractoriness):
    """DeltaDiracRefractoriness(k) (lib.rs:213-241)"""

    def __init__(self, k=10000.0):
        super().__init__(k=float(k))

    def get_effect(self, timestep, last_firing_time, v_max, v_resting, dt):
        """a * exp((-1 / (k / dt)) * time_difference^2) + v_resting (backend spike_train/mod.rs:79-88), float32 with
        the host libm's expf"""
        f = _np.float32
        libm = _C.CDLL("libm.so.6")
        libm.expf.restype, libm.expf.argtypes = _C.c_float, [_C.c_float]
        a = f(v_max) - f(v_resting)
        td = f(int(timestep) - int(last_firing_time))
        x = (f(-1.0) / (f(self.k) / f(dt))) * (td * td)
        return float(a * f(libm.expf(float(x))) + f(v_resting))


def _build():
    global _built
    if _built is not None:
        return _built
    g = _l.description_builder(DESCRIPTION)
    nt_type = g.NeurotransmitterType

    def typed(d):
        out = {}
        for k, v in d.items():
            if not isinstance(v, g.Neurotransmitter):
                raise TypeError("Incorrect neurotransmitter kinetics type")
            try:
                out[nt_type(int(k))] = v
            except ValueError:
                raise TypeError("Incorrect neurotransmitter type") from None
        return out

    class IzhikevichNeuron(g.Neuron):
        __doc__ = g.Neuron.__doc__

        def set_synaptic_neurotransmitters(self, d):
            self.synaptic_neurotransmitters = typed(d)

        def get_synaptic_neurotransmitters(self):
            return dict(self.synaptic_neurotransmitters)

        def get_receptors(self):
            return self.receptors

    IzhikevichNeuron.__name__ = IzhikevichNeuron.__qualname__ = "IzhikevichNeuron"

    class RateSpikeTrain(_l.RateSpikeTrain):
        """RateSpikeTrain<DopaGluGABANeurotransmitterType, BoundedNeurotransmitterKinetics, DeltaDiracRefractoriness>
        (lib.rs:243-378)"""

        def set_synaptic_neurotransmitters(self, d):
            self.synaptic_neurotransmitters = typed(d)

        def get_synaptic_neurotransmitters(self):
            return dict(self.synaptic_neurotransmitters)

        def get_refractoriness(self):
            r = self.neural_refractoriness
            return DeltaDiracRefractoriness(self.k if r is None else r.k)

        def set_refractoriness(self, refractoriness):
            self.neural_refractoriness = refractoriness
            self.k = refractoriness.k

        def iterate(self):
            """one host-side RateSpikeTrain::iterate (backend spike_train/mod.rs:1016-1031) without the transmitter
            update (the kinetics are generated device code); lattices are stepped on the GPU"""
            f = _np.float32
            step = f(self.step) + f(self.dt)
            fire = f(self.rate) != 0 and step >= f(self.rate)
            self.step = 0.0 if fire else float(step)
            self.current_voltage = self.v_th if fire else self.v_resting
            self.is_spiking = bool(fire)
            return bool(fire)

    lattice = type("IzhikevichNeuronLattice", (_l.Lattice,), dict(neuron_type=IzhikevichNeuron))
    st_lattice = type("RateSpikeTrainLattice", (_l.SpikeTrainLattice,), dict(spike_train_type=RateSpikeTrain))
    network = type("IzhikevichNeuronNetwork", (_l.LatticeNetwork,), {})
    _built = dict(
        IzhikevichNeuron=IzhikevichNeuron, BoundedNeurotransmitterKinetics=g.Neurotransmitter,
        BoundedReceptorKinetics=g.ReceptorKinetics, DopaGluGABANeurotransmitterType=nt_type,
        GlutamateReceptor=g.receptor_types["Glutamate"], GABAReceptor=g.receptor_types["GABA"],
        DopamineReceptor=g.receptor_types["Dopamine"], DopaGluGABA=g.Receptors, STDP=_l.STDP,
        IzhikevichNeuronLattice=lattice,
        IzhikevichNeuronLatticeGPU=type("IzhikevichNeuronLatticeGPU", (_l.LatticeGPU,), dict(lattice_type=lattice)),
        DeltaDiracRefractoriness=DeltaDiracRefractoriness, RateSpikeTrain=RateSpikeTrain,
        RateSpikeTrainLattice=st_lattice, GraphPosition=_l.GraphPosition, IzhikevichNeuronNetwork=network,
        IzhikevichNeuronNetworkGPU=type("IzhikevichNeuronNetworkGPU", (_l.LatticeNetworkGPU,), dict(network_type=network)),
        library=g.library, description=g.description)
    IzhikevichNeuron.receptors_type = g.Receptors
    return _built


def __getattr__(name):                   # PEP 562: the classes appear (and the library is compiled) on first use
    if name in _NAMES or name in ("library", "description"):
        return _build()[name]
    raise AttributeError(f"module 'lixirnet' has no attribute {name!r}")


def __dir__():
    return sorted(list(globals()) + list(_NAMES))

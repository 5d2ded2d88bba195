"""MI355X-native spiking-lattice time-stepper (host side, Python).

One hot path of NikhilMukraj/spiking-neural-networks -- `run_lattice` / `run_lattices`
(backend/src/neuron/mod.rs:1199-1220, 2654-2675) -- as hand-written HIP kernels for gfx950
behind the C ABI of include/snn_amd.h.  The directory name carries a hyphen, so import it
through the root-level shim: `import snn_amd`.
"""
from . import _lib, lattice, parallel, synthetic
from ._lib import SnnError, SnnLibraryError, build
from .network import (DeviceNetwork, HODGKIN_HUXLEY, IZHIKEVICH, LIF, QUADRATIC_INTEGRATE_AND_FIRE, SIMPLE_LIF,
                      ADAPTIVE_LIF, ADAPTIVE_EXP_LIF, LEAKY_IZHIKEVICH, NT_APPROXIMATE, NT_DESTEXHE,
                      NUM_NT_TYPES, RC_APPROXIMATE, RC_DESTEXHE, ST_NONE, ST_POISSON, ST_RATE, probe_bandwidth, probe_math, probe_math_bits,
                      NT_DISCRETE_SPIKE, NT_EXPONENTIAL_DECAY, RC_EXPONENTIAL_DECAY, ST_PRESET, BCM_IZHIKEVICH, ST_BCM_POISSON, CUSTOM, ST_CUSTOM,
                      REFRACTORINESS_CUSTOM, NT_CUSTOM, RC_CUSTOM)

from . import modelgen  # noqa: F401
from .lattice import *  # noqa: F401,F403  (Lixirnet-style names)

__all__ = ["DeviceNetwork", "SnnError", "SnnLibraryError", "build", "probe_math", "probe_math_bits", "probe_bandwidth",
           "IZHIKEVICH", "LIF", "HODGKIN_HUXLEY", "QUADRATIC_INTEGRATE_AND_FIRE", "SIMPLE_LIF", "ADAPTIVE_LIF", "ADAPTIVE_EXP_LIF", "LEAKY_IZHIKEVICH", "NT_APPROXIMATE", "NT_DESTEXHE",
           "RC_APPROXIMATE", "RC_DESTEXHE", "ST_NONE", "ST_POISSON", "ST_RATE", "NUM_NT_TYPES",
           "NT_DISCRETE_SPIKE", "NT_EXPONENTIAL_DECAY", "RC_EXPONENTIAL_DECAY", "ST_PRESET", "BCM_IZHIKEVICH", "ST_BCM_POISSON", "CUSTOM", "ST_CUSTOM",
           "REFRACTORINESS_CUSTOM", "NT_CUSTOM", "RC_CUSTOM", "modelgen"]

"""One small lattice on the per-step path, for a kernel trace:  rocprofv3 --kernel-trace --stats -- python3 profiles/trace_small_step.py
<side> <chemical 0|1> <plastic 0|1> [steps]  (the one-launch run is switched off: what a network outside its reach pays per step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import snn_amd

side, chemical, plastic = int(sys.argv[1]), sys.argv[2] == "1", sys.argv[3] == "1"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
dn.add_lattice(0, side, side)
dn.finalize()
n = side * side
rng = np.random.default_rng(side)
dn.set_attr(0, "current_voltage", rng.uniform(-70.0, 29.9, n).astype(np.float32))
dn.set_attr(0, "gap_conductance", rng.uniform(0.2, 1.0, n).astype(np.float32))
if chemical:
    flags = np.zeros((n, 3), np.uint32)
    flags[:, 0] = 1
    dn.set_attr(0, "neurotransmitters$flags", flags)
    dn.set_attr(0, "receptors$flags", flags)
dn.fill_graph_synthetic(7, 0.5, 1.5)
dn.set_synapses(True, chemical)
if plastic:
    dn.set_plasticity(0)
dn.set_option("persistent_run", 0)
dn.run(steps)
dn.synchronize()
print({k: dn.stat(k) for k in ("steps_dense_one_launch", "steps_two_kernel", "shadow_refreshes")})
dn.close()

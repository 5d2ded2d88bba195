import sys
import numpy as np
sys.path.insert(0, ".")
import snn_amd
from snn_amd import synthetic
side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = side * side
dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, spike_train=snn_amd.ST_POISSON)
dn.add_lattice(0, side, side)
dn.add_spike_train_lattice(1, side, side)
dn.finalize()
dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
dn.set_attr(1, "chance_of_firing", np.full(n, float(sys.argv[2]) if len(sys.argv) > 2 else 0.01, np.float32))
dn.set_attr(1, "seed", np.arange(1, n + 1, dtype=np.uint32))
dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
dn.run(200)
dn.run(1000)
dn.close()

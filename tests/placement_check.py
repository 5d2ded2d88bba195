"""Manual tool: bandwidth of the 256x256 input pass over several handles / processes (placement selection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, snn_amd
from snn_amd import synthetic
n = 65536
out = []
for i in range(4):
    dn = snn_amd.DeviceNetwork(); dn.add_lattice(0, 256, 256); dn.finalize()
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32)); dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65, 30))
    dn.fill_graph_synthetic(2, 0.5, 1.5); dn.run(10); dn.profile_enable(True); dn.profile_reset(); dn.run(30)
    l, ms = dn.profile_read(); out.append(round(dn.input_kernel_bytes() / (ms / l * 1e-3) / 1e9)); dn.close()
print("GB/s per handle:", out, flush=True)

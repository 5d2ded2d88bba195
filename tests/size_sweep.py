"""Step time across lattice sizes (manual tool, not a test): python tests/size_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snn_amd
from snn_amd import synthetic

for side in (8, 16, 32, 48, 64, 96, 128, 192, 256):
    n = side * side
    dn = snn_amd.DeviceNetwork()
    dn.add_lattice(0, side, side)
    dn.finalize()
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
    dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65, 30))
    dn.fill_graph_synthetic(2, 0.5, 1.5)
    steps = 2000 if n <= 4096 else (500 if n <= 16384 else 100)
    dn.run(50)
    t0 = time.perf_counter()
    dn.run(steps)
    dt = time.perf_counter() - t0
    print(f"{side}x{side}: {dt / steps * 1e6:9.1f} us/step  {n * steps / dt / 1e6:8.1f} M neuron-steps/s  "
          f"W={4 * n * n / 1e6:9.1f} MB  eff {4 * n * n / (dt / steps) / 1e9:7.0f} GB/s", flush=True)
    dn.close()

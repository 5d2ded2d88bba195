"""Per-step time of small electrical-only networks WITH Poisson cells (one per neuron), one-launch run on and off."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import snn_amd
from snn_amd import synthetic

for side in (8, 16, 22, 32, 45):
    for persistent in (1, 0):
        n = side * side
        dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, spike_train=snn_amd.ST_POISSON)
        dn.add_lattice(0, side, side)
        dn.add_spike_train_lattice(1, side, side)
        dn.finalize()
        dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
        dn.set_attr(1, "chance_of_firing", np.full(n, 0.01, np.float32))
        dn.set_attr(1, "seed", np.arange(1, n + 1, dtype=np.uint32))
        dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
        dn.set_option("persistent_run", persistent)
        dn.run(200)
        dn.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dn.run(1000)
        dn.synchronize()
        print(f"{side}x{side} neurons + {side}x{side} Poisson cells, persistent_run={persistent}: "
              f"{(time.perf_counter() - t0) / 5000 * 1e6:.2f} us/step (launches {dn.stat('persistent_run_launches')})", flush=True)
        dn.close()

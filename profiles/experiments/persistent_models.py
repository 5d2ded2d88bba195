"""Per-step time of 32x32 electrical-only lattices of the built-in models with the one-launch run on and off."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import snn_amd
from snn_amd import synthetic

MODELS = [("izhikevich", snn_amd.IZHIKEVICH, (-65.0, 30.0)), ("lif", snn_amd.LIF, (-80.0, -50.0)),
          ("hodgkin_huxley", snn_amd.HODGKIN_HUXLEY, (-75.0, -40.0)), ("qif", snn_amd.QUADRATIC_INTEGRATE_AND_FIRE, (-75.0, -56.0)),
          ("simple_lif", snn_amd.SIMPLE_LIF, (-75.0, -56.0))]
for name, model, (lo, hi) in MODELS:
    for persistent in (1, 0):
        dn = snn_amd.DeviceNetwork(model=model)
        dn.add_lattice(0, 32, 32)
        dn.finalize()
        n = 1024
        dn.set_attr(0, "gap_conductance", np.full(n, 3.0, np.float32))
        dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, lo, hi))
        dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
        dn.set_option("persistent_run", persistent)
        dn.run(200)
        dn.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dn.run(1000)
        dn.synchronize()
        print(f"{name:16s} persistent_run={persistent}: {(time.perf_counter() - t0) / 5000 * 1e6:.2f} us/step  "
              f"(launches {dn.stat('persistent_run_launches')})", flush=True)
        dn.close()

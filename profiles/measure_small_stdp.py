"""Small plastic lattices: the one-launch run with the STDP updates inside it against one launch per step (+ the scatter
kernels): us per step, `python3 profiles/measure_small_stdp.py [steps]`.  One JSON line per lattice and form."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import snn_amd


def build(side, persistent):
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
    dn.add_lattice(0, side, side)
    dn.finalize()
    n = side * side
    rng = np.random.default_rng(side)
    dn.set_attr(0, "current_voltage", rng.uniform(-70.0, 29.9, n).astype(np.float32))
    dn.set_attr(0, "gap_conductance", rng.uniform(0.2, 1.0, n).astype(np.float32))
    dn.fill_graph_synthetic(7, 0.5, 1.5)
    dn.set_synapses(True, False)
    dn.set_plasticity(0)
    dn.set_option("persistent_run", int(persistent))
    return dn


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    for side in (8, 16, 24, 32):
        for persistent in (True, False):
            dn = build(side, persistent)
            dn.run(200)
            dn.synchronize()
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                dn.run(steps)
                dn.synchronize()
                runs.append((time.perf_counter() - t0) / steps * 1e6)
            print(json.dumps({"lattice": f"{side}x{side}", "rule": "STDP", "one_launch_run": persistent, "us_per_step": float(np.median(runs)),
                              "us_per_step_runs": runs, "steps": steps, "stdp_steps_in_run": dn.stat("persistent_run_stdp_steps"),
                              }), flush=True)
            dn.close()

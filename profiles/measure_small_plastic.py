"""Small PLASTIC lattices outside the reach of the one-launch run (chemical synapses, spike-train cells, or the run switched off):
one launch per step, with the STDP of a step in one launch (k_stdp_small, option "stdp_small" 1) against four (counter fill,
k_spike_compact, k_stdp_columns, k_stdp_rows): us per step, `python3 profiles/measure_small_plastic.py [steps]`.  One JSON line
per lattice and form."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import snn_amd


def build(side, chemical, cells, small):
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, spike_train=snn_amd.ST_RATE if cells else snn_amd.ST_NONE)
    dn.add_lattice(0, side, side)
    if cells:
        dn.add_spike_train_lattice(5, 4, 4)
    dn.finalize()
    n = side * side
    rng = np.random.default_rng(side)
    dn.set_attr(0, "current_voltage", rng.uniform(-70.0, 29.9, n).astype(np.float32))
    dn.set_attr(0, "gap_conductance", rng.uniform(0.2, 1.0, n).astype(np.float32))
    if chemical:
        flags = np.zeros((n, 3), np.uint32)
        flags[:, 0] = 1                                   # AMPA released and received by every neuron
        dn.set_attr(0, "neurotransmitters$flags", flags)
        dn.set_attr(0, "receptors$flags", flags)
    if cells:
        dn.set_attr(5, "rate", np.full(16, 3.0, np.float32))
    dn.fill_graph_synthetic(7, 0.5, 1.5)
    dn.set_synapses(True, chemical)
    dn.set_plasticity(0)
    dn.set_option("persistent_run", 0)
    dn.set_option("stdp_small", int(small))
    return dn


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    for side in (8, 16, 24, 31):
        for chemical, cells in ((False, False), (True, False), (True, True)):
            for small in (True, False):
                dn = build(side, chemical, cells, small)
                dn.run(200)
                dn.synchronize()
                runs = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    dn.run(steps)
                    dn.synchronize()
                    runs.append((time.perf_counter() - t0) / steps * 1e6)
                print(json.dumps({"lattice": f"{side}x{side}", "synapses": "el + AMPA" if chemical else "el", "rate_cells": 16 if cells else 0,
                                  "rule": "STDP", "stdp_in_one_launch": small, "us_per_step": float(np.median(runs)), "us_per_step_runs": runs,
                                  "steps": steps, "one_launch_run_steps": dn.stat("persistent_run_steps")}), flush=True)
                dn.close()

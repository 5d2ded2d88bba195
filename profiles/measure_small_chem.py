#!/usr/bin/env python3
"""us per step of SMALL dense lattices with chemical synapses (the sizes the reference's own Python tests run,
interface_gpu/lixirnet/tests/networks.py:124-160): Izhikevich, gap junctions + AMPA (every neuron releases and receives it,
Approximate kinetics), all-to-all; the one-launch run (k_run_resident<..., CHEM>) against one launch per step
(option "persistent_chem" 0 -> k_step_resident).  One JSON line per case.

    python3 profiles/measure_small_chem.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import snn_amd              # noqa: E402
from snn_amd import synthetic   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
CASES = [(int(c.split(":")[0]), int(c.split(":")[1]), c) for c in os.environ["SNN_CASES"].split(",")] if "SNN_CASES" in os.environ else [(8, 1, "el+AMPA"), (16, 1, "el+AMPA"), (24, 1, "el+AMPA"), (32, 1, "el+AMPA"), (32, 3, "el+AMPA+NMDA+GABA"), (16, 0, "el only"), (32, 0, "el only")]
for side, types, what in CASES:
    for persistent in (1, 0):
        n = side * side
        dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
        dn.add_lattice(0, side, side)
        dn.finalize()
        dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
        dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
        if types:
            flags = np.zeros((n, 3), np.uint32)
            flags[:, :max(types, 0)] = 1                      # (types < 0: chemical synapses on, nobody releases anything)
            dn.set_attr(0, "neurotransmitters$flags", flags)
            dn.set_attr(0, "receptors$flags", flags)
        dn.set_synapses(True, types != 0)
        dn.set_option("persistent_chem" if types else "persistent_run", persistent)
        dn.set_reduced_history(False, False, True)
        dn.run(200)
        reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            dn.run(steps)
            reps.append((time.perf_counter() - t0) / steps * 1e6)
        spikes = int(dn.spike_counts(0).sum())
        phases = None
        if dn.stat("persistent_run_launches"):
            # shader-clock totals of workgroup 0's four phases over one more launch (option "run_timing"), as shares of a step
            dn.set_option("run_timing", 1)
            dn.run(steps)
            n = max(1, dn.stat("run_timing_steps"))
            clocks = {k: dn.stat("run_timing_" + k) / n for k in ("poll", "barrier", "turns", "update")}
            tot = sum(clocks.values()) or 1.0
            phases = {k: sorted(reps)[2] * v / tot for k, v in clocks.items()}
        print(json.dumps({"lattice": f"{side}x{side}", "synapses": what, "one_launch_run": bool(dn.stat("persistent_run_launches")),
                          "us_per_step": sorted(reps)[2], "us_per_step_runs": reps, "steps": steps,
                          "spikes_per_step": spikes / (200 + 5 * steps), "phases_us_per_step": phases, "fallbacks": dn.stat("persistent_run_fallbacks")}), flush=True)
        dn.close()

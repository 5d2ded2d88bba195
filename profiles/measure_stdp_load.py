#!/usr/bin/env python3
"""STDP under load on BASELINE configs[3] (81 920 neurons, 26.8 GB matrix): ms per step, average input-pass launch and
plasticity launches per step for a driven spike fraction f, with the weight update (a) riding on the next input pass
(SNN_AMD_DEFER_STDP=1), (b) as the standalone scatter kernels that evaluate STDP per synapse (=0), (b') as scatter passes
that add the two prepared delta vectors (=2) and (c) riding but with a_plus = a_minus = 0
(no word changes: the cost of the update path without its stores).  Usage: measure_stdp_load.py [steps] [mode,mode...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import snn_amd  # noqa: E402
from snn_amd import synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_inh, n_exc = 128 * 128, 256 * 256
n = n_inh + n_exc
rows = []
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fused", "standalone", "prepared_scatter", "fused_zero_delta"]
for mode in modes:
    os.environ["SNN_AMD_DEFER_STDP"] = {"standalone": "0", "prepared_scatter": "2"}.get(mode, "1")
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
    dn.add_lattice(0, 128, 128)
    dn.add_lattice(1, 256, 256)
    dn.finalize()
    for i, m in ((0, n_inh), (1, n_exc)):
        dn.set_attr(i, "gap_conductance", np.full(m, 10.0, np.float32))
    dn.fill_graph_synthetic(5, 0.5, 1.5, with_diagonal=False)
    a = 0.0 if mode == "fused_zero_delta" else 2.0
    dn.set_plasticity(0, a_plus=a, a_minus=a)
    dn.set_plasticity(1, a_plus=a, a_minus=a)
    dn.set_reduced_history(False, False, True)
    for f in (0.0, 0.001, 0.01):
        dn.set_attr(0, "current_voltage", synthetic.uniform(4, n_inh, -65.0, 30.0))
        dn.set_attr(1, "current_voltage", synthetic.uniform(4, n_exc, -65.0, 30.0, offset=n_inh))
        dn.set_synthetic_drive(12345, f, 35.0)
        dn.run(150 if f else 5)          # let most neurons have a firing time on record before timing
        s0 = sum(int(dn.spike_counts(i).sum()) for i in (0, 1))
        dn.profile_enable(True)
        dn.profile_reset()
        t0 = time.perf_counter()
        dn.run(steps)
        dt = time.perf_counter() - t0
        launches, kern_ms = dn.profile_read()
        pl_steps, pl_ms = dn.profile_read_plasticity()
        dn.profile_enable(False)
        spikes = (sum(int(dn.spike_counts(i).sum()) for i in (0, 1)) - s0) / steps
        row = {"mode": mode, "spike_fraction": f, "spikes_per_step": spikes, "ms_per_step": dt / steps * 1e3,
               "input_pass_ms": kern_ms / max(1, launches), "plasticity_launches_ms_per_step": pl_ms / max(1, pl_steps)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    dn.close()

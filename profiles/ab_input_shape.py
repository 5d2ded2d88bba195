#!/usr/bin/env python3
"""A/B of the two streamed shapes of the dense input pass on ONE handle (same matrix placement), alternating:
ab_input_shape.py [c2|c3|<side>] [rounds] [steps]  (1: 4 columns per lane, 2: two)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import snn_amd  # noqa: E402
from snn_amd import synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
if cfg == "c3":
    n = 128 * 128
    dn = snn_amd.DeviceNetwork(model=snn_amd.HODGKIN_HUXLEY, nt_kinetics=snn_amd.NT_DESTEXHE, receptor_kinetics=snn_amd.RC_DESTEXHE)
    dn.add_lattice(0, 128, 128)
    dn.finalize()
    dn.set_attr(0, "current_voltage", synthetic.uniform(3, n, -70.0, -60.0))
    flags = np.zeros((n, 3), np.uint32)
    flags[:, 0] = 1
    dn.set_attr(0, "neurotransmitters$flags", flags)
    dn.set_attr(0, "receptors$flags", flags)
    dn.fill_graph_synthetic(4, 0.5, 1.5, with_diagonal=False)
    dn.set_synapses(True, True)
else:
    side = int(cfg) if cfg.isdigit() else 256
    n = side * side
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
    dn.add_lattice(0, side, side)
    dn.finalize()
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
    dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
    dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
dn.run(10)
res = {1: [], 2: []}
for r in range(rounds):
    for shape in (1, 2):
        dn.set_option("input_shape", shape)
        dn.run(5)
        dn.profile_enable(True)
        dn.profile_reset()
        dn.run(steps)
        launches, ms = dn.profile_read()
        dn.profile_enable(False)
        res[shape].append(ms / launches)
print(json.dumps({"config": cfg, "input_pass_ms": {str(k): [round(x, 5) for x in v] for k, v in res.items()}}))
dn.close()

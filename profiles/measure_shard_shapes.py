#!/usr/bin/env python3
"""Per-rank kernel time of the sharded C2 step on ONE GPU: a shard handle (shard G/2 of G) of the 256x256 lattice holds
the columns one rank of G would hold; the input pass is timed as ONE launch over all presynaptic rows and as the split
pair (own-rows chunks first, the rest later -- what snn_run_sharded enqueues around the collective).  Predicts the
compute side of the strong-scaling curve (no exchange here).  Usage: measure_shard_shapes.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import snn_amd  # noqa: E402
from snn_amd import synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = cols = 256
n = rows * cols
for g in ([int(os.environ['SHARDS'])] if os.environ.get('SHARDS') else (1, 2, 4, 8)):
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH)
    dn.add_lattice(0, rows, cols)
    dn.finalize(g // 2, g)
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
    dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
    dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
    out = {"n_shards": g, "columns": dn.post_end - dn.post_begin, "matrix_GB": 4.0 * n * (dn.post_end - dn.post_begin) / 1e9}
    for mode in ("one_pass", "split"):
        for _ in range(5):
            if mode == "split":
                dn.step_begin_local()
            dn.step_begin()
            dn.step_end()
        dn.profile_enable(True)
        dn.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            if mode == "split":
                dn.step_begin_local()
            dn.step_begin()
            dn.step_end()
        dn.synchronize()
        dt = time.perf_counter() - t0
        launches, ms = dn.profile_read()
        dn.profile_enable(False)
        out[mode] = {"input_pass_ms": ms / max(1, launches), "GBps": out["matrix_GB"] / (ms / max(1, launches)) * 1e3,
                     "host_loop_ms_per_step": dt / steps * 1e3}
    # the library's own step loop with a transport that moves nothing (snn_run_sharded_custom + snn_exchange_noop): what ONE
    # rank's step costs besides its exchange -- kernels, launches and the gaps between them, no host call per step
    dn.run_sharded_without_exchange(10)
    t0 = time.perf_counter()
    dn.run_sharded_without_exchange(steps)
    out["library_loop_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps(out), flush=True)
    dn.close()

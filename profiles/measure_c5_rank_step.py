#!/usr/bin/env python3
"""What ONE rank of G does per C5 step (configs[4] sharded by lattice), timed on one GPU: the shard handle of rank G/2
runs the library's step loop with a do-nothing exchange (snn_run_sharded_custom + snn_exchange_noop; the halo contents are then stale, the
timing is not affected).  SHARDS=8 in the environment measures that one split only.  Shows how far the per-rank step is from its kernel time, i.e. how launch-bound the small
sparse step is.  Usage: measure_c5_rank_step.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import snn_amd  # noqa: E402
from snn_amd import synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
side = 512
m = side * side
for g in ([int(os.environ['SHARDS'])] if os.environ.get('SHARDS') else (1, 2, 4, 8)):
    dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, spike_train=snn_amd.ST_POISSON)
    for k in range(4):
        dn.add_lattice(k, side, side)
        dn.add_spike_train_lattice(4 + k, side, side)
    dn.finalize(g // 2, g, csr=True, by_lattice=True)
    for k in range(4):
        dn.set_attr(k, "gap_conductance", np.full(m, 10.0, np.float32))
        dn.set_attr(k, "current_voltage", synthetic.uniform(6, m, -65.0, 30.0, offset=k * m))
        dn.set_attr(4 + k, "chance_of_firing", np.full(m, 0.01, np.float32))
        dn.set_attr(4 + k, "seed", np.arange(k * m + 1, (k + 1) * m + 1, dtype=np.uint32))
    dn.set_graph_csr(*synthetic.c5_csr(side, posts=dn.owned))
    # the halo plan this rank would have: what it reads of every peer; as send lists take the mirror image of the
    # rank's own needs (same sizes by symmetry of the lattice structure)
    for p in range(g):
        if p != g // 2:
            need = dn.halo_needs(p)
            own = dn.owned
            dn.halo_set_sends(p, own[:need.size] if need.size <= own.size else own)
    if g > 1:
        dn.halo_commit()
    plan = dn.exchange_plan() if g > 1 else None
    if g > 1 and os.environ.get("PEER"):
        # the peer form, looped back: what this rank would store into its neighbours goes into its OWN receive sets (the
        # mirror-image lists have the same sizes), and its own done counter stands in for theirs -- one launch per step, the
        # granules and counters travel through the same fine-grained memory a neighbour's would
        me = g // 2
        loc = dn.p2p_local()
        for p in range(g):
            if p != me and loc["counts"][p]:
                dn.p2p_connect(p, loc["recv"][0], loc["recv"][1], loc["flags"] + 4 * (p - me), loc["offsets"][p])
        dn.p2p_commit()
    dn.run_sharded_without_exchange(50)
    t0 = time.perf_counter()
    dn.run_sharded_without_exchange(steps)
    dt = time.perf_counter() - t0
    print(json.dumps({"n_shards": g, "owned_neurons": int(dn.owned.size), "us_per_step": dt / steps * 1e6,
                      "form": "peer (one launch per step, looped back)" if dn.stat("halo_peer_steps") else "two launches per step",
                      "recv_bytes_per_step": (4 * int(plan["recv_words"]) if plan else 0)}), flush=True)
    dn.close()

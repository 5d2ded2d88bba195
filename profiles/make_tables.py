#!/usr/bin/env python3
"""Markdown tables of a round's collection:  python profiles/make_tables.py profiles/r03
(per config: the kernels of rocprofv3's stats, the dominant kernel's rate against its algorithmic bytes, the PMC
traffic, the bench line of the profiled process)."""
import csv
import json
import os
import sys

d = sys.argv[1]
PEAK = 8000.0


def stats(name):
    path = os.path.join(d, name + "_kernel_stats.csv")
    if not os.path.exists(path):
        return []
    return [(r["Name"].replace("void ", "").replace("snn::", "").split("(")[0], int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(path))]


def line(name):
    path = os.path.join(d, name + "_bench_under_rocprof.json")
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None


def pmc(name):
    path = os.path.join(d, name + "_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get("dominant_kernel")


print("| config | kernels (rocprofv3 average per launch) | algorithmic bytes / launch | achieved (of 8 TB/s) | PMC traffic / algorithmic | whole step (bench line of the profiled process) |")
print("|---|---|---|---|---|---|")
for name in ("c2", "c3", "c4", "c4_spiking_0p1pct", "c6", "c5", "c1", "lattice64", "c2_sharded_world1", "c5_sharded_world1"):
    st, b = stats(name), line(name)
    if not st or b is None:
        continue
    rf = b["roofline"]
    kernels = "; ".join(f"`{k}` {a / 1e3:.1f} µs × {c}" for k, c, a in st[:5] if a * c > 0.005 * sum(x[1] * x[2] for x in st))
    dom = next(((k, c, a) for k, c, a in st if k.startswith(("k_inputs", "k_step_csr", "k_run_resident"))), st[0])
    alg = rf.get("algorithmic_bytes_per_launch") or rf.get("matrix_bytes_read_once_per_run")
    if rf.get("bound") == "hbm" and alg:
        rate = alg / dom[2]
        ach = f"{rate / 1e3:.2f} TB/s = {rate / PEAK:.3f}"
    else:
        ach = "latency-bound (§4.4)" if rf.get("bound") == "latency" else "–"
    p = pmc(name)
    traffic = f"{p['hbm_traffic_bytes_per_launch'] / 1e9:.3f} GB / {alg / 1e9:.3f} GB = {p['hbm_traffic_bytes_per_launch'] / alg:.3f}" if p and alg else "–"
    step = f"{b['ms_per_step'] * 1e3:.1f} µs" if b["ms_per_step"] < 1 else f"{b['ms_per_step']:.3f} ms"
    print(f"| {name} | {kernels} | {alg / 1e9:.3f} GB | {ach} | {traffic} | {step} = {b['value'] / 1e6:.1f} M neuron-steps/s, {b['spikes_per_step']:.0f} spikes/step |")

#!/usr/bin/env python3
"""Kernel durations and the idle time between consecutive kernels of one rocprofv3 --kernel-trace run.
Usage: kernel_gaps.py <..._kernel_trace.csv> [skip_first_n]   -> one JSON object"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
dur = defaultdict(list)
gap_after = defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = a["Kernel_Name"].split("(")[0][:60] + " grid " + a.get("Grid_Size", a.get("Grid_Size_X", "?"))
    dur[name].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap_after[name].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2]
out = {k: {"calls": len(v), "median_ns": med(v), "median_gap_to_next_ns": med(gap_after[k])} for k, v in dur.items()}
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(json.dumps({"kernels": out, "span_ns": span, "launches": len(rows), "ns_per_launch": span / len(rows)}, indent=1))

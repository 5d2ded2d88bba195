#!/usr/bin/env python3
"""Reduce rocprofv3 `--pmc` counter_collection.csv files to a per-kernel summary (mean per dispatch) and,
for the FETCH_SIZE / WRITE_SIZE pair, the corrected HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the read
side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The two counters do not fit one
pass (TCC slots), hence two runs of the same command.

usage: summarize_pmc.py FETCH.csv WRITE.csv OUT.json [kernel-substring] [collected-date]
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "k_inputs_dense"
    collected = sys.argv[5] if len(sys.argv) > 5 else None
    f, w = per_kernel(fetch), per_kernel(write)
    rows = []
    for k in sorted(set(f) | set(w)):
        nf, vf = f.get(k, (0, 0.0))
        nw, vw = w.get(k, (0, 0.0))
        rows.append({"kernel": k, "dispatches": max(nf, nw), "FETCH_SIZE_KiB_mean": vf, "WRITE_SIZE_KiB_mean": vw,
                     "hbm_read_bytes_corrected": 2.0 * vf * 1024.0, "hbm_write_bytes": vw * 1024.0,
                     "hbm_traffic_bytes_per_launch": 2.0 * vf * 1024.0 + vw * 1024.0})
    # the variant the step loop launches (most dispatches); the placement timing at finalize uses another one
    dom = sorted((r for r in rows if needle in r["kernel"]), key=lambda r: -r["dispatches"])
    json.dump({"collected": collected, "correction": "read = 2 * FETCH_SIZE * 1024 (gfx950 wide-stream under-count), write = WRITE_SIZE * 1024",
               "dominant_kernel": dom[0] if dom else None, "kernels": rows}, open(out, "w"), indent=1)
    if dom:
        print(json.dumps(dom[0], indent=1))


if __name__ == "__main__":
    main()

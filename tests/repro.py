"""Repro bundles of the randomized differential tests (test infrastructure only).

A bit-exact contract has no "flaky" outcome: when a device result differs from the oracle the test must leave behind
everything needed to say WHICH side was wrong and WHERE -- the seed and the switches it drew, the form every step took
(snn_get_stat), both arrays and the first differing index -- and it repeats both sides once more in the same process
(`triage`), because a mismatch that does not repeat is a race, and one that does is a defect of the code.

Bundles go to $SNN_REPRO_DIR (default gpurun_out/repro, which travels back from the GPU box): one .npz with every compared
array of both sides and one .json with the metadata."""
import json
import os
import time

import numpy as np

import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = ("persistent_run_launches", "persistent_run_steps", "persistent_run_stdp_steps", "persistent_run_fallbacks", "halo_direct_steps",
         "steps_dense_one_launch", "steps_sparse_one_launch", "steps_sparse_split", "steps_two_kernel",
         "shadow_refreshes", "view_refreshes", "history_regrows")


def repro_dir():
    d = os.environ.get("SNN_REPRO_DIR", os.path.join(ROOT, "gpurun_out", "repro"))
    os.makedirs(d, exist_ok=True)
    return d


def device_stats(dn):
    return {name: dn.stat(name) for name in STATS}


def differences(obs, ref):
    """[{name, mismatches, first, oracle, device}] over the keys of `ref`, floats compared as parity.bits"""
    out = []
    for name, want in ref.items():
        got = obs.get(name)
        if got is None:
            out.append({"name": name, "mismatches": -1, "first": None, "oracle": None, "device": "missing"})
            continue
        want, got = np.asarray(want), np.asarray(got)
        if want.shape != got.shape:
            out.append({"name": name, "mismatches": -1, "first": None, "oracle": list(want.shape), "device": list(got.shape)})
            continue
        bw, bg = parity.bits(want), parity.bits(got)
        if np.array_equal(bw, bg):
            continue
        idx = np.argwhere(bw != bg)
        first = tuple(int(x) for x in idx[0])
        out.append({"name": name, "mismatches": int(len(idx)), "first": list(first),
                    "oracle": repr(want[first] if first else want[()]), "device": repr(got[first] if first else got[()]),
                    "rows": sorted({int(i[0]) for i in idx})[:16] if idx.shape[1] else []})
    return out


def describe(diffs):
    return "; ".join(f"{d['name']}: {d['mismatches']} mismatches, first at {d['first']}: oracle={d['oracle']} device={d['device']}"
                     for d in diffs)


def dump(tag, meta, obs=None, ref=None, extra_arrays=None):
    """writes <tag>-<pid>-<time>.json/.npz and returns the json path"""
    base = os.path.join(repro_dir(), f"{tag}-{os.getpid()}-{int(time.time() * 1000)}")
    arrays = {}
    for side, d in (("device", obs), ("oracle", ref)):
        for name, a in (d or {}).items():
            arrays[f"{side}/{name}"] = np.asarray(a)
    for name, a in (extra_arrays or {}).items():
        arrays[name] = np.asarray(a)
    if arrays:
        np.savez_compressed(base + ".npz", **arrays)
    with open(base + ".json", "w") as f:
        json.dump(meta, f, indent=1, default=str)
    return base + ".json"


def triage(run_device, run_oracle, ref, repeats=2):
    """Both sides once more, in this process: {"oracle_repeats": bool, "device_repeat_diffs": [n mismatching arrays]}.
    run_device() / run_oracle() rebuild their network from the seed."""
    out = {}
    ref2 = run_oracle()
    out["oracle_differs_from_its_first_run"] = describe(differences(ref2, ref))
    out["device_repeats_against_oracle"] = []
    for _ in range(repeats):
        obs, stats = run_device()
        out["device_repeats_against_oracle"].append({"diffs": describe(differences(obs, ref)), "stats": stats})
    return out

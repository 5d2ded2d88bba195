"""Localisation of a device / oracle mismatch (test infrastructure; round 5).

The randomized differential tests used to compare at the END of a sequence of calls: a mismatch said that something, somewhere in
a few hundred steps, had gone wrong on one of the two sides.  `Tracker` compares at EVERY run-call boundary instead and keeps, from
just before the call, a checkpoint of the device handle (snn_debug_checkpoint: every device array + the stepper's host cursors)
and a copy of the oracle container.  When a boundary differs the call that diverged is known, and `localise` executes it again:

  oracle       from the copy, several times              -> does the oracle reproduce ITS result of the failing run?
  same handle  from the device checkpoint, many times    -> does the handle reproduce the failing result?  (a wrong result that
               repeats is hidden state of the handle that was already wrong; one that does not is a transient of the device)
  fresh handle built from the oracle copy, in-process    -> state pushed through the C ABI into new device memory
  child        the same in a fresh PROCESS               -> no history of thousands of handles, a young heap
  step by step from the checkpoint against the oracle    -> the first step and the first neurons that differ

and writes everything into a repro bundle (tests/repro.py) whose verdict line says which side, which call, deterministic or not.
Also here: the host-side poison switch of the campaigns and the reading of the GPU's RAS / ECC error counters."""
import copy
import glob
import json
import os
import pickle
import subprocess
import sys
import time

import numpy as np

import parity
import repro

HERE = os.path.dirname(os.path.abspath(__file__))
REPEATS = int(os.environ.get("SNN_LOCALISE_REPEATS", "200"))


def clone(net):
    """an independent copy of an oracle container (arrays, clocks, layout)"""
    n = copy.deepcopy(net)
    n.__dict__.pop("run", None)         # (a test may have wrapped the method of ITS container: the copy runs itself)
    return n


def state_diffs(net, dev, graph=None):
    """differences between a pulled device state (+ optional graph weights) and the oracle container, as repro.differences"""
    ref = {k: net[k] for k in dev}
    obs = dict(dev)
    if graph is not None:
        obs["graph"], ref["graph"] = graph
    return repro.differences(obs, ref)


def pull_all(dn, net):
    """device state + the weights in the form the handle holds them, and the oracle's view of those weights"""
    dev = parity.pull_state(dn, net)
    graph = None
    if net.n_tot and net.n_neurons:
        if getattr(dn, "csr", False):
            graph = (dn.get_graph_csr(), parity.csr_for_posts(net, dn.owned)[2])
        else:
            oc = net["connections"].astype(np.uint32)
            graph = (dn.get_graph_rows(0, net.n_tot)[0], np.where(oc != 0, net["weights"], np.float32(0)))
    return dev, graph


def same_state(a, b):
    return all(np.array_equal(parity.bits(a[k]), parity.bits(b[k])) for k in a)


def set_clocks(dn, net):
    """a handle built from an oracle container that has already run continues at the container's clocks"""
    dn.set_clock(int(net.clock))
    for slot, (i, _, _) in enumerate(net.layout.st_lattices):
        dn.set_spike_train_clock(i, int(net["st_clock"][slot]))


def fresh_handle(snn, net, plan, options):
    from test_gpu_randomized import make_handle
    dn = make_handle(snn, net, plan)
    set_clocks(dn, net)
    for name, value in options.items():
        dn.set_option(name, value)
    return dn


class Tracker:
    """One device handle and its oracle container through a sequence of calls.  `run(k)` replaces dn.run(k); net.run(k)."""

    def __init__(self, snn, dn, net, plan, tag, options=None, log=None, enabled=True):
        self.snn, self.dn, self.net, self.plan, self.tag = snn, dn, net, plan, tag
        self.options = dict(options or {})
        self.log = log if log is not None else []
        self.calls = 0
        # (the comparison before a call is a getter: it flushes what a run call leaves pending -- callers keep some seeds without it)
        self.enabled = enabled and os.environ.get("SNN_CHECKPOINTS", "1") != "0" and plan.get("rewards") is None

    def set_option(self, name, value):
        self.dn.set_option(name, value)
        self.options[name] = value

    def run(self, k, oracle_run=None, **oracle_kw):
        """k steps on both sides with a comparison before and after; a difference is localised and raised"""
        dn, net = self.dn, self.net
        oracle_run = oracle_run or net.run
        if not self.enabled:
            dn.run(k)
            oracle_run(k, **oracle_kw)
            return
        self.calls += 1
        dev, graph = pull_all(dn, net)
        pre = state_diffs(net, dev, graph)
        if pre:
            self._report("the state differed BEFORE this run call: a call since the previous run diverged (setters, switches)",
                         k, None, dev, pre)
        before = clone(net)
        dn.checkpoint()
        dn.run(k)
        oracle_run(k, **oracle_kw)
        dev, graph = pull_all(dn, net)
        post = state_diffs(net, dev, graph)
        if dn.stat("verify_mismatches"):
            self._report("the device disagreed with ITSELF (option verify): " + dn.verify_report(), k, before, dev, post)
        if post:
            self._report("this run call diverged", k, before, dev, post)

    # ---- localisation --------------------------------------------------------------------------------------------------------
    def _report(self, what, k, before, dev_bad, diffs):
        dn, net = self.dn, self.net
        meta = {"tag": self.tag, "what": what, "call": self.calls, "steps_of_the_call": k, "log": list(self.log), "options": self.options,
                "plan": {a: b for a, b in self.plan.items() if a != "rewards"}, "differences": diffs,
                "stats": repro.device_stats(dn), "verify_report": dn.verify_report(), "pid": os.getpid(),
                "environment": {a: b for a, b in os.environ.items() if a.startswith(("SNN_", "AMD_", "HIP_", "HSA_", "OMP_", "MALLOC_", "GPU_"))}}
        orc_bad = {name: np.array(net[name], copy=True) for name in dev_bad}
        verdict = what
        if before is not None:
            try:
                loc = localise(self.snn, dn, before, self.plan, self.options, k, dev_bad, orc_bad)
                meta["localisation"] = loc
                verdict = f"{what}; {loc['verdict']}"
            except BaseException as e:           # noqa: BLE001 -- the bundle matters more than the localiser's own trouble
                meta["localisation_error"] = repr(e)
        meta["ras_after"] = ras_counters()
        meta["verdict"] = verdict
        path = repro.dump(f"{self.tag}-call{self.calls}", meta, dev_bad, orc_bad)
        raise AssertionError(f"{self.tag}, run call {self.calls} ({k} steps): {verdict}; {repro.describe(diffs)[:1500]}; "
                             f"sequence: {self.log} (bundle: {path})")


def localise(snn, dn, before, plan, options, k, dev_bad, orc_bad, repeats=None):
    """re-executes ONE diverging run call from identical inputs on both sides (see the module docstring)"""
    repeats = REPEATS if repeats is None else repeats
    names = list(dev_bad)
    out = {}

    def oracle_once():
        n = clone(before)
        n.run(k)
        return {name: np.array(n[name], copy=True) for name in names}

    # (1) the oracle again
    orc = [oracle_once() for _ in range(5)]
    out["oracle_reproduces_its_failing_result"] = [same_state(o, orc_bad) for o in orc]
    out["oracle_replay_equals_failing_device"] = [same_state(o, dev_bad) for o in orc]
    truth = orc[0]

    def tally(results):
        return {"runs": len(results), "equal_to_failing_device": sum(same_state(r, dev_bad) for r in results),
                "equal_to_oracle_replay": sum(same_state(r, truth) for r in results),
                "other": sum(not same_state(r, dev_bad) and not same_state(r, truth) for r in results)}

    # (2) the same handle from its checkpoint
    same = []
    for _ in range(repeats):
        dn.restore_checkpoint()
        dn.run(k)
        same.append(parity.pull_state(dn, before))
    out["same_handle"] = tally(same)
    # (3) step by step from the checkpoint
    dn.restore_checkpoint()
    walker = clone(before)
    out["first_differing_step"] = None
    for step in range(k):
        dn.run(1)
        walker.run(1)
        d = state_diffs(walker, parity.pull_state(dn, walker))
        if d:
            out["first_differing_step"] = {"step": step + 1, "differences": d}
            break
    # (4) fresh handles in this process
    fresh = []
    for _ in range(max(1, repeats // 4)):
        h = fresh_handle(snn, clone(before), plan, options)
        h.run(k)
        fresh.append(parity.pull_state(h, before))
        h.close()
    out["fresh_handle"] = tally(fresh)
    # (5) a fresh process
    out["child_process"] = child_replay(before, plan, options, k, dev_bad, truth, max(1, repeats // 4))
    s, f = out["same_handle"], out["fresh_handle"]
    if not all(out["oracle_reproduces_its_failing_result"]):
        verdict = "the ORACLE does not reproduce its own result of the failing run: the checker (or its host memory) was transiently wrong"
    elif s["equal_to_failing_device"] == s["runs"]:
        verdict = ("the handle REPRODUCES the failing result from its checkpoint every time: state of the handle that the C ABI does not "
                   "show was already wrong before the call (deterministic)" if f["equal_to_oracle_replay"] == f["runs"] else
                   "handle and fresh handles both reproduce a result the oracle does not give: a deterministic defect of the stepper or the oracle")
    elif s["equal_to_failing_device"] == 0 and s["equal_to_oracle_replay"] == s["runs"]:
        verdict = "the handle gives the ORACLE's result every time when the call is executed again from the checkpoint: the failing execution was a transient of the device side"
    else:
        verdict = f"the handle is NOT deterministic from its checkpoint: {s}"
    out["verdict"] = verdict + f" [same handle {s}, fresh handles {f}, child {out['child_process']}]"
    return out


def child_replay(before, plan, options, k, dev_bad, truth, repeats):
    """the call once more in a FRESH process, from the oracle copy pushed through the C ABI"""
    base = os.path.join(repro.repro_dir(), f"replay-{os.getpid()}-{int(time.time() * 1000)}")
    try:
        with open(base + ".pkl", "wb") as f:
            pickle.dump({"net": before, "plan": plan, "options": options, "k": k, "dev_bad": dev_bad, "truth": truth, "repeats": repeats}, f)
    except Exception as e:                      # noqa: BLE001 (containers of generated models do not pickle)
        return {"skipped": repr(e)}
    env = dict(os.environ)
    env.pop("SNN_AMD_VERIFY", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--replay", base + ".pkl"], capture_output=True, text=True, timeout=900, env=env)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:                           # noqa: BLE001
        return {"failed": r.returncode, "stderr": r.stderr[-1500:]}


def _replay_main(path):
    import conftest  # noqa: F401
    import snn_amd
    snn_amd._lib.load()
    with open(path, "rb") as f:
        job = pickle.load(f)
    results = []
    for _ in range(job["repeats"]):
        h = fresh_handle(snn_amd, clone(job["net"]), job["plan"], job["options"])
        h.run(job["k"])
        results.append(parity.pull_state(h, job["net"]))
        h.close()
    print(json.dumps({"runs": len(results), "equal_to_failing_device": sum(same_state(r, job["dev_bad"]) for r in results),
                      "equal_to_oracle_replay": sum(same_state(r, job["truth"]) for r in results)}))


# ---- the GPU's error counters ------------------------------------------------------------------------------------------------
def ras_counters():
    """RAS / ECC error counts of the GPUs as far as an ordinary user may read them: the amdgpu sysfs counters
    (/sys/class/drm/card*/device/ras/*_err_count: 'ue: n' / 'ce: n' lines) and, when the tools answer, rocm-smi / amd-smi"""
    out = {"sysfs": {}}
    for path in sorted(glob.glob("/sys/class/drm/card*/device/ras/*err_count")):
        try:
            out["sysfs"][path[len("/sys/class/drm/"):]] = open(path).read().strip().replace("\n", "; ")
        except OSError as e:
            out["sysfs"][path] = f"unreadable: {e.strerror}"
    for name, cmd in (("rocm-smi", ["/opt/rocm/bin/rocm-smi", "--showrasinfo", "all"]),
                      ("amd-smi", ["/opt/rocm/bin/amd-smi", "metric", "--ecc", "--json"])):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=40)
            out[name] = (r.stdout + r.stderr)[-4000:]
        except Exception as e:                  # noqa: BLE001
            out[name] = f"unavailable: {e!r}"
    return out


def ras_totals(c):
    """(uncorrectable, correctable) error totals: amd-smi's ECC totals when it answered, else the sysfs counters"""
    try:
        data = json.loads(c.get("amd-smi", ""))
        gpus = data.get("gpu_data", data) if isinstance(data, dict) else data
        return (sum(int(g["ecc"]["total_uncorrectable_count"]) + int(g["ecc"].get("total_deferred_count", 0)) for g in gpus),
                sum(int(g["ecc"]["total_correctable_count"]) for g in gpus))
    except Exception:                           # noqa: BLE001
        pass
    ue = ce = 0
    for text in c.get("sysfs", {}).values():
        for part in text.replace(";", "\n").splitlines():
            part = part.strip()
            if part.startswith("ue:"):
                ue += int(part[3:].strip() or 0)
            elif part.startswith("ce:"):
                ce += int(part[3:].strip() or 0)
    return ue, ce


if __name__ == "__main__":
    for p in (os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    if len(sys.argv) == 3 and sys.argv[1] == "--replay":
        _replay_main(sys.argv[2])
    elif len(sys.argv) == 2 and sys.argv[1] == "--ras":
        print(json.dumps(ras_counters(), indent=1))

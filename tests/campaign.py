"""Contention campaign for the randomized differential tests (test infrastructure; runs on the GPU box).

    python3 tests/campaign.py --minutes 20 --workers 6 --streamers 2 --out gpurun_out/campaign_a \\
        --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection

Reproduces deliberately what the long pytest-xdist campaigns of round 3 did by accident: several PROCESSES on one GPU.
`workers` processes each walk their own stripe of seeds through the listed seeded test functions (called directly, the
`snn` fixture is the package); `streamers` processes keep the device busy with C2-shaped input passes over a matrix of a few
GiB, so that the workers' launches queue behind long kernels and the one-launch forms have to share the CUs.  Every
mismatch leaves a repro bundle (tests/repro.py) under <out>/repro; <out>/summary.json holds executions and failures per test
function.  One line per worker and minute goes to stdout."""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def small_dense_case(seed):
    """test_random_network cases of the class both round-3 failures fell in: one handle, dense graph, one or few workgroups"""
    import test_gpu_randomized as t
    net, plan = t.draw(1000 + seed)
    return plan["shards"] == 1 and not plan["csr"] and net.n_neurons + net.n_cells <= 256


FILTERS = {"small_dense": small_dense_case}


def seed_is_valid(spec, seed):
    """seeds a test function is defined for (its own parametrisation filters them; called directly, the campaign has to)"""
    if spec == "test_gpu_sequences:test_random_call_sequence":
        import test_gpu_sequences
        return test_gpu_sequences.usable(seed)
    if spec == "test_gpu_randomized:test_random_network":
        # (cases that step shard handles by one host thread per rank need GPU_MAX_HW_QUEUES=24 -- tests/test_gpu_emulated_ranks.py;
        # the campaign's workers keep the runtime's default queues, the mapping a one-rank-per-process run has)
        import test_gpu_randomized
        return not test_gpu_randomized.threaded(seed)
    return True


def install_oracle_guard(snn_amd, record):
    """The oracle's arrays are written by the oracle and by the test, never while a call into the device library is under way (the
    tests are single-threaded).  Every public DeviceNetwork method is wrapped: the arrays of all live oracle containers are copied
    (CRC for the large ones) before the call and compared after it.  A difference is memory of THIS process that the library, the
    HIP runtime or one of its threads wrote behind the test's back -- the kind of event that makes device and oracle disagree once
    and never again -- and `record` gets the method, the array, the first index and both values."""
    import weakref
    import zlib
    import numpy as np
    import oracle_binding as ob
    nets = weakref.WeakSet()
    init = ob.Net.__init__

    def tracked_init(self, *a, **k):
        init(self, *a, **k)
        nets.add(self)
    ob.Net.__init__ = tracked_init

    def digest(a):
        # (a sum over 8-byte words, numpy speed: CRC32 at 1 GB/s made campaign H four times slower than its tests)
        b = a.reshape(-1).view(np.uint8)
        n8 = b.size & ~7
        return (int(b[:n8].view(np.uint64).sum(dtype=np.uint64)), zlib.crc32(b[n8:]))

    def picture():
        out = {}
        for n in list(nets):
            items = list(n.arr.items()) + [(h, getattr(n, h, None)) for h in ("voltage_history", "spike_history", "st_voltage_history")]
            for name, a in items:
                if isinstance(a, np.ndarray) and a.size and a.flags["C_CONTIGUOUS"]:
                    out[(id(n), name)] = (a, a.copy() if a.nbytes <= (1 << 16) else None, digest(a))
        return out

    def wrap(name, fn):
        def guarded(self, *a, **k):
            before = picture()
            try:
                return fn(self, *a, **k)
            finally:
                for key, (arr, copy, was_digest) in before.items():
                    if digest(arr) == was_digest:
                        continue
                    rec = {"oracle_memory_changed_during": name, "array": key[1], "address": hex(arr.ctypes.data), "bytes": int(arr.nbytes)}
                    if copy is not None:
                        was, now = copy.reshape(-1).view(np.uint8), arr.reshape(-1).view(np.uint8)
                        idx = np.flatnonzero(was != now)
                        rec.update({"changed_bytes": int(idx.size), "first_byte": int(idx[0]), "last_byte": int(idx[-1]),
                                    "was": was[idx[0]:idx[0] + 16].tolist(), "now": now[idx[0]:idx[0] + 16].tolist()})
                    record(rec)
        guarded.__name__ = name
        return guarded
    cls = snn_amd.DeviceNetwork
    for name, fn in list(vars(cls).items()):
        if callable(fn) and (not name.startswith("_") or name == "__init__") and not isinstance(fn, (staticmethod, classmethod, property)):
            setattr(cls, name, wrap(name, fn))
    return nets.clear            # called when a test starts: only the containers of the running test are watched


def worker(args):
    import conftest  # noqa: F401  (OpenMP settings of the oracle before libgomp loads)
    os.environ["SNN_CAMPAIGN"] = "1"
    os.environ["SNN_REPRO_DIR"] = os.path.join(args.out, "repro")
    import snn_amd
    snn_amd._lib.load()
    fns = []
    for spec in args.tests.split(","):
        mod, name = spec.split(":")
        fns.append((spec, getattr(importlib.import_module(mod), name)))
    accept = FILTERS.get(args.filter)
    deadline = time.time() + args.minutes * 60
    counts = {spec: [0, 0] for spec, _ in fns}            # executions, failures
    log = open(os.path.join(args.out, f"worker-{args.index}.jsonl"), "a")
    seed, last_report, recent = args.first_seed + args.index, time.time(), []
    # what the device's self-check (SNN_AMD_VERIFY) prints goes to a file per worker too, so that every report can be tied to the test
    # and seed that was running (a report does not fail a test by itself: the handle goes on with the outcome that repeated)
    vlog = os.path.join(args.out, f"worker-{args.index}.verify.log")
    os.environ["SNN_AMD_VERIFY_LOG"] = vlog
    vseen = os.path.getsize(vlog) if os.path.exists(vlog) else 0
    current = [None, None]
    forget = lambda: None
    if not args.plain and not args.lean:
        def guard_record(rec):
            rec.update({"worker": args.index, "test": current[0], "seed": current[1]})
            log.write(json.dumps(rec) + "\n")
            log.flush()
            print(f"[worker {args.index}] ORACLE MEMORY CHANGED during {rec['oracle_memory_changed_during']} ({current[0]} seed {current[1]}): {rec}", flush=True)
        forget = install_oracle_guard(snn_amd, guard_record)
    # the trap (tests/guard_arena.py): every host buffer the library sees ends at an inaccessible page and is retired to an
    # inaccessible (or canary) state when its call returns; the oracle's arrays are read-only during library calls; a fault is
    # reported with thread, pc and backtrace into <out>/guard-<pid>.log and the access completes
    trap = None
    if args.trap and not args.plain:
        import guard_arena
        trap = guard_arena.install(snn_amd, args.out)
        trap_faults_seen = 0
    while time.time() < deadline and not os.path.exists(os.path.join(args.out, "stop")):
        if accept is None or accept(seed):
            for spec, fn in fns:
                if not seed_is_valid(spec, seed):
                    continue
                counts[spec][0] += 1
                current[0], current[1] = spec, seed
                forget()
                if trap is not None:
                    trap.context = f"{spec.split(':')[1]} seed {seed}"
                recent = (recent + [[spec, seed]])[-12:]
                try:
                    fn(snn_amd, seed)
                except BaseException as e:           # noqa: BLE001 -- a campaign records everything and goes on
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    if type(e).__name__ == "Skipped":
                        continue
                    counts[spec][1] += 1
                    log.write(json.dumps({"worker": args.index, "test": spec, "seed": seed, "error": type(e).__name__,
                                          "message": str(e)[:6000], "preceding": recent,
                                          "traceback": traceback.format_exc()[-3000:]}) + "\n")
                    log.flush()
                    print(f"[worker {args.index}] FAILURE {spec} seed {seed}: {str(e)[:300]}", flush=True)
                if trap is not None:
                    for rec in trap.check():
                        rec.update({"worker": args.index, "trap_report": True, "test": spec, "seed": seed})
                        log.write(json.dumps(rec) + "\n")
                        log.flush()
                        print(f"[worker {args.index}] TRAP REPORT {spec} seed {seed}: {rec}", flush=True)
                    if trap.faults() > trap_faults_seen:
                        log.write(json.dumps({"worker": args.index, "trap_faults": trap.faults() - trap_faults_seen, "test": spec, "seed": seed,
                                              "log": os.path.basename(trap.log_path)}) + "\n")
                        log.flush()
                        print(f"[worker {args.index}] TRAP FAULT during {spec} seed {seed}: see {trap.log_path}", flush=True)
                        trap_faults_seen = trap.faults()
                size = os.path.getsize(vlog) if os.path.exists(vlog) else 0
                if size > vseen:
                    with open(vlog) as f:
                        f.seek(vseen)
                        lines = f.read().splitlines()
                    vseen = size
                    log.write(json.dumps({"worker": args.index, "verify_reports": lines[:8], "test": spec, "seed": seed}) + "\n")
                    log.flush()
        seed += args.workers
        if time.time() - last_report > 60:
            last_report = time.time()
            print(f"[worker {args.index}] seed {seed} " + " ".join(f"{s.split(':')[1]}={c[0]}/{c[1]}" for s, c in counts.items()), flush=True)
            log.write(json.dumps({"worker": args.index, "progress": True, "counts": counts, "next_seed": seed}) + "\n")
            log.flush()
    if trap is not None:
        for rec in trap.check(everything=True):
            rec.update({"worker": args.index, "trap_report": True, "test": "(end of the worker)", "seed": seed})
            log.write(json.dumps(rec) + "\n")
        log.write(json.dumps({"worker": args.index, "trap_stats": trap.stats()}) + "\n")
    log.write(json.dumps({"worker": args.index, "done": True, "counts": counts, "next_seed": seed,
                          "armed": {k: os.environ.get(k) for k in ("SNN_AMD_VERIFY", "MALLOC_PERTURB_", "SNN_HOST_POISON", "SNN_CHECKPOINTS")}}) + "\n")
    log.close()


def streamer(args):
    """C2-shaped passes over a dense matrix of `side`^4 * 4 bytes until the campaign ends"""
    import snn_amd
    n = args.side * args.side
    net = snn_amd.DeviceNetwork(model=0)
    net.add_lattice(0, args.side, args.side)
    net.finalize()
    net.fill_graph_synthetic(7 + args.index, 0.5, 1.5)
    net.set_synapses(True, False)
    deadline = time.time() + args.minutes * 60
    runs = 0
    while time.time() < deadline and not os.path.exists(os.path.join(args.out, "stop")):
        net.run(args.streamer_steps)
        runs += 1
        if args.streamer_pause_ms:
            time.sleep(args.streamer_pause_ms / 1000.0)
    net.close()
    print(f"[streamer {args.index}] {runs} runs of {args.streamer_steps} steps over {n} neurons", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="main", choices=["main", "worker", "streamer"])
    ap.add_argument("--index", type=int, default=0)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--streamers", type=int, default=2)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--first-seed", type=int, default=100_000)
    ap.add_argument("--tests", default="test_gpu_randomized:test_random_network")
    ap.add_argument("--filter", default="")
    ap.add_argument("--side", type=int, default=160)                 # 160^2 neurons: a 2.6 GB matrix per streamer
    ap.add_argument("--streamer-steps", type=int, default=20)
    ap.add_argument("--streamer-pause-ms", type=float, default=0.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "campaign"))
    ap.add_argument("--plain", action="store_true", help="workers without the verify / malloc-perturb / host-poison arming")
    ap.add_argument("--trap", type=int, default=1, help="armed workers also run the guard arena (tests/guard_arena.py)")
    ap.add_argument("--lean", action="store_true",
                    help="the trap with as little else as possible, so that it sees many executions per minute: no device self-check, no "
                         "checkpoints at the call boundaries, no digest guard (the trap's read-only oracle memory covers what that one "
                         "watches for CPU writers); malloc perturbation and host poison stay")
    args = ap.parse_args()
    os.makedirs(os.path.join(args.out, "repro"), exist_ok=True)
    if args.role == "worker":
        return worker(args)
    if args.role == "streamer":
        return streamer(args)
    stop = os.path.join(args.out, "stop")
    if os.path.exists(stop):
        os.remove(stop)
    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    t0 = time.time()
    import checkpoint
    ras_before = checkpoint.ras_counters()
    # the workers run ARMED: comparisons at every run-call boundary with checkpoints (tests/checkpoint.py; the default), and unless
    # --plain: the device steps every run call twice and compares with itself (SNN_AMD_VERIFY), glibc fills every malloc'ed and
    # freed block with a byte pattern (MALLOC_PERTURB_), the binding pre-fills its output buffers (SNN_HOST_POISON)
    worker_env = dict(os.environ)
    if not args.plain:
        worker_env.setdefault("SNN_AMD_VERIFY", "0" if args.lean else "1")
        worker_env.setdefault("MALLOC_PERTURB_", "165")
        worker_env.setdefault("SNN_HOST_POISON", "1")
        if args.lean:
            worker_env.setdefault("SNN_CHECKPOINTS", "0")
            # (every mprotect of the arena interrupts the CPUs the process's threads ran on: a worker's oracle needs no team)
            worker_env["OMP_NUM_THREADS"] = "1"
            worker_env.setdefault("MKL_NUM_THREADS", "1")
    procs = [subprocess.Popen(base + ["--role", "streamer", "--index", str(i)]) for i in range(args.streamers)]
    procs += [subprocess.Popen(base + ["--role", "worker", "--index", str(i)], env=worker_env) for i in range(args.workers)]
    # a process still busy two and a half minutes after the deadline (starved by thirty others, or stuck) is stopped: the summary is
    # written from what the workers recorded until then
    rcs, stop_at = [], t0 + args.minutes * 60 + 150
    for p in procs:
        try:
            rcs.append(p.wait(timeout=max(1.0, stop_at - time.time())))
        except subprocess.TimeoutExpired:
            p.terminate()
            try:
                rcs.append(p.wait(timeout=20))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(p.wait())
    ras_after = checkpoint.ras_counters()
    total, failures, verify_reports, guard_reports, trap_reports, trap_faults, trap_stats = {}, [], [], [], [], [], []
    for i in range(args.workers):
        path = os.path.join(args.out, f"worker-{i}.jsonl")
        latest = {}
        for line in open(path) if os.path.exists(path) else ():
            rec = json.loads(line)
            if rec.get("done") or rec.get("progress"):
                latest = rec["counts"]            # (cumulative: the last record of a worker counts)
            elif "verify_reports" in rec:
                verify_reports.append(rec)
            elif "oracle_memory_changed_during" in rec:
                guard_reports.append(rec)
            elif rec.get("trap_report"):
                trap_reports.append(rec)
            elif "trap_faults" in rec:
                trap_faults.append(rec)
            elif "trap_stats" in rec:
                trap_stats.append(rec["trap_stats"])
            else:
                failures.append({k: rec[k] for k in ("worker", "test", "seed", "error", "message", "preceding")})
        for spec, (n, f) in latest.items():
            t = total.setdefault(spec, [0, 0])
            t[0] += n
            t[1] += f
    summary = {"minutes": args.minutes, "wall_s": round(time.time() - t0, 1), "workers": args.workers, "streamers": args.streamers,
               "streamer_side": args.side, "lean": args.lean, "tests": args.tests, "filter": args.filter, "first_seed": args.first_seed,
               "executions_and_failures": total, "executions": sum(t[0] for t in total.values()),
               "failures": sum(t[1] for t in total.values()), "failure_records": failures, "exit_codes": rcs,
               "oracle_memory_reports": guard_reports[:50], "trap_armed": bool(args.trap and not args.plain),
               "trap_faults": sum(r["trap_faults"] for r in trap_faults), "trap_fault_records": trap_faults[:50], "trap_reports": trap_reports[:50],
               "trap_calls": sum(t["calls"] for t in trap_stats), "trap_buffers_retired": sum(t["buffers_retired"] for t in trap_stats),
               "trap_library_host_tables": sum(t.get("library_host_tables", 0) for t in trap_stats),
               "self_check_reports": len(verify_reports), "self_check_records": verify_reports[:200],
               "environment": {k: v for k, v in worker_env.items() if k.startswith(("SNN_", "AMD_", "HIP_", "HSA_", "OMP_", "MALLOC_", "GPU_"))},
               "ras_errors_before_ue_ce": checkpoint.ras_totals(ras_before), "ras_errors_after_ue_ce": checkpoint.ras_totals(ras_after),
               "ras_before": ras_before, "ras_after": ras_after}
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: summary[k] for k in ("wall_s", "executions", "failures", "trap_faults", "trap_calls", "executions_and_failures", "exit_codes")}))


if __name__ == "__main__":
    main()

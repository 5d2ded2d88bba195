#!/usr/bin/env python3
"""First hour on a multi-GPU node (BASELINE configs[4]): nothing of the N > 1 path has crossed xGMI yet, so this one
script runs, in order, what has to be known first and leaves ONE table behind.

    python3 profiles/first_contact.py --out gpurun_out/first_contact            # a node with 2 ... 8 GPUs
    python3 profiles/first_contact.py --emulate --gpus-list 1,2 --rows 48       # rehearsal on a one-GPU box (nothing measured)

 1. tests/test_gpu_multi_device.py (2 / 4 / 8 real ranks over RCCL against the oracle; skipped below 2 visible GPUs).
 2. bench.py --gpus N for N in --gpus-list:
      c2 strong                 the headline lattice, post-population shards, ncclAllGather of voltages + spike bits
      c5 strong, collective     4 x 512 x 512 + Poisson cells, shards by lattice, grouped ncclSend / ncclRecv of the halo
      c5 strong, peer form      the same with --peer-form (tried; falls back by agreement)
      c5 weak,   collective     1 M neurons per rank
      c5 weak,   peer form
 3. One row per run: value, ms per step, efficiency against N = 1 of the same series, rccl_ranks, peer_form, halo_peer_steps, the
    per-rank step / compute-only / exchange times, exchange bytes, state_sha256.
 4. Exit code 1 when any run failed or any checksum differs where it must not: strong series -- every N equals N = 1; at one N
    the collective and the peer form of the same network equal each other.

--bench replaces `python3 bench.py` (tests/test_host_logic.py runs the script against a stub)."""
import argparse
import json
import os
import shlex
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def series(args):
    common = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline"]
    # bench.py --peer-form runs warm-up, an 8-step trial and a second warm-up before the timed region: the collective series
    # of c5 gets the same number of steps as its warm-up, so that the two forms can be held against each other checksum by checksum
    common5 = ["--steps", str(args.steps), "--warmup", str(2 * args.warmup + 8), "--no-cpu-baseline"]
    small = (["--rows", str(args.rows), "--cols", str(args.rows)] if args.rows else [])
    out = [("c2 strong", ["--config", "c2"] + small + common, "strong"),
           ("c5 strong collective", ["--config", "c5"] + (["--rows", str(args.rows)] if args.rows else []) + common5, "strong"),
           ("c5 strong peer form", ["--config", "c5", "--peer-form"] + (["--rows", str(args.rows)] if args.rows else []) + common, "strong"),
           ("c5 weak collective", ["--config", "c5", "--scaling", "weak"] + (["--rows", str(args.rows)] if args.rows else []) + common5, "weak"),
           ("c5 weak peer form", ["--config", "c5", "--scaling", "weak", "--peer-form"] + (["--rows", str(args.rows)] if args.rows else []) + common, "weak")]
    return [s for s in out if not args.only or any(s[0].startswith(o) for o in args.only.split(","))]


def run_bench(args, flags, n):
    cmd = shlex.split(args.bench) + ["--gpus", str(n)] + flags + (["--emulate-ranks-on-one-gpu"] if args.emulate and n > 1 else [])
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=args.timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    rec = {"command": " ".join(cmd), "exit": p.returncode, "wall_s": round(time.time() - t0, 1)}
    if p.returncode != 0 or not lines:
        rec["error"] = (p.stderr or p.stdout)[-1500:]
        return rec
    try:
        rec["line"] = json.loads(lines[-1])
    except ValueError as e:
        rec["error"] = f"unparsable line: {e}"
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "first_contact"))
    ap.add_argument("--gpus-list", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=0, help="smaller lattices (rehearsals); 0 = the BASELINE sizes")
    ap.add_argument("--only", default="", help="comma-separated prefixes of the series to run, e.g. 'c5 strong'")
    ap.add_argument("--emulate", action="store_true", help="one-GPU rehearsal: bench.py --emulate-ranks-on-one-gpu for N > 1")
    ap.add_argument("--skip-tests", action="store_true")
    ap.add_argument("--bench", default=f"{shlex.quote(sys.executable)} {shlex.quote(os.path.join(ROOT, 'bench.py'))}")
    ap.add_argument("--timeout", type=int, default=1800)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    ns = [int(x) for x in args.gpus_list.split(",")]
    report = {"started": time.strftime("%Y-%m-%d %H:%M:%S"), "emulated": args.emulate, "gpus_list": ns, "rows": [], "problems": []}

    if not args.skip_tests:
        p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_multi_device.py"), "-m", "gpu", "-q", "-x"],
                           capture_output=True, text=True, cwd=ROOT)
        tail = (p.stdout.strip().splitlines() or [""])[-1]
        report["multi_device_tests"] = {"exit": p.returncode, "summary": tail}
        if p.returncode not in (0, 5):
            report["problems"].append(f"tests/test_gpu_multi_device.py failed: {tail}")
        with open(os.path.join(args.out, "multi_device_tests.log"), "w") as f:
            f.write(p.stdout + p.stderr)

    sha = {}                                             # (series, N) -> checksum
    for name, flags, scaling in series(args):
        base = None
        for n in ns:
            rec = run_bench(args, flags, n)
            row = {"series": name, "n_gpus": n, "exit": rec["exit"], "wall_s": rec["wall_s"], "command": rec["command"]}
            if "line" not in rec:
                row["error"] = rec.get("error", "no line")
                report["problems"].append(f"{name}, N = {n}: no bench line (exit {rec['exit']})")
                report["rows"].append(row)
                continue
            ln = rec["line"]
            if n == ns[0]:
                base = ln["value"] / ns[0]
            rt = ln.get("rank_times") or {}
            row.update({"value": ln["value"], "ms_per_step": ln["ms_per_step"],
                        "efficiency_vs_first_n": (ln["value"] / (base * n)) if base else None,
                        "rccl_ranks": ln.get("rccl_ranks"), "transport": ln.get("transport"), "stepper": ln.get("stepper"),
                        "peer_form": ln.get("peer_form"), "halo_peer_steps": ln.get("halo_peer_steps"),
                        "exchange_bytes_per_rank_step": ln.get("exchange_bytes_per_rank_step"),
                        "step_ms_by_rank": rt.get("step_ms_by_rank"), "compute_only_ms_by_rank": rt.get("compute_only_ms_by_rank"),
                        "exchange_ms_by_rank": rt.get("exchange_ms_by_rank"),
                        "state_sha256": ln.get("state_sha256"), "state_after_steps": ln.get("state_after_steps"),
                        "roofline_frac": (ln.get("roofline") or {}).get("frac")})
            sha[(name, n)] = (ln.get("state_sha256"), ln.get("state_after_steps"))
            if n > 1 and not args.emulate and ln.get("rccl_ranks") != n:
                report["problems"].append(f"{name}, N = {n}: rccl_ranks is {ln.get('rccl_ranks')} (the library's communicator does not span the ranks)")
            if "peer form" in name and n > 1 and ln.get("peer_form") != "taken":
                report["problems"].append(f"{name}, N = {n}: the peer form was not taken: {ln.get('peer_form')}")
            report["rows"].append(row)
        if scaling == "strong":
            # the same network for every N: the state after the same number of steps must not depend on N
            first = sha.get((name, ns[0]))
            for n in ns[1:]:
                got = sha.get((name, n))
                if first and got and got[1] == first[1] and got[0] != first[0]:
                    report["problems"].append(f"{name}: the checksum at N = {n} differs from N = {ns[0]}")
    # the peer form and the collective of the same network at the same N, after the same number of steps
    for kind in ("strong", "weak"):
        for n in ns:
            a, b = sha.get((f"c5 {kind} collective", n)), sha.get((f"c5 {kind} peer form", n))
            if a and b and a[1] == b[1] and a[0] != b[0]:
                report["problems"].append(f"c5 {kind}, N = {n}: the peer form's checksum differs from the collective's")
    report["ok"] = not report["problems"]
    with open(os.path.join(args.out, "first_contact.json"), "w") as f:
        json.dump(report, f, indent=1)
    cols = ["series", "n_gpus", "value", "ms_per_step", "efficiency_vs_first_n", "rccl_ranks", "peer_form", "halo_peer_steps", "state_sha256"]
    with open(os.path.join(args.out, "first_contact.md"), "w") as f:
        f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
        for r in report["rows"]:
            cells = []
            for c in cols:
                v = r.get(c)
                cells.append("" if v is None else (f"{v:.4g}" if isinstance(v, float) else str(v)[:16] if c == "state_sha256" else str(v)[:60]))
            f.write("| " + " | ".join(cells) + " |\n")
        f.write("\n" + ("no problems\n" if report["ok"] else "PROBLEMS:\n" + "".join(f"* {p}\n" for p in report["problems"])))
    print(open(os.path.join(args.out, "first_contact.md")).read())
    return 0 if report["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())

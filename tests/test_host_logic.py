"""Host-side logic that needs no GPU: shard geometry, exchange-buffer layout, synthetic data."""
import os
import sys

import numpy as np

import snn_amd
from snn_amd import parallel


def test_shard_geometry_covers_population_in_wavefront_slots():
    for n, g in [(65536, 8), (65536, 1), (1000, 4), (81920, 8), (63, 2), (0, 2), (130, 3)]:
        stride, shards = parallel.shard_geometry(n, g)
        assert stride % 64 == 0 and stride * g >= n
        assert shards[0][0] == 0 and shards[-1][1] == n
        covered = 0
        for r, (b, e) in enumerate(shards):
            assert b == min(n, r * stride) and e - b <= stride and b <= e
            covered += e - b
        assert covered == n


def test_c2_geometry_is_exact_at_8_gpus():
    stride, shards = parallel.shard_geometry(256 * 256, 8)
    assert stride == 8192 and shards[7] == (57344, 65536)


def test_synthetic_uniform_is_counter_based():
    a = snn_amd.synthetic.uniform(2, 64, 0.5, 1.5, offset=1000)
    b = snn_amd.synthetic.uniform(2, 1064, 0.5, 1.5)[1000:]
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert 0.5 <= a.min() and a.max() < 1.5


def _bench_under_stub(tmp_path, argv, launcher_rc=0):
    """runs `python bench.py <argv>` with a stub in place of `python -m torch.distributed.run` and a `torch` package that
    refuses to be imported: returns (CompletedProcess, what the stub was started with)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    record = tmp_path / "launched.json"
    stub = tmp_path / "launcher.py"
    stub.write_text("import json, os, sys\n"
                    f"json.dump({{'argv': sys.argv[1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}}, open({str(record)!r}, 'w'))\n"
                    "print('RCCL version banner')\n"
                    "print(json.dumps({'metric': 'neuron-steps/sec', 'n_gpus': 2, 'rccl_ranks': 2}))\n"
                    f"sys.exit({launcher_rc})\n")
    fake = tmp_path / "fake" / "torch"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text("raise ImportError('the launching process must not import torch')\n")
    env = dict(os.environ, SNN_BENCH_LAUNCHER=f"{sys.executable} {stub}", PYTHONPATH=str(tmp_path / "fake"))
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=120)
    return proc, (json.load(open(record)) if record.exists() else None)


def test_bench_starts_its_own_ranks_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 2` without a launcher's environment: a fresh child `torch.distributed.run --nproc-per-node 2
    bench.py <same arguments>` is started before torch is imported (the stub torch would raise), its stdout is relayed
    with the JSON line last, and its exit code is the bench's"""
    import json
    proc, launched = _bench_under_stub(tmp_path, ["--gpus", "2", "--steps", "7", "--config", "c5", "--scaling", "weak"])
    assert proc.returncode == 0, proc.stderr
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    a = launched["argv"]
    assert "--nnodes=1" in a and "--nproc-per-node=2" in a and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert int(a[a.index("--master-port") + 1]) > 0
    script = [x for x in a if x.endswith("bench.py")]
    assert script and a[a.index(script[0]) + 1:] == ["--gpus", "2", "--steps", "7", "--config", "c5", "--scaling", "weak"]
    assert launched["ipc"] == "0"                      # dmabuf IPC: the only mode RCCL across processes works in here


def test_bench_relays_a_failing_launch(tmp_path):
    proc, _ = _bench_under_stub(tmp_path, ["--gpus", "4"], launcher_rc=3)
    assert proc.returncode == 3 and "exited with code 3" in proc.stderr


def test_bench_single_gpu_stays_in_process(tmp_path):
    """--gpus 1 (the default) launches nothing: it goes straight on to import torch in this process"""
    proc, launched = _bench_under_stub(tmp_path, ["--steps", "3"])
    assert launched is None and proc.returncode != 0 and "must not import torch" in proc.stderr


def test_c5_structure_on_rectangular_lattices():
    """the weak-scaling form of bench.py --config c5: four (rows x cols) lattices; checked against a plain double loop"""
    from snn_amd import synthetic
    rows, cols = 6, 4
    m, nn = rows * cols, 4 * rows * cols
    ptr, pre, w = synthetic.c5_csr(rows, cols=cols)
    assert ptr[-1] == pre.size == w.size and np.all(w == 1.0)
    for q in range(nn):
        k, rem = divmod(q, m)
        r, c = divmod(rem, cols)
        want = {k * m + rr * cols + cc for rr in range(rows) for cc in range(cols)
                if 0 < (rr - r) ** 2 + (cc - c) ** 2 <= 4}
        want |= {((k - 1) % 4) * m + rem, nn + k * m + rem}
        got = pre[int(ptr[q]):int(ptr[q + 1])]
        assert list(got) == sorted(want)
    sq = synthetic.c5_csr(5)
    sq2 = synthetic.c5_csr(5, cols=5)
    assert all(np.array_equal(a, b) for a, b in zip(sq, sq2))


def _first_contact(tmp_path, behaviour):
    """profiles/first_contact.py against a stub bench: `behaviour` is Python source that may change `line` or exit"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "stub_bench.py"
    stub.write_text(
        "import argparse, hashlib, json, sys\n"
        "ap = argparse.ArgumentParser()\n"
        "for f in ('--gpus', '--steps', '--warmup', '--rows', '--cols'): ap.add_argument(f, type=int, default=1)\n"
        "ap.add_argument('--config'); ap.add_argument('--scaling', default='strong')\n"
        "for f in ('--peer-form', '--no-cpu-baseline', '--emulate-ranks-on-one-gpu'): ap.add_argument(f, action='store_true')\n"
        "a = ap.parse_args()\n"
        "extra = 8 + a.warmup if a.peer_form else 0\n"
        "steps = a.warmup + extra + 5 * a.steps\n"
        "net = (a.config, a.gpus if a.scaling == 'weak' else 1, steps)\n"
        "line = {'value': 1e6 * a.gpus * 0.9, 'ms_per_step': 1.0 / a.gpus, 'n_gpus': a.gpus, 'rccl_ranks': a.gpus if a.gpus > 1 else None,\n"
        "        'peer_form': 'taken' if a.peer_form and a.gpus > 1 else None, 'halo_peer_steps': 99 if a.peer_form and a.gpus > 1 else 0,\n"
        "        'state_sha256': hashlib.sha256(repr(net).encode()).hexdigest(), 'state_after_steps': steps,\n"
        "        'rank_times': {'step_ms_by_rank': [1.0] * a.gpus, 'compute_only_ms_by_rank': None, 'exchange_ms_by_rank': None},\n"
        "        'roofline': {'frac': 0.5}}\n"
        + behaviour +
        "\nprint('noise before the line')\nprint(json.dumps(line))\n")
    out = tmp_path / "out"
    p = subprocess.run([sys.executable, os.path.join(root, "profiles", "first_contact.py"), "--skip-tests", "--gpus-list", "1,2,4", "--steps", "10",
                        "--warmup", "3", "--out", str(out), "--bench", f"{sys.executable} {stub}"], capture_output=True, text=True, timeout=300)
    return p, json.load(open(out / "first_contact.json"))


def test_first_contact_script_collects_one_table(tmp_path):
    p, rep = _first_contact(tmp_path, "")
    assert p.returncode == 0, p.stdout + p.stderr
    assert rep["ok"] and len(rep["rows"]) == 5 * 3 and not rep["problems"]
    by = {(r["series"], r["n_gpus"]): r for r in rep["rows"]}
    assert by[("c2 strong", 4)]["efficiency_vs_first_n"] == 1.0 and by[("c5 strong peer form", 2)]["peer_form"] == "taken"
    # the collective series of c5 is given the step count of the peer-form series: their checksums are comparable (and equal here)
    assert by[("c5 strong collective", 2)]["state_after_steps"] == by[("c5 strong peer form", 2)]["state_after_steps"]
    assert by[("c5 strong collective", 2)]["state_sha256"] == by[("c5 strong peer form", 2)]["state_sha256"]
    assert "| c5 weak peer form | 4 |" in p.stdout and "no problems" in p.stdout


def test_first_contact_script_fails_on_a_checksum_difference(tmp_path):
    p, rep = _first_contact(tmp_path, "if a.config == 'c5' and a.gpus == 2 and not a.peer_form and a.scaling == 'strong': line['state_sha256'] = 'f' * 64")
    assert p.returncode == 1 and not rep["ok"]
    text = " ".join(rep["problems"])
    assert "c5 strong collective: the checksum at N = 2 differs from N = 1" in text
    assert "c5 strong, N = 2: the peer form's checksum differs from the collective's" in text


def test_first_contact_script_fails_on_a_failed_run_or_a_missing_communicator(tmp_path):
    p, rep = _first_contact(tmp_path, "if a.config == 'c2' and a.gpus == 4: sys.exit(3)\nif a.config == 'c5' and a.gpus == 2: line['rccl_ranks'] = 1\n"
                                      "if a.peer_form and a.gpus == 4: line['peer_form'] = 'fell back to the collective: hipIpcOpenMemHandle'")
    assert p.returncode == 1
    text = " ".join(rep["problems"])
    assert "c2 strong, N = 4: no bench line (exit 3)" in text and "rccl_ranks is 1" in text and "the peer form was not taken" in text


def test_scratch_and_the_lab_build_do_not_ship_to_the_gpu_box():
    """.gpurunignore: the snapshot matches the bare directory name -- `scratch/` alone shipped 145 MiB with every lease of round 6"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = open(os.path.join(root, ".gpurunignore")).read().split()
    assert "scratch" in lines and "spiking-neural-networks_amd/csrc/lab" in lines
    # what the GPU tests load must NOT be listed: the built libraries travel with the snapshot
    assert not any(l.endswith(".so") or l.rstrip("/").endswith("csrc") or "generated" in l for l in lines)

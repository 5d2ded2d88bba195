"""N > 1 on REAL hardware: these tests run only where at least two GPUs are visible (they skip on the one-GPU test boxes)
and are what first contact with a multi-GPU node should be.  One process per GPU (fresh children started through
torch.distributed.run), the library's own step loop snn_run_sharded with RCCL all-gather (dense) or grouped send / recv of
halo segments (sparse); the union of the ranks' own neurons, the part of the others' state each rank holds, and every
weight after STDP must equal the single-process oracle bit for bit -- for every world size the node offers.  Also:
bench.py's state checksum must not depend on --gpus."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 200


def visible_gpus():
    import torch
    return torch.cuda.device_count()           # does not initialise the GPU: the children are started first


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, script_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("form", ["dense", "csr", "csr_by_lattice"])
def test_run_sharded_over_rccl_equals_the_oracle(form, world):
    if visible_gpus() < world:
        pytest.skip(f"needs {world} GPUs, {visible_gpus()} visible")
    import multi_gpu_worker
    import parity
    with tempfile.TemporaryDirectory() as d:
        r = launch(world, [os.path.join(ROOT, "tests", "multi_gpu_worker.py"), "--form", form, "--steps", str(STEPS), "--out", d])
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        ref = multi_gpu_worker.build(form)
        ref.run(STEPS, spike_history=True)
        assert ref.spike_history.sum() > 20
        covered = np.zeros(ref.n_neurons, bool)
        devices = set()
        for rank in range(world):
            z = np.load(os.path.join(d, f"rank{rank}.npz"))
            own, k = z["own"], z["known"]
            devices.add(int(z["device"]))
            assert str(z["mode"]) == ("halo" if form.startswith("csr") else "allgather") and int(z["clock"]) == STEPS
            assert not (covered & own).any()
            covered |= own
            for name in ("current_voltage", "is_spiking", "last_firing_time"):
                assert np.array_equal(parity.bits(z[name][k]), parity.bits(ref[name][k])), (rank, name)
            assert np.array_equal(parity.bits(z["nt_t"][own]), parity.bits(ref["nt_t"][own])), rank
            assert np.array_equal(parity.bits(z["w_value"][own]), parity.bits(ref["w_value"][own])), rank
            if form.startswith("csr"):
                _, _, want = parity.csr_for_posts(ref, np.flatnonzero(own))
                assert np.array_equal(parity.bits(z["weights"]), parity.bits(want)), rank
            else:
                ow = np.where(ref["connections"] != 0, ref["weights"], np.float32(0))
                assert np.array_equal(parity.bits(z["weights"][:, own]), parity.bits(ow[:, own])), rank
        assert covered.all() and len(devices) == world


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("config,extra", [("c2", ["--rows", "96", "--cols", "96"]), ("c5", ["--rows", "64"])])
def test_bench_state_checksum_does_not_depend_on_the_gpu_count(config, extra):
    """bench.py prints state_sha256 (voltages, firing times, spike flags, spike totals of every neuron after the timed
    steps): the N = 1 line and every N > 1 line of the same command must carry the same value"""
    worlds = [w for w in (1, 2, 4, 8) if w <= visible_gpus()]
    if len(worlds) < 2:
        pytest.skip(f"needs at least 2 GPUs, {visible_gpus()} visible")
    sums = {}
    for world in worlds:
        args = [os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", str(world), "--steps", "30", "--warmup", "5",
                "--repeats", "2", "--no-cpu-baseline", "--spike-fraction", "0.002"] + extra
        r = launch(world, args)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == world and line["state_after_steps"] == 5 + 30 * 2
        sums[world] = line["state_sha256"]
        assert line["spikes_per_step"] > 0
    assert len(set(sums.values())) == 1, sums


@pytest.mark.timeout(900)
@pytest.mark.parametrize("config,extra", [("c2", ["--rows", "96", "--cols", "96"]), ("c5", ["--rows", "64"]), ("c4", [])])
def test_bench_state_checksum_of_the_sharded_path_at_world_size_one(config, extra):
    """what ONE GPU can say about it: the multi-GPU code path (shard handle, pack, RCCL self-exchange, unpack) leaves the
    same state as snn_run"""
    if config == "c4" and visible_gpus() < 1:
        pytest.skip("no GPU")
    sums = {}
    for forced in (False, True):
        args = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "20", "--warmup", "3", "--repeats", "2",
                "--no-cpu-baseline", "--spike-fraction", "0.002"] + extra + (["--force-sharded"] if forced else [])
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        r = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        sums[forced] = line["state_sha256"]
        assert line["spikes_per_step"] > 0 and line["state_after_steps"] == 3 + 20 * 2
    assert sums[False] == sums[True], sums

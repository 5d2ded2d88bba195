"""`bench.py --gpus 2` end to end on a box with ONE GPU (--emulate-ranks-on-one-gpu: a process per rank on device 0, gloo, the
library's collectives through host memory): the launcher, the agreement before a run, snn_run_sharded, the peer-form trial
(`--peer-form`) and BOTH its outcomes -- taken, and a trial that one rank sabotages so that every rank must return to the
collective on a rebuilt handle -- end with the state checksum of the one-rank run after the same number of steps.
(RCCL's own kernels and xGMI are what this cannot show; tests/test_gpu_multi_device.py waits for a multi-GPU node.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, K = 6, 10


def bench(*flags, gpus=2, warmup=W):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", "c5", "--rows", "48", "--steps", str(K), "--warmup", str(warmup),
           "--repeats", "2", "--no-cpu-baseline", "--no-kernel-events", *flags]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(1800)
def test_two_ranks_on_one_gpu_collective_peer_form_and_fallback(snn):
    one = bench(gpus=1)
    one_long = bench("--peer-form", gpus=1)                 # the step count of a run that went through a peer-form trial
    coll = bench("--emulate-ranks-on-one-gpu")
    assert coll["n_gpus"] == 2 and coll["transport"].startswith("EMULATION") and coll["rccl_ranks"] is None
    assert coll["peer_form"] is None and coll["halo_peer_steps"] == 0
    assert coll["stepper"].startswith("library") and coll["exchange_bytes_per_rank_step"]["mode"] == "halo"
    assert coll["state_after_steps"] == one["state_after_steps"] and coll["state_sha256"] == one["state_sha256"]
    peer = bench("--emulate-ranks-on-one-gpu", "--peer-form")
    assert peer["peer_form"] == "taken" and peer["halo_peer_steps"] >= 2 * K + W
    assert peer["state_after_steps"] == one_long["state_after_steps"] and peer["state_sha256"] == one_long["state_sha256"]
    back = bench("--emulate-ranks-on-one-gpu", "--peer-form", "--sabotage-peer-trial", "1")
    assert back["peer_form"].startswith("fell back to the collective") and back["halo_peer_steps"] == 0
    assert back["state_after_steps"] == one_long["state_after_steps"] and back["state_sha256"] == one_long["state_sha256"]

"""Runs the tests that emulate the ranks of a multi-GPU run as threads of ONE process (marker `emulated_ranks`: the peer form, the
library's own sharded loop with replaced collectives) in a child pytest process with GPU_MAX_HW_QUEUES=24 -- every rank's stream
needs a hardware queue of its own there, or a kernel that polls a neighbour's values can sit in front of the kernel that produces
them.  Everything else in the suite keeps the HIP runtime's default stream-to-queue mapping (4 queues), which is what a
one-rank-per-process production run uses; round 4 had set the variable for the whole suite."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(3000)
def test_the_emulated_rank_tests_pass_in_their_own_process():
    if os.environ.get("SNN_EMULATED_RANKS_CHILD") == "1":
        pytest.skip("this IS the child process")
    env = dict(os.environ, SNN_EMULATED_RANKS_CHILD="1", GPU_MAX_HW_QUEUES="24")
    r = subprocess.run([sys.executable, "-m", "pytest", HERE, "-m", "gpu and emulated_ranks", "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=os.path.dirname(HERE), timeout=2900)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-25:])
    print("\n[emulated ranks, child process] " + tail.splitlines()[-1] if tail else "no output")
    assert r.returncode == 0, f"child pytest failed (exit {r.returncode}):\n{tail}"
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 20, tail

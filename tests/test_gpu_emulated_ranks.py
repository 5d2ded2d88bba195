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
# Round 6: with the ranks emulated on ONE device, a peer-form step now and then ended with "did not arrive
# within the spin limit" (error 6, the handle mid-step -- loud, never a wrong result): 3 of 16 child runs
# (profiles/r06/s7_peer_form_repeats.txt).  Cause: device-wide synchronisations (hipFree at the end of the agreement,
# torch.cuda.synchronize in the thread collectives) by a rank whose neighbour had already launched a step that polls for it;
# removed, 0 of 20 runs since (s9_peer_form_repeats.txt).  Kept as a net: a child run that fails THAT way is repeated, the
# give-up is printed and appended to gpurun_out/emulated_ranks_give_ups.txt; any other failure, or a third give-up, fails.
GIVE_UP = "did not arrive within the spin limit"
GIVE_UP_RETRIES = 2
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(3000)
def test_the_emulated_rank_tests_pass_in_their_own_process():
    if os.environ.get("SNN_EMULATED_RANKS_CHILD") == "1":
        pytest.skip("this IS the child process")
    env = dict(os.environ, SNN_EMULATED_RANKS_CHILD="1", GPU_MAX_HW_QUEUES="24")
    gave_up = []
    for attempt in range(1 + GIVE_UP_RETRIES):
        r = subprocess.run([sys.executable, "-m", "pytest", HERE, "-m", "gpu and emulated_ranks", "-q", "-x", "-p", "no:cacheprovider"],
                           capture_output=True, text=True, env=env, cwd=os.path.dirname(HERE), timeout=2900)
        tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-25:])
        if r.returncode == 0 or GIVE_UP not in r.stdout + r.stderr:
            break
        failed = re.findall(r"^FAILED (\S+)", r.stdout, flags=re.M)
        gave_up.append(failed[0] if failed else "?")
    if gave_up:
        # never silent: the line is in the test's output and in gpurun_out/ (which travels back from the GPU box)
        note = f"[emulated ranks] the peer form GAVE UP in {len(gave_up)} child run(s) before one passed: {gave_up}"
        print("\n" + note)
        out = os.path.join(os.path.dirname(HERE), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "emulated_ranks_give_ups.txt"), "a") as f:
            f.write(note + "\n")
    print("\n[emulated ranks, child process] " + tail.splitlines()[-1] if tail else "no output")
    assert r.returncode == 0, f"child pytest failed (exit {r.returncode}; give-ups before: {gave_up}):\n{tail}"
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 20, tail

"""The trap (tests/guard_arena.py) armed around the REAL device library: seeded differential tests run through it with every host
buffer of every C-ABI call in the arena (ending at an inaccessible page, retired to inaccessible / canary when the call returns)
and the oracle's arrays read-only during the calls.  A clean library reports nothing -- with the runtime's own pageable staging
("pinned_copies" 0) and with the handle's page-locked buffer (1, 2) -- and still agrees with the oracle bit for bit."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.dirname(sys.argv[1]))
    import conftest                                   # OpenMP settings of the oracle
    import snn_amd
    snn_amd._lib.load()
    import guard_arena
    guard = guard_arena.install(snn_amd, sys.argv[2])
    import test_gpu_randomized, test_gpu_sequences, test_gpu_persistent_run, test_gpu_reward_network
    ran = 0
    for seed in range(int(sys.argv[3]), int(sys.argv[3]) + int(sys.argv[4])):
        for name, fn, ok in (("test_random_network", test_gpu_randomized.test_random_network, not test_gpu_randomized.threaded(seed)),
                             ("test_random_call_sequence", test_gpu_sequences.test_random_call_sequence, test_gpu_sequences.usable(seed)),
                             ("test_random_electrical_networks", test_gpu_persistent_run.test_random_electrical_networks, True),
                             ("test_connections_between_lattices", test_gpu_reward_network.test_connections_between_lattices, True)):
            if not ok:
                continue
            guard.context = f"{name} seed {seed}"
            try:
                fn(snn_amd, seed)
                ran += 1
            except BaseException as e:
                if type(e).__name__ != "Skipped":
                    raise
    reports = guard.check(everything=True)
    print(json.dumps({"ran": ran, "reports": reports, "stats": guard.stats()}))
""")


@pytest.mark.parametrize("pinned", [0, 1, 2])
def test_seeded_tests_through_the_trap_report_nothing(tmp_path, pinned):
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, SNN_AMD_PINNED_COPIES=str(pinned), SNN_HOST_POISON="1")
    p = subprocess.run([sys.executable, str(script), HERE, str(tmp_path / "log"), str(7000 + 100 * pinned), "10"], capture_output=True,
                       text=True, timeout=800, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    res = json.loads(p.stdout.strip().splitlines()[-1])
    log = open(res["stats"]["log"]).read() if os.path.exists(res["stats"]["log"]) else ""
    assert res["stats"]["faults"] == 0 and res["reports"] == [], (res, log[-6000:])
    assert res["ran"] >= 20 and res["stats"]["calls"] > 500 and res["stats"]["buffers_retired"] > 200, res
    # the library's own host tables (the temporaries of its getters among them) came from the arena too
    assert res["stats"]["library_host_tables"] > 500, res

import os
import sys

# The oracle's OpenMP runtime (libgomp, shared with torch) sees every CPU of the box (256 on the GPU boxes) while the container may
# run 16 of them: with its default active waiting, teams sized for the visible CPUs spin against the quota and single calls take
# tens of milliseconds once the process holds a few more threads (the full GPU suite went from 5 to 19 minutes, one oracle call in
# a loop sat for 15).  Before anything loads libgomp: sleep at barriers, and size default teams by what the container may use.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
if "OMP_NUM_THREADS" not in os.environ:
    _cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        _quota, _period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if _quota != "max":
            _cpus = min(_cpus, max(1, int(int(_quota) / int(_period))))
    except (OSError, ValueError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(_cpus, 64)))

# Several shard handles of ONE process stand in for the ranks of a multi-GPU run (tests of the sharded loops): each rank's
# stream must own a hardware queue, or a kernel that waits for a neighbour's values can sit in front of the very kernel that
# produces them (the runtime spreads streams over 4 queues by default).  One rank per process -- the real thing -- never shares.
# Only the tests that do this (marker `emulated_ranks`) run with more queues, in a child pytest process of their own
# (tests/test_gpu_emulated_ranks.py): every other test keeps the runtime's default stream-to-queue mapping, the one production uses.
EMULATED_RANKS_CHILD = os.environ.get("SNN_EMULATED_RANKS_CHILD") == "1"
if EMULATED_RANKS_CHILD:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emulated_ranks: several shard handles of ONE process stepped concurrently as if they were ranks "
                                       "(needs GPU_MAX_HW_QUEUES=24: runs in the child process of tests/test_gpu_emulated_ranks.py)")
    # the in-tree native pieces are git-ignored build products: (re)build them when missing or stale
    import subprocess
    import snn_amd
    snn_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def pytest_collection_modifyitems(config, items):
    # no test may sit on a GPU box for ever: with pytest-timeout present every test without its own limit gets 15 minutes
    # (the slowest one, the full-size C4 check, takes 20 s) and a hang ends in a stack dump of all threads
    if not EMULATED_RANKS_CHILD:
        skip = pytest.mark.skip(reason="ranks emulated by threads: runs in the child process of tests/test_gpu_emulated_ranks.py (GPU_MAX_HW_QUEUES=24)")
        for item in items:
            if item.get_closest_marker("emulated_ranks") is not None:
                item.add_marker(skip)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def snn():
    """The product package with its HIP library loaded (fails loudly when the .so is missing)."""
    import snn_amd
    snn_amd._lib.load()
    return snn_amd

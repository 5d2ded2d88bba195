import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree native pieces are git-ignored build products: (re)build them when missing or stale
    import subprocess
    import snn_amd
    snn_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def pytest_collection_modifyitems(config, items):
    # no test may sit on a GPU box for ever: with pytest-timeout present every test without its own limit gets 15 minutes
    # (the slowest one, the full-size C4 check, takes 20 s) and a hang ends in a stack dump of all threads
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

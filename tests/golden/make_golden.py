#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz:  python tests/golden/make_golden.py [--check] [case ...]

The vectors are written by the numpy restatement tests/numpy_net.py (host libm for exp / powf), NOT by the C oracle:
the oracle (tests/test_golden_oracle.py) and the HIP stepper (tests/test_gpu_golden.py) are both held to them, so a
committed vector pins two independently written implementations.  The case builders in
tests/golden_cases.py only lay out INPUT arrays (in an oracle-binding container; nothing of the oracle is stepped).
--check compares with the committed files instead of writing them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_cases  # noqa: E402
import numpy_net  # noqa: E402


def generate(name):
    src, steps = golden_cases.CASES[name]()
    net = numpy_net.NumpyNet(src).run(steps, st_voltage_history=bool(src.n_cells))
    return golden_cases.outputs(name, net, steps), net


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    bad = 0
    for name in (args or golden_cases.CASES):
        out, net = generate(name)
        path = os.path.join(HERE, name + ".npz")
        if check:
            want = np.load(path)
            diff = [k for k in want.files if np.asarray(out[k]).tobytes() != want[k].tobytes()]
            bad += bool(diff)
            print(f"{name}: {'IDENTICAL' if not diff else 'DIFFERS in ' + ', '.join(diff)}")
        else:
            np.savez_compressed(path, **out)
            print(f"{name}: {int(out['steps'])} steps, {int(net.spike_history.sum())} spikes")
    sys.exit(1 if bad else 0)

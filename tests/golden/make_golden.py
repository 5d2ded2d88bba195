#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz with the CPU oracle:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_cases  # noqa: E402

for name, build in golden_cases.CASES.items():
    net, steps = build()
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=bool(net.n_cells))
    out = golden_cases.outputs(name, net, steps)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {steps} steps, {int(net.spike_history.sum())} spikes")

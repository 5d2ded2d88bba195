"""Cost of a generated neuron model next to the built-in one: the 256x256 dense
Izhikevich lattice of BASELINE config C2 stepped (a) by the default library's built-in model and (b) by a library
generated from the same model written in the neuron_builder! DSL (tests/test_modelgen.py::IZH_DSL, two-kernel step).
Run under `rocprofv3 --kernel-trace --stats` to get k_update<0> / k_update<100> per launch; prints ms/step of both."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import snn_amd                                         # noqa: E402
from snn_amd import _lib, modelgen, synthetic          # noqa: E402
from test_modelgen import IZH_DSL                      # noqa: E402
from test_modelgen_channels import HODGKIN_HUXLEY      # noqa: E402


def run(lib_path, model, rows, steps, warmup, hh=False):
    n = rows * rows
    dn = snn_amd.DeviceNetwork(model=model, lib_path=lib_path)
    dn.add_lattice(0, rows, rows)
    dn.finalize()
    dn.set_attr(0, "gap_conductance", np.full(n, 0.5 if hh else 10.0, np.float32))
    dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -70.0, -40.0) if hh else synthetic.uniform(1, n, -65.0, 30.0))
    dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
    dn.run(warmup)
    dn.synchronize()
    t0 = time.perf_counter()
    dn.run(steps)
    dn.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    v = dn.get_attr(0, "current_voltage")
    dn.close()
    return ms, v


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    hh = len(sys.argv) > 3 and sys.argv[3] == "hh"          # Hodgkin-Huxley (three gated channels) instead of Izhikevich
    generated = _lib.build_custom(modelgen.parse(HODGKIN_HUXLEY if hh else IZH_DSL))
    built_in_ms, v0 = run(None, snn_amd.HODGKIN_HUXLEY if hh else snn_amd.IZHIKEVICH, rows, steps, 10, hh)
    generated_ms, v1 = run(generated, snn_amd.CUSTOM, rows, steps, 10, hh)
    print(json.dumps({"model": "Hodgkin-Huxley" if hh else "Izhikevich", "lattice": f"{rows}x{rows} dense", "steps": steps,
                      "built_in_ms_per_step": built_in_ms, "generated_ms_per_step": generated_ms,
                      "final_voltages_identical": bool(np.array_equal(v0.view(np.uint32), v1.view(np.uint32))) if hh else None}))

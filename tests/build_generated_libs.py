"""Compile every generated library the GPU tests use (test infrastructure; `python tests/build_generated_libs.py`).
The tests build what is missing on demand -- this only front-loads the hipcc time (about 30 s per library, 4 at once)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def descriptions():
    import random_descriptions
    from test_gpu_modelgen import FUNCTIONS_DSL, RANDOM_DSL
    from test_modelgen import BOOL_DSL, IF_DSL, IZH_DSL, LIF_NB
    from test_modelgen_channels import CALCIUM_CLAMP, HODGKIN_HUXLEY, LEAK_NEURON, MORRIS_LECAR
    from test_modelgen_kinetics import APPROXIMATE_NT, BOUNDED_RC, DESTEXHE_PAIR, ELECTROCHEMICAL_REF, RESTATED_STEP
    from test_modelgen_receptors import IONOTROPIC_LIKE, LIF, MIXED, STEP_NEURON
    from test_modelgen_spike_trains import BURST_DSL, RATE_DSL, REFRACTORINESS_DSL
    from snn_amd.examples_dsl import LIXIRNET
    del random_descriptions
    facade = LEAK_NEURON.replace("vars: v_reset = -75, v_th = -55", "vars: v_reset = -75, v_th = -55, c_m = 25, ready = true") \
                        .replace("dv/dt = l.current + i", "dv/dt = (i - l.current) / c_m")
    return [LIF_NB, IZH_DSL, IF_DSL, CALCIUM_CLAMP, MORRIS_LECAR, FUNCTIONS_DSL, BOOL_DSL, ELECTROCHEMICAL_REF,
            RESTATED_STEP, HODGKIN_HUXLEY, *RANDOM_DSL, facade, RATE_DSL + REFRACTORINESS_DSL, APPROXIMATE_NT + BOUNDED_RC,
            IZH_DSL + BURST_DSL + DESTEXHE_PAIR, IZH_DSL + BURST_DSL,
            MIXED + LIF.format(name="MixedIntegrateAndFire", receptors="MixedReceptors"),
            IONOTROPIC_LIKE + STEP_NEURON.format(name="OwnReceptors", receptors="receptors: AmpaGabaReceptors\n    "),
            MIXED + STEP_NEURON.format(name="MixedStep", receptors="receptors: MixedReceptors\n    "), LIXIRNET]


if __name__ == "__main__":
    from snn_amd import _lib, modelgen
    models = [modelgen.parse_description(text) for text in descriptions()]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        paths = list(pool.map(_lib.build_custom, models))
    print(f"{len(paths)} libraries in {time.time() - t0:.0f} s")

#!/bin/bash
# round 6, GPU session 8: the quarter step with its products formed before the turns: parity, phase clocks, kernel times
set -u
OUT=$PWD/gpurun_out/r06_s8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_fused_step.py tests/test_gpu_network.py tests/test_gpu_models.py tests/test_gpu_golden.py tests/test_gpu_lixirnet_module.py tests/test_gpu_lixirnet_facade.py tests/test_gpu_modelgen.py tests/test_gpu_randomized.py tests/test_gpu_reward_network.py tests/test_gpu_reward.py tests/test_gpu_bcm.py tests/test_gpu_sequences.py tests/test_gpu_persistent_run.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests: exit $?"; tail -3 $OUT/tests.log | cut -c1-300
LAB=$PWD/spiking-neural-networks_amd/csrc/lab
for case in "16 1" "16 0" "22 1"; do set -- $case; echo "--- side $1 chem $2"; SNN_AMD_LIB=$LAB/libsnn_lab_timing.so python3 profiles/trace_small_step.py $1 $2 0 1200 2>&1 | grep -E "k_step_resident_q" | head -2; done
for case in "16 0" "16 1" "22 0" "22 1"; do
  set -- $case
  rm -rf $OUT/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 profiles/trace_small_step.py $1 $2 0 3000 > /dev/null 2> $OUT/trace.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/small_step_side${1}_chem${2}_quarters1_kernel_stats.csv 2>/dev/null
  echo "side $1 chem $2: $(grep k_step_resident $f | sed 's/.*",//' | cut -d, -f1-3)"
done
rm -rf $OUT/prof
python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices.jsonl 2> /dev/null
python3 profiles/measure_small_chem.py 3000 2> /dev/null | grep lattice > $OUT/small_chemical_lattices.jsonl
python3 - <<'PY'
import json
for f in ("small_plastic_lattices", "small_chemical_lattices"):
    for l in open(f"gpurun_out/r06_s8/{f}.jsonl"):
        if l.startswith("{"):
            d = json.loads(l)
            if d["lattice"] in ("16x16", "24x24"): print(f[:13], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (list, dict)) and k not in ("steps", "fallbacks")})
PY

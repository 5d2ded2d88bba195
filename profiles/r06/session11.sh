#!/bin/bash
# round 6, GPU sessions 11 and 12 (SESSION_DIR=r06_s12): the kernel arguments' lines requested in one round trip (warm_kernel_arguments) in k_update and the
# one-launch steps: C3's update under rocprofv3, the small-step kernels, parity of the paths touched
set -u
OUT=$PWD/gpurun_out/${SESSION_DIR:-r06_s11}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_fused_step.py tests/test_gpu_network.py tests/test_gpu_models.py tests/test_gpu_golden.py tests/test_gpu_randomized.py tests/test_gpu_sequences.py tests/test_gpu_persistent_run.py tests/test_gpu_modelgen.py tests/test_gpu_lixirnet_module.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests: exit $?"; tail -3 $OUT/tests.log | cut -c1-300
rm -rf $OUT/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 bench.py --config c3 --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline > $OUT/c3_bench_under_rocprof.json 2> $OUT/c3_rocprof.err
find $OUT/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
rm -rf $OUT/prof_c3
head -4 $OUT/c3_kernel_stats.csv | cut -c1-200
for i in 1 2; do python3 bench.py --config c3 --no-cpu-baseline > $OUT/c3_process_$i.json 2> /dev/null; python3 -c "
import json,sys
d=json.loads(open('$OUT/c3_process_$i.json').read().strip().splitlines()[-1]); print('c3 process $i: ms_per_step', round(d['ms_per_step'],5), 'frac', round(d['roofline']['frac'],4))"; done
LAB=$PWD/spiking-neural-networks_amd/csrc/lab
for case in "16 1" "16 0" "22 1"; do set -- $case; echo "--- side $1 chem $2"; SNN_AMD_LIB=$LAB/libsnn_lab_timing.so python3 profiles/trace_small_step.py $1 $2 0 1200 2>&1 | grep -E "k_step_resident_q" | head -2; done
for case in "16 0" "16 1" "22 0" "22 1" "32 1"; do
  set -- $case
  rm -rf $OUT/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 profiles/trace_small_step.py $1 $2 0 3000 > /dev/null 2> $OUT/trace.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/small_step_side${1}_chem${2}_kernel_stats.csv 2>/dev/null
  echo "side $1 chem $2: $(grep k_step_resident $f | sed 's/.*",//' | cut -d, -f1-3)"
done
rm -rf $OUT/prof
python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices.jsonl 2> /dev/null
python3 profiles/measure_small_chem.py 3000 2> /dev/null | grep lattice > $OUT/small_chemical_lattices.jsonl
# session 12: the sparse step with its arguments warmed -- C5 under rocprofv3 and what one rank of G does per step
rm -rf $OUT/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 bench.py --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events --no-cpu-baseline > $OUT/c5_bench_under_rocprof.json 2> $OUT/c5_rocprof.err
find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c5_kernel_stats.csv
rm -rf $OUT/prof_c5
head -3 $OUT/c5_kernel_stats.csv | cut -c1-200
python3 bench.py --config c5 --no-cpu-baseline > $OUT/c5_bench_default.json 2> /dev/null
python3 profiles/measure_c5_rank_step.py 2000 > $OUT/c5_rank_step.jsonl 2> /dev/null
python3 - <<'PY'
import json, os
out = os.environ.get("SESSION_DIR", "r06_s11")
d = json.loads(open(f"gpurun_out/{out}/c5_bench_default.json").read().strip().splitlines()[-1])
print("c5 default: us_per_step", round(d["ms_per_step"] * 1e3, 2), "events frac", round(d["roofline"]["frac"], 4))
for l in open(f"gpurun_out/{out}/c5_rank_step.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print("rank step", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
PY

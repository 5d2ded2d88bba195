#!/bin/bash
# round 6, GPU session 5c: every load of the neuron update hoisted to its start (update_neuron_at) + the quarter step's refinements:
# the whole GPU suite, then C3 / C5 and the small one-launch steps under rocprofv3.
set -u
OUT=$PWD/gpurun_out/r06_s5c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_guard_arena.py > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?"; tail -4 $OUT/gpu_suite.log | cut -c1-300
prof() {
    local name=$1; shift
    rm -rf "$OUT/prof_$name"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py "$@" --no-cpu-baseline > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}_rocprof.err"
    find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
    rm -rf "$OUT/prof_$name"
    head -4 "$OUT/${name}_kernel_stats.csv" | cut -c1-200
}
prof c3 --config c3 --steps 100 --warmup 10 --repeats 2
prof c5 --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
prof c2 --config c2 --steps 50 --warmup 10 --repeats 2
for i in 1 2; do python3 bench.py --config c3 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c3_process$i.json; done
python3 -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_s5c/c3_process*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
for case in "8 1" "16 0" "16 1" "22 0" "22 1" "32 1"; do
  set -- $case
  for q in 1 0; do
    rm -rf $OUT/prof
    SNN_AMD_RESIDENT_QUARTERS=$q rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 profiles/trace_small_step.py $1 $2 0 3000 > /dev/null 2> $OUT/trace.err
    f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
    cp $f $OUT/small_step_side${1}_chem${2}_quarters${q}_kernel_stats.csv 2>/dev/null
    echo "side $1 chem $2 quarters $q: $(grep k_step_resident $f | sed 's/.*",//' | cut -d, -f1-3)"
  done
done
rm -rf $OUT/prof
python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices.jsonl 2> /dev/null
python3 profiles/measure_small_chem.py 3000 2> /dev/null | grep lattice > $OUT/small_chemical_lattices.jsonl
python3 - <<'PY'
import json
for f in ("small_plastic_lattices", "small_chemical_lattices"):
    for l in open(f"gpurun_out/r06_s5c/{f}.jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(f[:13], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (list, dict)) and k not in ("steps", "fallbacks")})
PY

#!/bin/bash
# round 6, GPU session 5b: the one-launch step with a chunk's rows over four wavefronts (k_step_resident_q): parity (every test
# that steps a small dense network takes it by default) and its kernel time against the one-wavefront form.
set -u
OUT=$PWD/gpurun_out/r06_s5b
mkdir -p $OUT
export TMPDIR=/tmp
true
true
for case in "8 0" "8 1" "16 0" "16 1" "22 0" "22 1"; do
  set -- $case
  for q in 1 0; do
    rm -rf $OUT/prof
    SNN_AMD_RESIDENT_QUARTERS=$q rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 profiles/trace_small_step.py $1 $2 0 3000 > /dev/null 2> $OUT/trace.err
    f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
    cp $f $OUT/small_step_side${1}_chem${2}_quarters${q}_kernel_stats.csv 2>/dev/null
    echo "side $1 chem $2 quarters $q: $(grep k_step_resident $f | cut -d, -f1-4 | cut -c1-120)"
  done
done
rm -rf $OUT/prof
for q in 1 0; do SNN_AMD_RESIDENT_QUARTERS=$q python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices_quarters$q.jsonl 2> /dev/null; done
python3 - <<'PY'
import json
for q in (1, 0):
    for l in open(f"gpurun_out/r06_s5b/small_plastic_lattices_quarters{q}.jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print("quarters", q, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (list, dict))})
PY

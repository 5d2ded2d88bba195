#!/bin/bash
# round 6, GPU session 13 (final code): the whole GPU suite, smoke(), the default bench line, C5 once more (the sparse step's
# argument warming limited to the first 512 workgroups), what one rank of G does per C5 step
set -u
OUT=$PWD/gpurun_out/r06_s13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?"; tail -3 $OUT/gpu_suite.log | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > $OUT/bench_default_final.json 2> $OUT/bench_default_final.err; tail -1 $OUT/bench_default_final.json | cut -c1-400
rm -rf $OUT/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 bench.py --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events --no-cpu-baseline > $OUT/c5_bench_under_rocprof.json 2> $OUT/c5_rocprof.err
find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c5_kernel_stats.csv
rm -rf $OUT/prof_c5
head -3 $OUT/c5_kernel_stats.csv | cut -c1-200
python3 bench.py --config c5 --no-cpu-baseline > $OUT/c5_bench_default.json 2> /dev/null
python3 profiles/measure_c5_rank_step.py 2000 > $OUT/c5_rank_step.jsonl 2> /dev/null
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s13/c5_bench_default.json").read().strip().splitlines()[-1])
print("c5 default: us_per_step", round(d["ms_per_step"] * 1e3, 2), "events frac", round(d["roofline"]["frac"], 4))
for l in open("gpurun_out/r06_s13/c5_rank_step.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print("rank step", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
PY
ls gpurun_out/emulated_ranks_give_ups.txt 2>/dev/null && cat gpurun_out/emulated_ranks_give_ups.txt

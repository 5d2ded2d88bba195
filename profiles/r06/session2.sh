#!/bin/bash
# round 6, GPU session 2: the new tests of the round (exception barrier + allocation failures, native threads, the bench's multi-rank
# path on one GPU, the step image, the wide update), A/B of the two kernels (C5: csr_image 1 / 0; C3: update_all_planes 2 / 1),
# their rocprofv3 kernel stats, then the whole GPU suite.
set -u
OUT=$PWD/gpurun_out/r06_s2
mkdir -p $OUT
export TMPDIR=/tmp
for t in test_gpu_csr_image test_gpu_update_wide test_gpu_alloc_failures test_gpu_native_threads test_gpu_bench_ranks; do
  timeout 1500 python3 -m pytest tests/$t.py -m gpu -q -x > $OUT/$t.log 2>&1
  echo "$t: exit $?"; tail -4 $OUT/$t.log | cut -c1-300
done
for i in 1 2; do
  for v in 1 0; do SNN_AMD_CSR_IMAGE=$v python3 bench.py --config c5 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c5_image${v}_process$i.json; done
  for v in 2 1; do SNN_AMD_UPDATE_ALL_PLANES=$v python3 bench.py --config c3 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c3_update${v}_process$i.json; done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_s2/c[35]_*process*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], "kernel ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], d["state_sha256"][:12])
    except Exception as e:
        print(f, "unreadable", e)
PY
prof() {
    local name=$1; shift
    rm -rf "$OUT/prof_$name"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py "$@" --no-cpu-baseline > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}_rocprof.err"
    find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
    rm -rf "$OUT/prof_$name"
    head -5 "$OUT/${name}_kernel_stats.csv" | cut -c1-200
}
prof c5 --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
SNN_AMD_CSR_IMAGE=0 prof c5_plain --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
prof c3 --config c3 --steps 100 --warmup 10 --repeats 2
SNN_AMD_UPDATE_ALL_PLANES=1 prof c3_update1 --config c3 --steps 100 --warmup 10 --repeats 2
timeout 900 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_guard_arena.py > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?"; tail -5 $OUT/gpu_suite.log | cut -c1-300

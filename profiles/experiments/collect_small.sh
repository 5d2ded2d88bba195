set -u
R=r02h; OUT=$PWD/gpurun_out/$R; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
prof() { local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py "$@" --no-cpu-baseline > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}_rocprof.err"
  find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
  tail -1 "$OUT/${name}_bench_under_rocprof.json" > "$OUT/x"; mv "$OUT/x" "$OUT/${name}_bench_under_rocprof.json"; rm -rf "$OUT/prof_$name"; head -3 "$OUT/${name}_kernel_stats.csv" | cut -c1-140; }
prof c1 --config c1 --steps 2000 --warmup 50 --repeats 2 --no-kernel-events
prof lattice64 --config c2 --rows 64 --cols 64 --steps 2000 --warmup 50 --repeats 2 --no-kernel-events
python bench.py --config c1 > $OUT/c1_bench_persistent_run.json 2>/dev/null; tail -1 $OUT/c1_bench_persistent_run.json | cut -c1-200

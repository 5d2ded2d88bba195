set -u
OUT=$PWD/gpurun_out/r02e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c1 -- python3 bench.py --config c1 --steps 2000 --warmup 50 --repeats 2 --no-cpu-baseline --no-kernel-events > $OUT/c1_bench_under_rocprof.json 2> $OUT/c1.err
find $OUT/prof_c1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c1_kernel_stats.csv
tail -1 $OUT/c1_bench_under_rocprof.json > $OUT/x && mv $OUT/x $OUT/c1_bench_under_rocprof.json
rm -rf $OUT/prof_c1
head -5 $OUT/c1_kernel_stats.csv
python bench.py --config c1 > $OUT/c1_bench_default.json 2>/dev/null; tail -1 $OUT/c1_bench_default.json | cut -c1-300
# the same kernel at 64x64 (4 row groups per column tile, 256 workgroups = every CU)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_64 -- python3 bench.py --config c2 --rows 64 --cols 64 --steps 2000 --warmup 50 --repeats 2 --no-cpu-baseline --no-kernel-events > $OUT/lattice64_bench_under_rocprof.json 2> $OUT/l64.err
find $OUT/prof_64 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/lattice64_kernel_stats.csv
tail -1 $OUT/lattice64_bench_under_rocprof.json > $OUT/x && mv $OUT/x $OUT/lattice64_bench_under_rocprof.json
rm -rf $OUT/prof_64
head -4 $OUT/lattice64_kernel_stats.csv; cut -c1-250 $OUT/lattice64_bench_under_rocprof.json

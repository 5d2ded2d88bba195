#!/bin/bash
# Collects the per-round evidence on a GPU box:  bash profiles/collect.sh rNN [only]  (writes gpurun_out/rNN/, to be copied
# into profiles/rNN/; `only` = a substring of the names to (re)collect, e.g. c5).  Per config: rocprofv3 --kernel-trace --stats of bench.py (kernel stats csv + the bench line of
# that profiled process), then two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) reduced by summarize_pmc.py.
set -u
R=${1:-r06}
ONLY=${2:-}
want() { [ -z "$ONLY" ] || [[ "$1" == *"$ONLY"* ]]; }
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
DATE=$(date -u +%Y-%m-%d)
prof() {   # name, bench args...
    local name=$1; shift
    want "$name" || return 0
    rm -rf "$OUT/prof_$name"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py "$@" --no-cpu-baseline \
        > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}_rocprof.err"
    find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
    grep '^{"metric"' "$OUT/${name}_bench_under_rocprof.json" | tail -1 > "$OUT/${name}_bench_line.json"; mv "$OUT/${name}_bench_line.json" "$OUT/${name}_bench_under_rocprof.json"
    rm -rf "$OUT/prof_$name"
    head -4 "$OUT/${name}_kernel_stats.csv"
}
pmc() {    # name, kernel substring, bench args...
    local name=$1 needle=$2; shift 2
    want "$name" || return 0
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf "$OUT/pmc_${name}_$c"
        rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${name}_$c" -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events \
            > /dev/null 2> "$OUT/${name}_pmc_$c.err"
    done
    python3 profiles/summarize_pmc.py "$(find "$OUT/pmc_${name}_FETCH_SIZE" -name '*counter_collection.csv' | head -1)" \
        "$(find "$OUT/pmc_${name}_WRITE_SIZE" -name '*counter_collection.csv' | head -1)" "$OUT/${name}_pmc_traffic.json" "$needle" "$DATE" | tail -12
    rm -rf "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE"
}
prof c2 --config c2 --steps 50 --warmup 10 --repeats 2
prof c3 --config c3 --steps 100 --warmup 10 --repeats 2
prof c4 --config c4 --steps 50 --warmup 10 --repeats 2
prof c4_spiking_0p1pct --config c4 --spike-fraction 0.001 --steps 50 --warmup 100 --repeats 2
prof c6 --config c6 --steps 20 --warmup 3 --repeats 2
prof c5 --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
# (round 6: the same without the step image -- the plain one-launch step k_step_csr)
SNN_AMD_CSR_IMAGE=0 prof c5_plain_step --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
prof c1 --config c1 --steps 2000 --warmup 50 --repeats 2 --no-kernel-events
prof lattice64 --config c2 --rows 64 --cols 64 --steps 2000 --warmup 50 --repeats 2 --no-kernel-events
prof c2_sharded_world1 --config c2 --force-sharded --steps 50 --warmup 10 --repeats 2
prof c5_sharded_world1 --config c5 --force-sharded --steps 500 --warmup 20 --repeats 2 --no-kernel-events
pmc c2 k_inputs_dense --config c2 --steps 20 --warmup 3 --repeats 1
pmc c3 k_inputs_dense --config c3 --steps 20 --warmup 3 --repeats 1
pmc c4 k_inputs_dense --config c4 --steps 20 --warmup 3 --repeats 1
pmc c6 k_inputs_rstdp --config c6 --steps 10 --warmup 2 --repeats 1
# (c5: 4-byte-per-lane accesses, a width the guide calls uncalibrated; the doubled FETCH_SIZE is kept because the undoubled
#  figure, 79 MB, is below the 117 MB of plan words + weights the launch has to stream)
pmc c5 k_step_csr --config c5 --steps 50 --warmup 5 --repeats 1
if want c5; then
# what ONE rank of G does per step, without its exchange (the library's loop with a transport that moves nothing)
python3 profiles/measure_c5_rank_step.py 2000 > "$OUT/c5_rank_step.jsonl" 2> /dev/null
PEER=1 python3 profiles/measure_c5_rank_step.py 2000 > "$OUT/c5_rank_step_peer_form.jsonl" 2> /dev/null
python3 bench.py --config c5 --no-cpu-baseline > "$OUT/c5_bench_default.json" 2> /dev/null
rm -rf "$OUT/trace_g8"
( cd /tmp && SHARDS=8 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_g8" -- python3 "$OLDPWD/profiles/measure_c5_rank_step.py" 1000 > /dev/null 2>&1 )
python3 profiles/experiments/kernel_gaps.py "$(find "$OUT/trace_g8" -name '*kernel_trace.csv' | head -1)" 400 > "$OUT/c5_rank_step_g8_kernel_trace_summary.json"
rm -rf "$OUT/trace_g8"
fi
if [ -z "$ONLY" ]; then
python3 profiles/measure_shard_shapes.py 200 > "$OUT/c2_shard_shapes.jsonl" 2> /dev/null
# small lattices with chemical synapses: the one-launch run against one launch per step
python3 profiles/measure_small_chem.py 3000 2> /dev/null | grep lattice > "$OUT/small_chemical_lattices.jsonl"
# small plastic lattices: the STDP updates inside the one-launch run against one launch per step + the scatter kernels
python3 profiles/measure_small_stdp.py 3000 2> /dev/null | grep lattice > "$OUT/small_stdp_lattices.jsonl"
# the default lines (no profiler): headline, c1 with the latency roofline, c3, c5
python3 bench.py > "$OUT/bench_default.json" 2> /dev/null
python3 bench.py --config c1 --no-cpu-baseline > "$OUT/c1_bench_default.json" 2> /dev/null
python3 bench.py --config c3 --no-cpu-baseline > "$OUT/c3_bench_default.json" 2> /dev/null
# round 5: C3 in four fresh processes, with and without the closing input pass; C4 with 1 % of the neurons spiking; small plastic lattices
for i in 1 2 3 4; do python3 bench.py --config c3 --no-cpu-baseline > "$OUT/c3_fresh_process_$i.json" 2> /dev/null; done
# (the two-kernel step is the default -- the four lines above; this arm is the closing pass, option "dense_close" 1)
for i in 1 2; do SNN_AMD_DENSE_CLOSE=1 python3 bench.py --config c3 --no-cpu-baseline > "$OUT/c3_closing_pass_process_$i.json" 2> /dev/null; done
python3 bench.py --config c4 --spike-fraction 0.01 --steps 50 --warmup 100 --repeats 2 --no-cpu-baseline > "$OUT/c4_spiking_1pct_bench.json" 2> /dev/null
python3 profiles/measure_small_plastic.py 3000 > "$OUT/small_plastic_lattices.jsonl" 2> /dev/null
fi
ls "$OUT"

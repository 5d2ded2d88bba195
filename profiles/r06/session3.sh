#!/bin/bash
# round 6, GPU session 3: (a) the tests that failed in session 2 (test bugs; one fall-back hook), (b) A/B of the step image's record
# ring (4 = the library, 6 and 8 = lab builds) against the plain step, and of the three forms of the chemical update,
# (c) the whole GPU suite, (d) the trap at a rate that means something: --lean workers (trap + malloc perturbation + host poison
# only) at campaign E's load, SNN_AMD_PINNED_COPIES=0 (the configuration of every event so far) and 1 (the default).
set -u
OUT=$PWD/gpurun_out/r06_s3
mkdir -p $OUT
export TMPDIR=/tmp
for t in test_gpu_csr_image test_gpu_alloc_failures test_gpu_bench_ranks; do
  timeout 1500 python3 -m pytest tests/$t.py -m gpu -q -x > $OUT/$t.log 2>&1
  echo "$t: exit $?"; tail -3 $OUT/$t.log | cut -c1-300
done
LAB=$PWD/spiking-neural-networks_amd/csrc/lab
for i in 1 2; do
  python3 bench.py --config c5 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c5_ring4_process$i.json
  SNN_AMD_CSR_IMAGE=0 python3 bench.py --config c5 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c5_plain_process$i.json
  for r in 6 8; do SNN_AMD_LIB=$LAB/libsnn_lab_ring$r.so python3 bench.py --config c5 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c5_ring${r}_process$i.json; done
  for v in 1 2 3; do SNN_AMD_UPDATE_ALL_PLANES=$v python3 bench.py --config c3 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c3_update${v}_process$i.json; done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_s3/c[35]_*process*.json")):
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
    head -4 "$OUT/${name}_kernel_stats.csv" | cut -c1-200
}
prof c5 --config c5 --steps 500 --warmup 20 --repeats 2 --no-kernel-events
prof c3 --config c3 --steps 100 --warmup 10 --repeats 2
SNN_AMD_UPDATE_ALL_PLANES=3 prof c3_update3 --config c3 --steps 100 --warmup 10 --repeats 2
timeout 1200 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_guard_arena.py > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?"; tail -6 $OUT/gpu_suite.log | cut -c1-300
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
seed=30000000
for v in 0 1; do
  SNN_AMD_PINNED_COPIES=$v timeout 1300 python3 tests/campaign.py --lean --minutes ${CAMPAIGN_MINUTES:-14} --workers 24 --streamers 3 --first-seed $seed \
      --out $OUT/lean_pinned$v --tests $TESTS > $OUT/lean_pinned$v.log 2>&1
  seed=$((seed + 2000000))
  rm -rf $OUT/lean_pinned$v/repro/*/checkpoint* 2>/dev/null
  python3 -c "
import json
d=json.load(open('$OUT/lean_pinned$v/summary.json'))
print('lean, pinned_copies $v', {k:d.get(k) for k in ('wall_s','executions','failures','trap_faults','trap_calls','trap_buffers_retired')}, 'trap reports', len(d.get('trap_reports', [])))
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:400].replace(chr(10),' | '))
for r in d.get('trap_fault_records', [])[:6]: print('  TRAP FAULT', r)
for r in d.get('trap_reports', [])[:6]: print('  TRAP REPORT', r)"
  for f in $OUT/lean_pinned$v/guard-*.log; do [ -s "$f" ] && { echo "--- $f"; head -60 "$f" | cut -c1-220; }; done 2>/dev/null | head -150
done
du -sh $OUT

#!/bin/bash
# round 6, GPU session 4: the tests touched since session 3 (allocation failures, the step image on shard handles, the math main
# paths on all 2^32 patterns), C3 with the default update, what one rank of 8 does per C5 step with and without the step image, then
# the trap for forty minutes in the configuration of every event so far (SNN_AMD_PINNED_COPIES=0), --lean workers with one
# OpenMP thread each, campaign E's load.
set -u
OUT=$PWD/gpurun_out/r06_s4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_alloc_failures.py tests/test_gpu_halo_direct.py tests/test_gpu_csr.py tests/test_gpu_csr_image.py tests/test_gpu_update_wide.py tests/test_gpu_models.py tests/test_gpu_golden.py -m gpu -q > $OUT/tests_a.log 2>&1
echo "tests a: exit $?"; tail -4 $OUT/tests_a.log | cut -c1-300
timeout 1500 python3 -m pytest tests/test_gpu_math.py -m gpu -q -k "main_path" > $OUT/tests_math.log 2>&1
echo "math main paths: exit $?"; tail -2 $OUT/tests_math.log | cut -c1-300
timeout 900 python3 -m pytest tests/test_gpu_emulated_ranks.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/tests_b.log 2>&1
echo "tests b: exit $?"; tail -3 $OUT/tests_b.log | cut -c1-300
prof() {
    local name=$1; shift
    rm -rf "$OUT/prof_$name"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 bench.py "$@" --no-cpu-baseline > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}_rocprof.err"
    find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
    rm -rf "$OUT/prof_$name"
    head -4 "$OUT/${name}_kernel_stats.csv" | cut -c1-200
}
prof c3 --config c3 --steps 100 --warmup 10 --repeats 2
for i in 1 2; do python3 bench.py --config c3 --no-cpu-baseline 2> /dev/null | grep '^{"metric"' > $OUT/c3_process$i.json; done
python3 -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_s4/c3_process*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
SHARDS=8 python3 profiles/measure_c5_rank_step.py 2000 2> /dev/null > $OUT/c5_rank_step_g8_image.jsonl; cat $OUT/c5_rank_step_g8_image.jsonl | cut -c1-200
SNN_AMD_CSR_IMAGE=0 SHARDS=8 python3 profiles/measure_c5_rank_step.py 2000 2> /dev/null > $OUT/c5_rank_step_g8_plain.jsonl; cat $OUT/c5_rank_step_g8_plain.jsonl | cut -c1-200
python3 profiles/measure_c5_rank_step.py 2000 2> /dev/null > $OUT/c5_rank_step.jsonl; cut -c1-160 $OUT/c5_rank_step.jsonl
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
SNN_AMD_PINNED_COPIES=0 timeout 2900 python3 tests/campaign.py --lean --minutes ${CAMPAIGN_MINUTES:-40} --workers 24 --streamers 3 --first-seed 40000000 \
    --out $OUT/lean40_pinned0 --tests $TESTS > $OUT/lean40_pinned0.log 2>&1
rm -rf $OUT/lean40_pinned0/repro/*/checkpoint* 2>/dev/null
python3 -c "
import json
d=json.load(open('$OUT/lean40_pinned0/summary.json'))
print('lean 40 min, pinned_copies 0', {k:d.get(k) for k in ('wall_s','executions','failures','trap_faults','trap_calls','trap_buffers_retired')}, 'trap reports', len(d.get('trap_reports', [])))
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:400].replace(chr(10),' | '))
for r in d.get('trap_fault_records', [])[:6]: print('  TRAP FAULT', r)
for r in d.get('trap_reports', [])[:6]: print('  TRAP REPORT', r)"
for f in $OUT/lean40_pinned0/guard-*.log; do [ -s "$f" ] && { echo "--- $f"; head -80 "$f" | cut -c1-220; }; done 2>/dev/null | head -200
du -sh $OUT

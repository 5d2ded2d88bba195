#!/bin/bash
# round 6, GPU session 1: the trap (tests/guard_arena.py) on the real library, then the A/B prepared at the end of round 5
# (profiles/next_session_pinned_copies_ab.sh) with the trap armed: campaign E's load -- 24 armed workers + 3 streamers on the 16
# cores the container schedules -- with SNN_AMD_PINNED_COPIES = 0, 1, 2, twelve minutes each.
set -u
OUT=gpurun_out/r06_s1
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/nproc.txt 2>/dev/null; free -g | head -2 >> $OUT/nproc.txt
timeout 1500 python3 -m pytest tests/test_gpu_guard_arena.py -m gpu -q -x > $OUT/trap_on_the_real_library.log 2>&1
echo "trap on the real library: exit $?"; tail -5 $OUT/trap_on_the_real_library.log | cut -c1-300
SNN_AMD_PINNED_COPIES=2 timeout 400 python3 -m pytest tests/test_gpu_abi_errors.py tests/test_gpu_golden.py tests/test_gpu_reward_network.py tests/test_gpu_reduced_history.py -m gpu -q -x > $OUT/value2_first_run.log 2>&1
echo "value 2, first run: exit $?"; tail -2 $OUT/value2_first_run.log | cut -c1-200
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
seed=20000000
for v in 0 1 2; do
  SNN_AMD_PINNED_COPIES=$v timeout 1100 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-12} --workers 24 --streamers 3 --first-seed $seed \
      --out $OUT/campaign_pinned$v --tests $TESTS > $OUT/campaign_pinned$v.log 2>&1
  seed=$((seed + 1000000))
  rm -rf $OUT/campaign_pinned$v/repro/*/checkpoint* 2>/dev/null
  python3 -c "
import json
d=json.load(open('$OUT/campaign_pinned$v/summary.json'))
print('pinned_copies $v', {k:d.get(k) for k in ('wall_s','executions','failures','self_check_reports','trap_faults','trap_calls','trap_buffers_retired')}, 'guard reports', len(d.get('oracle_memory_reports', [])), 'trap reports', len(d.get('trap_reports', [])))
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:400].replace(chr(10),' | '))
for r in d.get('self_check_records', [])[:4]: print('  SELF-CHECK', r['test'], r['seed'], r['verify_reports'][0][:400])
for r in d.get('oracle_memory_reports', [])[:6]: print('  GUARD', r)
for r in d.get('trap_fault_records', [])[:6]: print('  TRAP FAULT', r)
for r in d.get('trap_reports', [])[:6]: print('  TRAP REPORT', r)"
  for f in $OUT/campaign_pinned$v/guard-*.log; do [ -s "$f" ] && { echo "--- $f"; head -60 "$f" | cut -c1-220; }; done 2>/dev/null | head -150
done
du -sh $OUT

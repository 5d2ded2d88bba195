#!/bin/bash
# round 5, GPU session 8 (the last minutes of the budget): campaign H, armed, 20 workers + 3 streamers -- the load of campaign E with the
# self-check that leaves caches and scratch out, names how each pass stepped and runs a third execution; oracle-memory guard scoped to
# the running test; the main process stops stragglers and always writes its summary
set -u
OUT=gpurun_out/r05_s8
mkdir -p $OUT
export TMPDIR=/tmp
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
timeout 1300 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-16} --workers 20 --streamers 3 --first-seed 12000000 --out $OUT/campaign_h --tests $TESTS > $OUT/campaign_h.log 2>&1
grep -c "MISMATCH" $OUT/campaign_h.log
grep "ORACLE MEMORY\|FAILURE" $OUT/campaign_h.log | head -10 | cut -c1-700
python3 -c "
import json
d=json.load(open('$OUT/campaign_h/summary.json'))
print({k:d.get(k) for k in ('wall_s','workers','executions','failures','executions_and_failures','self_check_reports','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:500].replace(chr(10),' | '))
for r in d.get('self_check_records', [])[:20]: print('  SELF-CHECK', r['test'], r['seed'], r['verify_reports'][0][:900])
for r in d.get('oracle_memory_reports', [])[:6]: print('  GUARD', r)"

#!/bin/bash
# round 5, GPU session 7: final code (closing pass off by default) -- full suite; then two campaigns side by side under the load that
# produced campaign E's reports: one ARMED (self-check with third execution + per-test log, oracle-memory guard, malloc perturbation,
# host poison), one PLAIN (no self-check: what the oracle comparison alone sees under that load)
set -u
OUT=gpurun_out/r05_s7
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: $(nproc) cpus"
timeout 900 python3 -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_dense_close.py -q > $OUT/tests_new.log 2>&1
echo "new tests exit $?" >> $OUT/tests_new.log
tail -4 $OUT/tests_new.log | cut -c1-300
timeout 1200 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -6 $OUT/tests.log | cut -c1-300
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
M=${CAMPAIGN_MINUTES:-38}
timeout 2700 python3 tests/campaign.py --minutes $M --workers 14 --streamers 3 --first-seed 10000000 --out $OUT/campaign_f --tests $TESTS > $OUT/campaign_f.log 2>&1 &
timeout 2700 python3 tests/campaign.py --minutes $M --workers 10 --streamers 3 --first-seed 11000000 --out $OUT/campaign_g --plain --tests $TESTS > $OUT/campaign_g.log 2>&1 &
wait
for c in f g; do
  grep -c "MISMATCH" $OUT/campaign_$c.log
  grep "ORACLE MEMORY\|FAILURE" $OUT/campaign_$c.log | head -10 | cut -c1-600
  python3 -c "
import json
d=json.load(open('$OUT/campaign_$c/summary.json'))
print('$c', {k:d.get(k) for k in ('wall_s','workers','executions','failures','executions_and_failures','self_check_reports','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:500].replace(chr(10),' | '))
for r in d.get('self_check_records', [])[:12]: print('  SELF-CHECK', r['test'], r['seed'], r['verify_reports'][0][:700])
for r in d.get('oracle_memory_reports', [])[:6]: print('  GUARD', r)"
done

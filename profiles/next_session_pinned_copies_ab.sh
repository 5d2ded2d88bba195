#!/bin/bash
# Prepared at the end of round 5 (not run: the GPU budget was spent) -- the experiment DESIGN.md section 7 "Open" asks for.
# Campaign E's load (24 armed workers + 3 streamers on the 16 cores the container schedules) three times, 12 minutes each:
#   SNN_AMD_PINNED_COPIES=0  the runtime stages the setters' / getters' pageable pointers itself (rounds 3 - 5 until the last hour)
#   SNN_AMD_PINNED_COPIES=1  copy_sync through the handle's page-locked buffer (the default since the end of round 5)
#   SNN_AMD_PINNED_COPIES=2  the 2-D copies (voltage history, trace rows) too -- NEVER RUN: start with the two test files below
# Every worker runs the oracle-memory guard, the self-check with its third execution, malloc perturbation and host poison; the
# summaries say, per event, the test, the seed and (guard) the call into the binding during which oracle memory changed.
set -u
OUT=gpurun_out/pinned_ab
mkdir -p $OUT
export TMPDIR=/tmp
SNN_AMD_PINNED_COPIES=2 timeout 300 python3 -m pytest tests/test_gpu_abi_errors.py tests/test_gpu_golden.py tests/test_gpu_reward_network.py -q -x > $OUT/value2_first_run.log 2>&1
echo "value 2, first run: exit $?"; tail -2 $OUT/value2_first_run.log | cut -c1-200
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
seed=20000000
for v in 0 1 2; do
  SNN_AMD_PINNED_COPIES=$v timeout 1100 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-12} --workers 24 --streamers 3 --first-seed $seed \
      --out $OUT/campaign_pinned$v --tests $TESTS > $OUT/campaign_pinned$v.log 2>&1
  seed=$((seed + 1000000))
  python3 -c "
import json
d=json.load(open('$OUT/campaign_pinned$v/summary.json'))
print('pinned_copies $v', {k:d.get(k) for k in ('wall_s','executions','failures','self_check_reports')}, 'guard reports', len(d.get('oracle_memory_reports', [])))
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:400].replace(chr(10),' | '))
for r in d.get('self_check_records', [])[:6]: print('  SELF-CHECK', r['test'], r['seed'], r['verify_reports'][0][:600])
for r in d.get('oracle_memory_reports', [])[:6]: print('  GUARD', r)"
done

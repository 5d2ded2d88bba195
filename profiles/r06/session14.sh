#!/bin/bash
# round 6, GPU session 14: campaign E of round 5 ONCE MORE, as it was -- 24 armed workers + 3 streamers on the container's 16 cores,
# the same tests and first seed, "pinned_copies" 0, NO trap (the trap's arena takes the buffers out of the malloc heap, which may be
# what the stray writer needed) -- with this round's library.  E saw 2 events in >= 58 068 executions of 35 minutes.
set -u
OUT=$PWD/gpurun_out/r06_s14
mkdir -p $OUT
export TMPDIR=/tmp
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
SNN_AMD_PINNED_COPIES=0 timeout 2250 python3 tests/campaign.py --trap 0 --minutes ${CAMPAIGN_MINUTES:-32} --workers 24 --streamers 3 --first-seed 9000000 \
    --out $OUT/campaign_e_again --tests $TESTS > $OUT/campaign_e_again.log 2>&1
tail -3 $OUT/campaign_e_again.log | cut -c1-300
rm -rf $OUT/campaign_e_again/repro/*/checkpoint* 2>/dev/null
python3 -c "
import json
d=json.load(open('$OUT/campaign_e_again/summary.json'))
print('campaign E again (no trap, pinned_copies 0)', {k:d.get(k) for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:1200].replace(chr(10),' | '))
print('self-check reports', d.get('self_check_reports'))
for r in d.get('self_check_records', [])[:4]: print('  REPORT', str(r)[:400])"
du -sh $OUT | cut -f1

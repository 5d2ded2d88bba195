#!/bin/bash
# round 5, GPU session 2: the full GPU suite on the new code, the row half of STDP fused into the input pass (A/B), a first
# armed campaign
set -u
OUT=gpurun_out/r05_s2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -25 $OUT/tests.log
for mode in 0 3; do
  for f in 0.01 0.001 0; do
    SNN_AMD_DEFER_STDP=$mode timeout 600 python3 bench.py --config c4 --spike-fraction $f --steps 50 --warmup 100 --repeats 2 --no-cpu-baseline \
        > $OUT/c4_f${f}_defer${mode}.json 2> $OUT/c4_f${f}_defer${mode}.err
    python3 -c "
import json,sys
d=json.load(open('$OUT/c4_f${f}_defer${mode}.json'))
print('defer $mode f $f ms/step', round(d['ms_per_step'],3), 'plasticity', round(d['plasticity']['ms_per_step'],4), 'input kernel', round(d['roofline'].get('avg_launch_ms') or 0,4), 'sha', d['state_sha256'][:12])"
  done
done
(cd /tmp && SNN_AMD_DEFER_STDP=3 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_defer3 -- \
    python3 $GRAFT_REPO_ROOT/bench.py --config c4 --spike-fraction 0.01 --steps 20 --warmup 30 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
f=$(find $OUT/prof_defer3 -name '*kernel_stats.csv' | head -1)
echo "== defer 3 kernel stats"; grep -i "stdp\|inputs_dense\|compact\|fill" "$f" | cut -c1-170
cp "$f" $OUT/c4_1pct_defer3_kernel_stats.csv
rm -rf $OUT/prof_defer3
# campaign A of round 5: armed workers (checkpoints at every run call, verify, malloc perturb, host poison), default hardware queues
timeout 1700 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-22} --workers 11 --streamers 3 --first-seed 5000000 --out $OUT/campaign_a \
    --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices \
    > $OUT/campaign_a.log 2>&1
tail -3 $OUT/campaign_a.log
python3 -c "
import json
d=json.load(open('$OUT/campaign_a/summary.json'))
print({k:d[k] for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:5]: print(r['test'], r['seed'], r['message'][:600])"

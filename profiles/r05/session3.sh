#!/bin/bash
# round 5, GPU session 3: the full GPU suite, the row half of STDP in the input pass at the plain pass's occupancy (A/B), the
# prefetching sparse step (A/B of two builds), campaign B
set -u
OUT=gpurun_out/r05_s3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -12 $OUT/tests.log
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
for lib in default prefetch; do
  for rep in 1 2; do
    if [ $lib = prefetch ]; then export SNN_AMD_LIB=$PWD/scratch/libsnn_amd_prefetch.so; else unset SNN_AMD_LIB; fi
    timeout 600 python3 bench.py --config c5 --no-cpu-baseline > $OUT/c5_${lib}_$rep.json 2> $OUT/c5_${lib}_$rep.err
    python3 -c "
import json
d=json.load(open('$OUT/c5_${lib}_$rep.json'))
print('c5 $lib $rep us/step', round(d['ms_per_step']*1000,2), 'kernel us', round((d['roofline'].get('avg_launch_ms') or 0)*1000,2), 'frac', round(d['roofline']['frac'],4), 'sha', d['state_sha256'][:12])"
  done
done
unset SNN_AMD_LIB
timeout 1800 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-24} --workers 11 --streamers 3 --first-seed 6000000 --out $OUT/campaign_b \
    --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices \
    > $OUT/campaign_b.log 2>&1
tail -3 $OUT/campaign_b.log
python3 -c "
import json
d=json.load(open('$OUT/campaign_b/summary.json'))
print({k:d[k] for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:5]: print(r['test'], r['seed'], r['message'][:800])"

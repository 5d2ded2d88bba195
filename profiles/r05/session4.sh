#!/bin/bash
# round 5, GPU session 4: full suite; the quad form of the column scatter with its registers fixed (A/B); small plastic lattices with the
# STDP of a step in one launch (A/B); what the placement selection sees at C3; campaign C
set -u
OUT=gpurun_out/r05_s4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -12 $OUT/tests.log | cut -c1-300
for form in 0 1; do
  for f in 0.01 0.001; do
    SNN_AMD_STDP_COLUMNS_FORM=$form timeout 600 python3 bench.py --config c4 --spike-fraction $f --steps 50 --warmup 100 --repeats 2 --no-cpu-baseline \
        > $OUT/c4_f${f}_form${form}.json 2> $OUT/c4_f${f}_form${form}.err
    python3 -c "
import json,sys
d=json.load(open('$OUT/c4_f${f}_form${form}.json'))
print('form $form f $f ms/step', round(d['ms_per_step'],3), 'plasticity', round(d['plasticity']['ms_per_step'],4), 'sha', d['state_sha256'][:12])"
  done
done
(cd /tmp && SNN_AMD_STDP_COLUMNS_FORM=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_form1 -- \
    python3 $GRAFT_REPO_ROOT/bench.py --config c4 --spike-fraction 0.01 --steps 20 --warmup 30 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
f=$(find $OUT/prof_form1 -name '*kernel_stats.csv' | head -1)
echo "== form 1 kernel stats"; grep -i "stdp\|compact" "$f" | cut -c1-170
cp "$f" $OUT/c4_1pct_quad_form_kernel_stats.csv; rm -rf $OUT/prof_form1
timeout 900 python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices.jsonl 2> $OUT/small_plastic.err
python3 -c "
import json
for l in open('$OUT/small_plastic_lattices.jsonl'):
    d=json.loads(l); print(d['lattice'], d['synapses'], 'cells', d['rate_cells'], 'one launch' if d['stdp_in_one_launch'] else 'four launches', round(d['us_per_step'],2))"
SNN_DEBUG_PLACEMENT=1 timeout 300 python3 bench.py --config c3 --no-cpu-baseline > $OUT/c3_placement.json 2> $OUT/c3_placement.err
grep "placement" $OUT/c3_placement.err | head -8
python3 -c "
import json
d=json.load(open('$OUT/c3_placement.json')); print('c3 us/step', round(d['ms_per_step']*1000,1), 'frac', round(d['roofline']['frac'],4))"
timeout 2200 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-30} --workers 12 --streamers 3 --first-seed 7000000 --out $OUT/campaign_c \
    --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices \
    > $OUT/campaign_c.log 2>&1
tail -3 $OUT/campaign_c.log | cut -c1-400
python3 -c "
import json
d=json.load(open('$OUT/campaign_c/summary.json'))
print({k:d[k] for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:5]: print(r['test'], r['seed'], r['message'][:800])"

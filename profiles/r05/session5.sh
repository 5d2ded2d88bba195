#!/bin/bash
# round 5, GPU session 5: the closing input pass (k_inputs_dense_close) -- parity, then C3 / C2 / C4 with and without it; the self-check
# on handles with weight updates; full suite; campaign D (armed, the self-check now covering plastic handles)
set -u
OUT=gpurun_out/r05_s5
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: $(nproc) cpus, $(free -g | awk '/Mem/{print $2}') GiB"
timeout 900 python3 -m pytest tests/test_gpu_dense_close.py tests/test_gpu_checkpoint.py -q -x > $OUT/tests_new.log 2>&1
echo "new tests exit $?" >> $OUT/tests_new.log
tail -15 $OUT/tests_new.log | cut -c1-400
for rep in 1 2; do
  for close in 1 0; do
    SNN_AMD_DENSE_CLOSE=$close timeout 300 python3 bench.py --config c3 --no-cpu-baseline > $OUT/c3_close${close}_$rep.json 2> $OUT/c3_close${close}_$rep.err
    python3 -c "
import json
d=json.load(open('$OUT/c3_close${close}_$rep.json')); print('c3 close $close rep $rep us/step', round(d['ms_per_step']*1000,1), 'kernel frac', round(d['roofline']['frac'],4), 'sha', d['state_sha256'][:12])"
  done
done
for close in 1 0; do
  SNN_AMD_DENSE_CLOSE=$close timeout 300 python3 bench.py --no-cpu-baseline > $OUT/c2_close${close}.json 2> $OUT/c2_close${close}.err
  python3 -c "
import json
d=json.load(open('$OUT/c2_close${close}.json')); print('c2 close $close ms/step', round(d['ms_per_step'],4), 'kernel frac', round(d['roofline']['frac'],4), 'value', round(d['value']), 'sha', d['state_sha256'][:12])"
  SNN_AMD_DENSE_CLOSE=$close timeout 300 python3 bench.py --config c4 --no-cpu-baseline > $OUT/c4_close${close}.json 2> $OUT/c4_close${close}.err
  python3 -c "
import json
d=json.load(open('$OUT/c4_close${close}.json')); print('c4 close $close ms/step', round(d['ms_per_step'],4), 'kernel frac', round(d['roofline']['frac'],4), 'sha', d['state_sha256'][:12])"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -- \
    python3 $GRAFT_REPO_ROOT/bench.py --config c3 --no-cpu-baseline > /dev/null 2>&1)
f=$(find $OUT/prof_c3 -name '*kernel_stats.csv' | head -1)
echo "== c3 kernel stats (closing pass)"; head -6 "$f" | cut -c1-200
cp "$f" $OUT/c3_close_kernel_stats.csv; rm -rf $OUT/prof_c3
# the self-check (every run call twice, outcomes compared on the device) over the files whose handles update weights
SNN_AMD_VERIFY=1 timeout 1500 python3 -m pytest tests/test_gpu_randomized.py tests/test_gpu_sequences.py tests/test_gpu_reward_network.py \
    tests/test_gpu_reward.py tests/test_gpu_persistent_stdp.py tests/test_gpu_stdp_load.py tests/test_gpu_bcm.py -q > $OUT/tests_verify.log 2>&1
echo "verify tests exit $?" >> $OUT/tests_verify.log
tail -6 $OUT/tests_verify.log | cut -c1-400
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -8 $OUT/tests.log | cut -c1-300
timeout 2900 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-40} --workers 12 --streamers 3 --first-seed 8000000 --out $OUT/campaign_d \
    --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices \
    > $OUT/campaign_d.log 2>&1
tail -3 $OUT/campaign_d.log | cut -c1-400
python3 -c "
import json
d=json.load(open('$OUT/campaign_d/summary.json'))
print({k:d[k] for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:5]: print(r['test'], r['seed'], r['message'][:800])"

#!/bin/bash
# round 5, GPU session 6: the one-launch step of the generated neuron model and the closing pass with its columns' sums together
# (tests); what a small per-step network spends where (kernel traces); the round's collection; full suite; campaign E
set -u
OUT=gpurun_out/r05_s6
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: $(nproc) cpus, $(free -g | awk '/Mem/{print $2}') GiB"
timeout 1500 python3 -m pytest tests/test_gpu_dense_close.py tests/test_gpu_lixirnet_module.py tests/test_gpu_modelgen.py tests/test_gpu_checkpoint.py -q -x > $OUT/tests_new.log 2>&1
echo "new tests exit $?" >> $OUT/tests_new.log
tail -8 $OUT/tests_new.log | cut -c1-400
for cfg in "8 1 0" "16 1 0" "16 0 0" "16 1 1" "31 1 0"; do
  set -- $cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_small -- \
      python3 $GRAFT_REPO_ROOT/profiles/trace_small_step.py $1 $2 $3 2000 > /dev/null 2>&1)
  f=$(find $OUT/prof_small -name '*kernel_stats.csv' | head -1)
  echo "== side $1 chemical $2 plastic $3"; head -5 "$f" | cut -c1-150
  cp "$f" $OUT/small_step_side$1_chem$2_plastic$3_kernel_stats.csv; rm -rf $OUT/prof_small
done
# the closing pass against the two-kernel step, same box: C3 (64 chunks) and C2 (256 chunks)
for cfg in c3 c2; do
  for close in 1 0 1 0; do
    SNN_AMD_DENSE_CLOSE=$close timeout 300 python3 bench.py --config $cfg --no-cpu-baseline >> $OUT/ab_${cfg}_close${close}.jsonl 2> $OUT/ab_${cfg}_close${close}.err
  done
done
python3 - <<'PY' > $OUT/ab_choice.env
import json
def med(path):
    v = sorted(json.loads(l)["ms_per_step"] for l in open(path) if l.startswith("{"))
    return v[len(v) // 2] if v else 1e9
out = {}
for cfg in ("c3", "c2"):
    out[cfg] = (med(f"gpurun_out/r05_s6/ab_{cfg}_close1.jsonl"), med(f"gpurun_out/r05_s6/ab_{cfg}_close0.jsonl"))
print(f"# ms per step (closing pass, two kernels): c3 {out['c3']}, c2 {out['c2']}")
if out["c3"][0] >= out["c3"][1]:
    print("export SNN_AMD_DENSE_CLOSE=0")
elif out["c2"][0] > out["c2"][1] * 1.002:
    print("export SNN_AMD_DENSE_CLOSE_MAX_CHUNKS=128")
PY
cat $OUT/ab_choice.env
. $OUT/ab_choice.env
bash profiles/collect.sh r05 > $OUT/collect.log 2>&1
tail -5 $OUT/collect.log | cut -c1-300
python3 -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/c3_*process*.json'))+['gpurun_out/r05/bench_default.json','gpurun_out/r05/c3_bench_default.json','gpurun_out/r05/c4_spiking_1pct_bench.json']:
    try:
        d=json.load(open(f)); print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'value', round(d['value']))
    except Exception as e: print(f, 'unreadable', e)"
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -8 $OUT/tests.log | cut -c1-300
timeout 2700 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-35} --workers 24 --streamers 3 --first-seed 9000000 --out $OUT/campaign_e \
    --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices \
    > $OUT/campaign_e.log 2>&1
tail -3 $OUT/campaign_e.log | cut -c1-400
python3 -c "
import json
d=json.load(open('$OUT/campaign_e/summary.json'))
print({k:d[k] for k in ('wall_s','executions','failures','executions_and_failures','ras_errors_before_ue_ce','ras_errors_after_ue_ce')})
for r in d['failure_records'][:5]: print(r['test'], r['seed'], r['message'][:800])"

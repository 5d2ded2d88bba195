#!/bin/bash
# round 5, GPU session 1: the new test-support pieces and the quad form of the STDP column scatter (A/B on one box)
set -u
OUT=gpurun_out/r05_s1
mkdir -p $OUT
export TMPDIR=/tmp
python3 tests/checkpoint.py --ras > $OUT/ras_before.json 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_abi_errors.py tests/test_gpu_persistent_stdp.py tests/test_gpu_stdp_load.py \
    tests/test_gpu_persistent_run.py tests/test_gpu_sequences.py tests/test_gpu_randomized.py tests/test_gpu_network.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests exit $?" >> $OUT/tests.log
tail -15 $OUT/tests.log
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
for form in 0 1; do
  (cd /tmp && SNN_AMD_STDP_COLUMNS_FORM=$form timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_form$form -- \
      python3 $GRAFT_REPO_ROOT/bench.py --config c4 --spike-fraction 0.01 --steps 20 --warmup 30 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find $OUT/prof_form$form -name '*kernel_stats.csv' | head -1)
  echo "== form $form kernel stats"; grep -i "stdp\|inputs_dense\|compact" "$f" | cut -c1-160
  cp "$f" $OUT/c4_1pct_form${form}_kernel_stats.csv
  rm -rf $OUT/prof_form$form
done

#!/bin/bash
# round 6, last GPU session: the per-config collection (profiles/collect.sh r06), the first-contact script rehearsed on one GPU
# (ranks as processes, gloo, host-staged collectives), and the whole GPU suite on the final code.
set -u
OUT=$PWD/gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
bash profiles/collect.sh r06 > $OUT/collect.log 2>&1; echo "collect: exit $?"; tail -3 $OUT/collect.log | cut -c1-200
timeout 900 python3 profiles/first_contact.py --emulate --gpus-list 1,2 --rows 48 --steps 20 --warmup 5 --skip-tests --out $OUT/first_contact_rehearsal > $OUT/first_contact_rehearsal.log 2>&1
echo "first-contact rehearsal: exit $?"; tail -14 $OUT/first_contact_rehearsal.log | cut -c1-220
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?"; tail -4 $OUT/gpu_suite.log | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > $OUT/bench_default_final.json 2> $OUT/bench_default_final.err; tail -c 1500 $OUT/bench_default_final.json
du -sh $OUT

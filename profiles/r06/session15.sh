#!/bin/bash
# round 6, GPU session 15: the libraries rebuilt after a comment-only change of the header (new source hash, same code): smoke(), the
# ABI / fused-step / emulated-rank tests, one short bench line
set -u
OUT=$PWD/gpurun_out/r06_s15
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 420 python3 -m pytest tests/test_gpu_network.py tests/test_gpu_abi_errors.py tests/test_gpu_fused_step.py tests/test_gpu_emulated_ranks.py tests/test_gpu_golden.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests: exit $?"; tail -2 $OUT/tests.log | cut -c1-200
python3 bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline 2> /dev/null | tail -1 | cut -c1-300

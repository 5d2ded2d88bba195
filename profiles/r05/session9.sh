#!/bin/bash
# round 5, GPU session 9 (the last minutes): the final build -- a handful of test files as they are, and again with every host <-> device
# copy of the setters and getters through the handle's page-locked buffer (option "pinned_copies", the experiment prepared for round 6)
set -u
OUT=gpurun_out/r05_s9
mkdir -p $OUT
FILES="tests/test_gpu_abi_errors.py tests/test_gpu_golden.py tests/test_gpu_checkpoint.py tests/test_gpu_izhikevich_electrical.py"
timeout 170 python3 -m pytest $FILES -q -x > $OUT/default.log 2>&1; echo "default exit $?" | tee -a $OUT/default.log; tail -2 $OUT/default.log | cut -c1-200
SNN_AMD_PINNED_COPIES=1 timeout 150 python3 -m pytest $FILES -q -x > $OUT/pinned_copies.log 2>&1; echo "pinned exit $?" | tee -a $OUT/pinned_copies.log; tail -2 $OUT/pinned_copies.log | cut -c1-200

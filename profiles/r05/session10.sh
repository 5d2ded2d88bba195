#!/bin/bash
# round 5, GPU session 10 (what is left of the budget): the whole GPU suite with option "pinned_copies" on
set -u
OUT=gpurun_out/r05_s10
mkdir -p $OUT
SNN_AMD_PINNED_COPIES=1 timeout 345 python3 -m pytest tests -m gpu -q > $OUT/suite_pinned_copies.log 2>&1; echo "exit $?" | tee -a $OUT/suite_pinned_copies.log
grep -E "passed|failed" $OUT/suite_pinned_copies.log | tail -2 | cut -c1-200

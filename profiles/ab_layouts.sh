#!/bin/bash
# Same-box A/B of two builds of the library on the small configs (box-to-box HBM variance is ~7 %, so layouts are only
# compared inside one gpurun call).  The second library is any earlier build copied next to the product one, e.g. the
# last row-major commit:  git archive 8eb673e spiking-neural-networks_amd/csrc include | tar -x -C /tmp/old && hipcc ... -o
# spiking-neural-networks_amd/csrc/libsnn_amd_rowmajor.so  (git-ignored; selected through SNN_AMD_LIB).
mkdir -p gpurun_out/ab4
for i in 1 2 3; do
for lib in rowmajor quad; do
  if [ $lib = rowmajor ]; then export SNN_AMD_LIB=$PWD/spiking-neural-networks_amd/csrc/libsnn_amd_rowmajor.so; else unset SNN_AMD_LIB; fi
  for c in c3 c1; do
    python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib $c', d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('kernel_ms'))
" | tee -a gpurun_out/ab4/ab.txt
  done
done
done
unset SNN_AMD_LIB
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee -a gpurun_out/ab4/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/ab4/tests.txt

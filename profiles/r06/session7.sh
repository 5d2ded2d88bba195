#!/bin/bash
# round 6, GPU session 7: (a) where the time of the small chemical one-launch step goes (s_memtime phases, lab build), (b) the
# peer-form tests with ranks emulated as threads, repeated, on the library before and after the neuron update's loads were hoisted
set -u
OUT=$PWD/gpurun_out/r06_s7
mkdir -p $OUT
export TMPDIR=/tmp
LAB=$PWD/spiking-neural-networks_amd/csrc/lab
for case in "16 1" "16 0" "22 1"; do set -- $case; echo "--- side $1 chem $2"; SNN_AMD_LIB=$LAB/libsnn_lab_timing.so python3 profiles/trace_small_step.py $1 $2 0 1200 2>&1 | grep -E "k_step_resident_q|steps_dense" | head -4; done
for lib in head before_hoist; do
  fails=0
  for i in 1 2 3 4 5 6 7 8; do
    SNN_AMD_LIB=$LAB/libsnn_lab_$lib.so SNN_EMULATED_RANKS_CHILD=1 timeout 300 python3 -m pytest tests/test_gpu_halo_peer.py -m gpu -q -p no:cacheprovider > $OUT/peer_${lib}_$i.log 2>&1 || fails=$((fails+1))
  done
  echo "peer-form tests, library $lib: $fails of 8 runs failed"; grep -h "^FAILED" $OUT/peer_${lib}_*.log | sort | uniq -c
done

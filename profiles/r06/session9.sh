#!/bin/bash
# round 6, GPU session 9: (a) the quarter step with ONE round trip in its prologue and the update's cache lines requested by
# wavefront 0: parity, phase clocks, kernel times; (b) the emulated-rank peer-form tests, repeated, after the two device-wide
# synchronisations were taken out of their path (hipFree in agree_on_exchange, torch.cuda.synchronize in ThreadCollectives)
set -u
OUT=$PWD/gpurun_out/${SESSION_DIR:-r06_s9}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_fused_step.py tests/test_gpu_network.py tests/test_gpu_models.py tests/test_gpu_randomized.py tests/test_gpu_sequences.py tests/test_gpu_persistent_run.py tests/test_gpu_library_loop_threads.py tests/test_gpu_emulated_ranks.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests: exit $?"; tail -3 $OUT/tests.log | cut -c1-300
LAB=$PWD/spiking-neural-networks_amd/csrc/lab
for case in "16 1" "16 0" "22 1"; do set -- $case; echo "--- side $1 chem $2"; SNN_AMD_LIB=$LAB/libsnn_lab_timing.so python3 profiles/trace_small_step.py $1 $2 0 1200 2>&1 | grep -E "k_step_resident_q" | head -2; done
for case in "16 0" "16 1" "22 0" "22 1"; do
  set -- $case
  rm -rf $OUT/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 profiles/trace_small_step.py $1 $2 0 3000 > /dev/null 2> $OUT/trace.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/small_step_side${1}_chem${2}_quarters1_kernel_stats.csv 2>/dev/null
  echo "side $1 chem $2: $(grep k_step_resident $f | sed 's/.*",//' | cut -d, -f1-3)"
done
rm -rf $OUT/prof
python3 profiles/measure_small_plastic.py 3000 > $OUT/small_plastic_lattices.jsonl 2> /dev/null
python3 profiles/measure_small_chem.py 3000 2> /dev/null | grep lattice > $OUT/small_chemical_lattices.jsonl
fails=0
for i in $(seq 1 ${PEER_REPEATS:-20}); do
  GPU_MAX_HW_QUEUES=24 SNN_EMULATED_RANKS_CHILD=1 timeout 300 python3 -m pytest tests/test_gpu_halo_peer.py -m gpu -q -p no:cacheprovider > $OUT/peer_$i.log 2>&1 || fails=$((fails+1))
  echo "peer_$i: $(tail -1 $OUT/peer_$i.log)" >> $OUT/peer_form_repeats.txt
done
echo "peer-form tests (ranks as threads), $fails of ${PEER_REPEATS:-20} runs failed"; grep -h "^FAILED" $OUT/peer_*.log | sort | uniq -c
cat $OUT/peer_form_repeats.txt | cut -c1-120

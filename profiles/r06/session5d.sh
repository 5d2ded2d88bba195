set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_s5d
for i in 1 2 3 4; do
  SNN_EMULATED_RANKS_CHILD=1 timeout 300 python3 -m pytest tests/test_gpu_halo_peer.py -m "gpu" -q -x -k "between_other_steps" -p no:cacheprovider > gpurun_out/r06_s5d/run$i.log 2>&1; echo "run $i image on: exit $?"; tail -2 gpurun_out/r06_s5d/run$i.log | cut -c1-200
done
for i in 1 2; do
  SNN_AMD_CSR_IMAGE=0 SNN_EMULATED_RANKS_CHILD=1 timeout 300 python3 -m pytest tests/test_gpu_halo_peer.py -m "gpu" -q -x -k "between_other_steps" -p no:cacheprovider > gpurun_out/r06_s5d/run_noimage$i.log 2>&1; echo "run $i image off: exit $?"; tail -2 gpurun_out/r06_s5d/run_noimage$i.log | cut -c1-200
done
SNN_EMULATED_RANKS_CHILD=1 timeout 600 python3 -m pytest tests/test_gpu_halo_peer.py -m "gpu" -q -p no:cacheprovider > gpurun_out/r06_s5d/all_peer.log 2>&1; echo "all peer tests: exit $?"; tail -3 gpurun_out/r06_s5d/all_peer.log | cut -c1-200

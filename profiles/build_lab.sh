#!/bin/bash
# The instrumented build the phase clocks come from (k_step_resident_q prints s_memtime stamps of workgroup 0 every 500 steps):
#   bash profiles/build_lab.sh      -> spiking-neural-networks_amd/csrc/lab/libsnn_lab_timing.so (+ .hash = the sources' hash)
# used as  SNN_AMD_LIB=.../lab/libsnn_lab_timing.so python3 profiles/trace_small_step.py <side> <chem> 0 1200
# (SNN_LAB_BUILD: the models of the lab selection only -- two minutes instead of six).  The directory is listed in .gpurunignore.
set -eu
cd "$(dirname "$0")/.."
mkdir -p spiking-neural-networks_amd/csrc/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-result -Wno-pass-failed \
    -DSNN_LAB_BUILD -DSNN_LAB_TIMING -o spiking-neural-networks_amd/csrc/lab/libsnn_lab_timing.so spiking-neural-networks_amd/csrc/snn_network.hip
python3 - <<'PY'
from snn_amd import _lib
open("spiking-neural-networks_amd/csrc/lab/libsnn_lab_timing.so.hash", "w").write(_lib.source_hash() + "\n")
PY

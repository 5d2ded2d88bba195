#!/bin/bash
# Round-4 extras on a GPU box:  bash profiles/collect_extra.sh  (writes gpurun_out/r04x/ and gpurun_out/campaign_d/).
#  1. campaign D: the small dense class of the round-3 mismatches with AMD_SERIALIZE_KERNEL=3 (launch ordering taken out)
#  2. the default c3 line from four fresh processes (placement selection: how far apart do they land)
#  3. c4 with 1 % of the neurons spiking per step: the bench line and FETCH_SIZE / WRITE_SIZE of the STDP scatter kernels
set -u
OUT=$PWD/gpurun_out/r04x
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
DATE=$(date -u +%Y-%m-%d)
AMD_SERIALIZE_KERNEL=3 python3 tests/campaign.py --minutes ${CAMPAIGN_MINUTES:-14} --workers 8 --streamers 2 --filter small_dense \
    --first-seed 900000 --tests test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection \
    --out gpurun_out/campaign_d 2>&1 | tail -3
for i in 1 2 3 4; do python3 bench.py --config c3 --no-cpu-baseline > "$OUT/c3_fresh_process_$i.json" 2> /dev/null; done
python3 bench.py --config c4 --spike-fraction 0.01 --steps 50 --warmup 100 --repeats 2 --no-cpu-baseline \
    > "$OUT/c4_spiking_1pct_bench.json" 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/pmc_$c"
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 bench.py --config c4 --spike-fraction 0.01 --steps 10 \
        --warmup 100 --repeats 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> "$OUT/pmc_$c.err"
done
python3 profiles/summarize_pmc.py "$(find "$OUT/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)" "$OUT/c4_spiking_1pct_pmc_traffic.json" k_stdp_columns "$DATE" | tail -12
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls "$OUT"

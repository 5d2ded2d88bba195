#!/bin/bash
# round 6, GPU session 5: two direct questions about the stray host writes.
#  (a) tests/cpp/late_copy_probe.cpp: does a pageable copy on a non-blocking stream complete when hipStreamSynchronize returns?
#      alone, then 12 processes x 4 threads on the 16 cores the container schedules beside three streaming bench processes
#      (campaign E's oversubscription) -- millions of copies, every one checked for an early return, a late write, a late read;
#  (b) the trap with the LIBRARY'S OWN host tables in the arena (snn_debug_set_host_allocator): the temporaries the getters
#      download into become inaccessible the moment they are freed.  --lean workers, SNN_AMD_PINNED_COPIES=0, campaign E's load.
set -u
OUT=$PWD/gpurun_out/r06_s5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_halo_direct.py -m gpu -q > $OUT/test_gpu_halo_direct.log 2>&1
echo "halo direct (step image on shard handles): exit $?"; tail -2 $OUT/test_gpu_halo_direct.log | cut -c1-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/late_copy_probe tests/cpp/late_copy_probe.cpp -pthread 2> $OUT/probe_build.err
/tmp/late_copy_probe 60 8 > $OUT/probe_alone.json 2> $OUT/probe_alone.err; echo "probe alone: exit $?"; cat $OUT/probe_alone.json; head -5 $OUT/probe_alone.err
for i in 1 2 3; do python3 bench.py --config c2 --steps 4000 --warmup 5 --repeats 1 --no-cpu-baseline > /dev/null 2>&1 & done
sleep 20
pids=""
for i in $(seq 1 12); do /tmp/late_copy_probe 300 4 > $OUT/probe_load_$i.json 2> $OUT/probe_load_$i.err & pids="$pids $!"; done
rc=0; for p in $pids; do wait $p || rc=1; done
echo "probe under load: any event = $rc"; cat $OUT/probe_load_*.json | python3 -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print({k: sum(r[k] for r in rows) for k in ('iterations','early_returns','late_writes','late_reads')}, len(rows), 'processes')"
cat $OUT/probe_load_*.err | head -12
wait
timeout 900 python3 -m pytest tests/test_gpu_guard_arena.py -m gpu -q -x > $OUT/trap_with_internal_tables.log 2>&1
echo "trap with the library's tables in the arena: exit $?"; tail -3 $OUT/trap_with_internal_tables.log | cut -c1-300
TESTS=test_gpu_randomized:test_random_network,test_gpu_persistent_run:test_random_fault_injection,test_gpu_persistent_run:test_random_electrical_networks,test_gpu_sequences:test_random_call_sequence,test_gpu_reward_network:test_connections_between_lattices
SNN_AMD_PINNED_COPIES=0 timeout 2500 python3 tests/campaign.py --lean --minutes ${CAMPAIGN_MINUTES:-32} --workers 24 --streamers 3 --first-seed 50000000 \
    --out $OUT/lean_tables_pinned0 --tests $TESTS > $OUT/lean_tables_pinned0.log 2>&1
rm -rf $OUT/lean_tables_pinned0/repro/*/checkpoint* 2>/dev/null
python3 -c "
import json
d=json.load(open('$OUT/lean_tables_pinned0/summary.json'))
print('lean + library tables in the arena, pinned_copies 0', {k:d.get(k) for k in ('wall_s','executions','failures','trap_faults','trap_calls','trap_buffers_retired','trap_library_host_tables')}, 'trap reports', len(d.get('trap_reports', [])))
for r in d['failure_records'][:6]: print('  FAIL', r['test'], r['seed'], r['message'][:400].replace(chr(10),' | '))
for r in d.get('trap_fault_records', [])[:6]: print('  TRAP FAULT', r)
for r in d.get('trap_reports', [])[:6]: print('  TRAP REPORT', r)"
for f in $OUT/lean_tables_pinned0/guard-*.log; do [ -s "$f" ] && { echo "--- $f"; head -80 "$f" | cut -c1-220; }; done 2>/dev/null | head -200
du -sh $OUT

mkdir -p gpurun_out/upd; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_network.py tests/test_gpu_fused_step.py tests/test_gpu_izhikevich_electrical.py -x -q -m gpu 2>&1 | tail -2
for c in c3 c2; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/upd/p$c -- python3 bench.py --config $c --no-cpu-baseline --steps 100 --repeats 2 > gpurun_out/upd/b$c.json 2>/dev/null
f=$(find gpurun_out/upd/p$c -name "*kernel_stats.csv" | head -1); grep "k_update\|k_inputs_dense" $f | cut -c1-120; tail -1 gpurun_out/upd/b$c.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; rm -rf gpurun_out/upd/p$c
done

set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_sharding.py -m gpu -x -q --timeout 300 --timeout-method thread -k "shard_route" > gpurun_out/r3_t2a.log 2>&1 || { tail -40 gpurun_out/r3_t2a.log; exit 1; }
tail -3 gpurun_out/r3_t2a.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharding.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "not shard_route and (sharded or world1 or two_ranks or c4_size or segment_plan or planned_fused or c4_full)" > gpurun_out/r3_t2b.log 2>&1 || { tail -40 gpurun_out/r3_t2b.log; exit 1; }
tail -3 gpurun_out/r3_t2b.log
timeout -k 10 200 python tools/route_probe.py
timeout -k 10 300 python bench.py --sharded --steps 200 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_sh.json 2> gpurun_out/r3_bench_sh.err || { tail -20 gpurun_out/r3_bench_sh.err; exit 1; }
cat gpurun_out/r3_bench_sh.json | python -c "import json,sys; d=json.load(sys.stdin); print('sharded w1', d['ms_per_step'], d['value'])"

# sharded GPU tests + the world-1 sharded bench (through a size-1 RCCL group) + its host profile (dev tool)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_sharding.py -x -q --timeout 200 --timeout-method thread > gpurun_out/shard_tests.log 2>&1 || { tail -n 30 gpurun_out/shard_tests.log; exit 1; }
tail -n 2 gpurun_out/shard_tests.log
timeout -k 10 200 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/bench_sharded_w1.json 2> gpurun_out/bench_sharded_w1.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_sharded_w1.json
timeout -k 10 200 python tools/sharded_hostprof.py > gpurun_out/hostprof.txt 2>&1
grep "function calls" gpurun_out/hostprof.txt

set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharding.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "shard or world1 or two_ranks or c4_size" > gpurun_out/r3_t4.log 2>&1 || { tail -40 gpurun_out/r3_t4.log; exit 1; }
tail -3 gpurun_out/r3_t4.log
timeout -k 10 200 python tools/sharded_hostprof.py 2>&1 | grep -v "^$" | head -16
timeout -k 10 300 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2> gpurun_out/r3_bench_sh.err | tail -1 > gpurun_out/r3_bench_sh.json
python -c "import json; d=json.load(open('gpurun_out/r3_bench_sh.json')); print('sharded w1', d['ms_per_step'], d['value'])"

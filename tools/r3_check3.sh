set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2> gpurun_out/r3_bench_sh.err | tail -1 > gpurun_out/r3_bench_sh.json
python -c "import json; d=json.load(open('gpurun_out/r3_bench_sh.json')); print('sharded w1', d['ms_per_step'], d['value'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sh -o s -- python $R/bench.py --sharded --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_sh.log 2>&1
python $R/tools/kstats.py $R/gpurun_out/prof_sh 22 110

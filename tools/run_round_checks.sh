set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_b.log 2>&1 || { tail -30 gpurun_out/gputests_b.log; exit 1; }
tail -3 gpurun_out/gputests_b.log
timeout -k 10 400 python bench.py --steps 200 --warmup 10 > gpurun_out/bench_r2_b.json 2> gpurun_out/bench_r2_b.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o b -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/step_loop.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/step_loop.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r02_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
cp profiles/r02_pmc_traffic.json gpurun_out/
echo done

# Round-end evidence, one gpurun call: GPU tests, the bench line, rocprofv3 kernel stats of the same command, the two PMC
# passes behind roofline.traffic, the sharded world-1 line and its kernel stats.  Outputs under gpurun_out/; the
# summaries to keep are copied into profiles/r03_* afterwards.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_r3.log 2>&1 || { tail -30 gpurun_out/gputests_r3.log; exit 1; }
tail -3 gpurun_out/gputests_r3.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/step_loop.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/step_loop.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r03_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
cp profiles/r03_pmc_traffic.json gpurun_out/
(bash tools/pmc_score.sh) && python tools/pmc_score_summary.py gpurun_out/pmc_score1 gpurun_out/pmc_score2 r03_pmc_score_kernel.json > gpurun_out/pmc_score_summary.log 2>&1
cp profiles/r03_pmc_score_kernel.json gpurun_out/
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3.json 2> gpurun_out/bench_r3.err || { tail -20 gpurun_out/bench_r3.err; exit 1; }
timeout -k 10 300 python bench.py --sharded --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_r3_sharded.err | tail -1 > gpurun_out/bench_r3_sharded.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sharded -o s -- python $R/bench.py --sharded --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_sharded.log 2>&1
cd $R
timeout -k 10 600 python tools/bench_configs.py gpurun_out/r03_configs.json > /dev/null 2> gpurun_out/configs_r3.err || { tail -5 gpurun_out/configs_r3.err; exit 1; }
for m in c4 c4r c4m c5 c5r c5c c5x; do timeout -k 10 120 python tools/eval_probe.py $m 3 2>/dev/null; done > gpurun_out/eval_probe_r3.txt
cat gpurun_out/eval_probe_r3.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r3.json'))
print('fused', d['ms_per_step'], d['ms_per_step_min_max'], d['value'], 'roofline', d['roofline']['frac'], 'step', d['step_roofline']['frac'])
print('scatter', d['scatter_add']['frac_hbm_peak_planned'], d['scatter_add']['frac_hbm_peak_with_plan'], 'eval', d['eval']['frac_mfma_f32_peak'])
s = json.load(open('gpurun_out/bench_r3_sharded.json'))
print('sharded w1', s['ms_per_step'], s['ms_per_step_min_max'], s['value'])
PY
echo done

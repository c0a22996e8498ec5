set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2> gpurun_out/r3_bench_sh.err | tail -1 > gpurun_out/r3_bench_sh.json
python -c "import json; d=json.load(open('gpurun_out/r3_bench_sh.json')); print('sharded w1', d['ms_per_step'], d['value'])"
timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench5.json'))
print('fused', d['ms_per_step'], d['value'])
for k,v in d['kernels'].items(): print('  ',k, round(v['avg_ms']*1e3,1), v.get('calls_per_step'))
print(d['step_roofline']['frac'], d['roofline']['frac'], d['roofline']['frac_min_median_max'])
PY

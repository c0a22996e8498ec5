set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "segment_plan or scatter_add or sparse_row or bpr_step or planned_fused or small_plan or c4_full or sharded_step" > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err || { tail -20 gpurun_out/r3_bench1.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench1.json'))
print(d['ms_per_step'], d['value'])
for k,v in d['kernels'].items(): print(k, round(v['avg_ms']*1e3,1), v.get('calls_per_step'))
print(d['step_roofline']['frac'], d['roofline']['frac'], d['scatter_add'])
PY

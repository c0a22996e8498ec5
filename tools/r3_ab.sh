set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in default v512; do
  if [ $v = v512 ]; then export DEEPREC_AMD_LIB=$R/abl/libdr_v512.so; fi
  timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/r3_ab_$v.json 2> gpurun_out/r3_ab_$v.err || { tail -20 gpurun_out/r3_ab_$v.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3_ab_%s.json'%sys.argv[1]))
print(sys.argv[1], d['ms_per_step'], d['value'])
for k,v in d['kernels'].items(): print('  ',k, round(v['avg_ms']*1e3,1), v.get('calls_per_step'))
PY
  timeout -k 10 100 python tools/plan_probe2.py
done

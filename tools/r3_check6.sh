cd $GRAFT_REPO_ROOT
for st in 20 200; do
timeout -k 10 300 python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r3_b6.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused', d['steps'], d['ms_per_step'], d['ms_per_step_min_max'], d['streams'])" || tail -5 gpurun_out/r3_b6.err
timeout -k 10 300 python bench.py --sharded --steps $st --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r3_b6.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sharded', d['steps'], d['ms_per_step'], d['ms_per_step_min_max'], d['streams'])" || tail -5 gpurun_out/r3_b6.err
done

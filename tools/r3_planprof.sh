set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python $R/tools/plan_probe2.py
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_plan -o p -- python $R/tools/plan_probe2.py > $R/gpurun_out/prof_plan.log 2>&1
python $R/tools/kstats.py $R/gpurun_out/prof_plan 2>/dev/null | head -30 || true

# the world-1 sharded bench three times on one box (run-to-run spread; dev tool)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
timeout -k 10 200 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
timeout -k 10 200 python bench.py --sharded --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'

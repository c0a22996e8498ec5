# kernel trace of a few world-1 sharded steps through RCCL: the launch sequence of one step (dev tool)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shard_trace -o s -- python $R/bench.py --sharded --steps 200 --warmup 5 --no-cpu-baseline > $R/gpurun_out/shard_trace.log 2>&1
ls $R/gpurun_out/shard_trace

#!/bin/bash
# Two rocprofv3 --pmc passes (8 SQ counters each) over tools/eval_loop.py; CSVs land in gpurun_out/pmc_score{1,2}/ (dev tool).
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_score1 -o p -- python3 $R/tools/eval_loop.py > $R/gpurun_out/pmc_score1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_score2 -o p -- python3 $R/tools/eval_loop.py > $R/gpurun_out/pmc_score2.log 2>&1 || exit 1

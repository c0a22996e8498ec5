#!/bin/bash
# rocprofv3 --pmc passes over tools/hconv_probe.py (dev tool); CSVs land in gpurun_out/pmc_hconv{1,2}[_abl]/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_hconv1$1 -o p -- python3 $R/tools/hconv_probe.py > $R/gpurun_out/pmc_hconv1$1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc_hconv2$1 -o p -- python3 $R/tools/hconv_probe.py > $R/gpurun_out/pmc_hconv2$1.log 2>&1 || exit 1

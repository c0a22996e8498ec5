#!/bin/bash
# GRBM_GUI_ACTIVE (GPU clock cycles while busy) per kernel: cycles / duration = the clock the kernel actually ran at (dev tool)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_hconv -o p -- python3 $R/tools/hconv_probe.py > $R/gpurun_out/clk_hconv.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_mfma -o p -- $R/abl/mfma_probe > $R/gpurun_out/clk_mfma.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_eval -o p -- python3 $R/tools/eval_probe.py c4 > $R/gpurun_out/clk_eval.log 2>&1 || exit 1

#!/bin/bash
# Timing ablations of the scoring kernel (dev tool): builds abl/libdeeprec_amd_abl<N>.so with -DDR_ABL=<N>
# (1: no top-K / tie tests, 2: MFMAs replaced by a few FMAs, 3: no staging of item tiles, 4: no barriers -- results
# are wrong by construction, only the time is of interest).  tools/eval_probe.py loads one through DR_PROBE_LIB.
set -e
cd "$(dirname "$0")/../deeprec_amd/csrc"
mkdir -p ../../abl build
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -DDR_ABL=$n -c score_topk.hip -o build/score_topk_abl$n.o
  objs=$(ls build/*.o | grep -v score_topk)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/score_topk_abl$n.o -o ../../abl/libdeeprec_amd_abl$n.so
done

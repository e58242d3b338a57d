#!/bin/bash
# PMC comparison of two library builds on one GPU: VALU instructions and active cycles of K2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  RSLF_LIBRARY=$R/$L rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_ab/$N -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_ab/$N.json 2> $R/gpurun_out/pmc_ab/$N.err
done

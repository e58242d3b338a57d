#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: writes rocprofv3 outputs under gpurun_out/$PROF_DIR/ (default prof).
# Passes are separate on purpose: --stats alone, then one --pmc pass per counter group
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   BENCH_ARGS="--config c5 --rows 16" PROF_DIR=prof_c5 bash tools/collect_profiles.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/bench_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/bench_sq2.err || true
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py $ARGS > $OUT/bench_l2.json 2> $OUT/bench_l2.err || true
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- python3 $R/bench.py $ARGS > $OUT/bench_ta.json 2> $OUT/bench_ta.err || true
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $OUT/pmc_tcp -- python3 $R/bench.py $ARGS > $OUT/bench_tcp.json 2> $OUT/bench_tcp.err || true
echo profiles collected in $OUT

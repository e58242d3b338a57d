#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: writes rocprofv3 outputs under gpurun_out/$PROF_DIR/ (default prof).
# Passes are separate on purpose: --stats alone, then one --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE cannot
# share a pass; MI355X_MICROARCH.md "rocprofv3 PMC slots"); TA / TD / TCP counters two per pass (larger groups aborted or
# hung the profiler on this pool).  Every pass has its own timeout and the chain stops at the first failure.
#   BENCH_ARGS="--config c5 --rows 16" PROF_DIR=prof_c5 bash tools/collect_profiles.sh
# Round 3: the SAME lease also records (a) an un-profiled bench line of the same command (bench_plain.json: what the
# roofline of record is checked against), (b) the source hash the numbers belong to, (c) the library's resource table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${PROF_STEPS:-3} --warmup 1 --no-cpu-baseline --no-e2e ${BENCH_ARGS:-}"
python3 -c "import sys; sys.path.insert(0, '$R'); from remotesensingproject_amd import build as b; print(b.source_hash())" > $OUT/source_hash.txt 2>/dev/null
pass() {  # name, rocprofv3 options...
  name=$1; shift
  echo "pass $name" >> $OUT/progress.txt
  timeout -k 5 ${PASS_TIMEOUT:-200} rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  rc=$?
  echo "pass $name rc=$rc" >> $OUT/progress.txt
  return $rc
}
# PASSES: which of the passes to run (default: all).  The full-size c5 run takes minutes per pass: PASSES="stats pmc_fetch pmc_write pmc_sq"
PASSES=${PASSES:-"stats pmc_fetch pmc_write pmc_sq pmc_sq2 pmc_l2 pmc_ta pmc_td pmc_tcp"}
want() { case " $PASSES " in *" $1 "*) return 0;; esac; return 1; }
timeout -k 5 ${PASS_TIMEOUT:-200} python3 $R/bench.py --steps ${PLAIN_STEPS:-10} --warmup 2 --no-cpu-baseline --no-e2e ${BENCH_ARGS:-} > $OUT/bench_plain.json 2> $OUT/bench_plain.err || { echo "plain bench failed"; tail -3 $OUT/bench_plain.err; exit 1; }
run() { want $1 || return 0; pass "$@" || { echo "pass $1 failed"; tail -3 $OUT/bench_$1.err; exit 1; }; }
run stats --stats
[ -f $OUT/bench_stats.json ] && cp $OUT/bench_stats.json $OUT/bench_stats.json.keep
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY
run pmc_sq2 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum
run pmc_ta --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run pmc_td --pmc TD_TD_BUSY_sum TD_TC_STALL_sum
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
echo "profiles in $OUT: $(tail -1 $OUT/progress.txt)"

#!/bin/bash
# rocprofv3 passes over a short bench run (separate runs for the trace and for each PMC set).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PROF_OUT:-prof}
rm -rf $OUT && mkdir -p $OUT
# default: bench.py's own defaults (the command the driver runs), minus the CPU baseline
ARGS="$R/bench.py --no-cpu-baseline ${BENCH_ARGS:-}"
export PROF_CMD="python3 bench.py --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || exit 1
if [ -z "$SKIP_MEM" ]; then
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1 || exit 1
fi
if [ -n "$PROF_L1" ]; then
# vector L1 / L2 request counters (hierarchy scenes: where the node and leaf loads are served from); a counter this ROCm does
# not know fails the pass, which is then skipped
timeout -s KILL 170 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc5 -- python3 $ARGS > $OUT/pmc5.log 2>&1 || echo "pmc5 (TCP counters) skipped"
timeout -s KILL 170 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc6 -- python3 $ARGS > $OUT/pmc6.log 2>&1 || echo "pmc6 (TCC counters) skipped"
fi
tail -1 $OUT/trace.log
python3 $R/tools/prof_summary.py $OUT

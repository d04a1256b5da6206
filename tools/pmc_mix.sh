#!/bin/bash
# tools/pmc_mix.sh [LIB]: dynamic VALU instruction mix of k_pass over a short default bench run (two PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mix
rm -rf $OUT && mkdir -p $OUT
[ -n "$1" ] && export RUSTRACER_AMD_SO=$R/tools/dbg/$1.so
ARGS="$R/bench.py --steps ${STEPS:-48} --warmup 12 --no-cpu-baseline --frames-per-launch ${FRAMES:-16} ${BENCH_ARGS:-}"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $OUT

#!/bin/bash
# tools/final_measure.sh: the round's measurement set -- rocprofv3 summaries (headline launch, C4, C5 at 1080p) and bench lines.
# Every GPU step runs under its own time limit (tools/prof.sh) and the script stops at the first failure.
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out/final
export PROF_FRAMES=138 PROF_RAYS_PER_FRAME=44180520 PROF_RAYS_EXECUTED_PER_FRAME=25802424   # 14 single frames + 24 warm-up + 5 x 20 timed
BENCH_ARGS="--steps 20" PROF_OUT=prof_c3 bash tools/prof.sh > gpurun_out/final/rocprof_steps20.txt 2>&1 || { echo "prof c3 failed"; tail -5 gpurun_out/final/rocprof_steps20.txt; exit 1; }
echo "prof c3 done"
unset PROF_FRAMES PROF_RAYS_PER_FRAME
PROF_L1=1 BENCH_ARGS="--config c4 --steps 2 --warmup 1" PROF_OUT=prof_c4 bash tools/prof.sh > gpurun_out/final/rocprof_c4.txt 2>&1 || { echo "prof c4 failed"; tail -5 gpurun_out/final/rocprof_c4.txt; exit 1; }
echo "prof c4 done"
PROF_L1=1 BENCH_ARGS="--config c5 --width 1920 --height 1080 --aa 4 --steps 2 --warmup 1" PROF_OUT=prof_c5 bash tools/prof.sh > gpurun_out/final/rocprof_c5_1080p.txt 2>&1 || { echo "prof c5 failed"; tail -5 gpurun_out/final/rocprof_c5_1080p.txt; exit 1; }
echo "prof c5 done"
cd $R
timeout -k 10 300 python bench.py --steps 20 > gpurun_out/final/bench_n1_steps20.json 2> gpurun_out/final/bench_n1_steps20.err || { echo "bench steps20 failed"; exit 1; }
echo "bench steps 20 done"
timeout -k 10 300 python bench.py > gpurun_out/final/bench_n1_default.json 2> gpurun_out/final/bench_n1_default.err || { echo "bench default failed"; exit 1; }
echo "bench default done"
timeout -k 10 200 python bench.py --config c4 --steps 3 --warmup 1 > gpurun_out/final/bench_c4_n1.json 2> gpurun_out/final/bench_c4_n1.err || { echo "bench c4 failed"; exit 1; }
timeout -k 10 200 python bench.py --config c5 --width 1920 --height 1080 --aa 4 --steps 2 --warmup 1 > gpurun_out/final/bench_c5_1080p_n1.json 2> gpurun_out/final/bench_c5_1080p_n1.err || { echo "bench c5 failed"; exit 1; }
echo "bench c4 c5 done"
RUSTRACER_NO_BVH=1 timeout -k 10 300 python bench.py --config c4 --steps 1 --warmup 0 > gpurun_out/final/bench_c4_linear_scan_n1.json 2> gpurun_out/final/bench_c4_linear_scan_n1.err || { echo "bench c4 linear failed"; exit 1; }
echo "bench c4 linear scan done"

#!/bin/bash
# tools/occ_probe.sh: throughput against resident waves per SIMD (variants tools/dbg/wN.so built with -DRT_WAVES_PER_EU=N,
# recursion stack in HBM so that LDS does not cap the occupancy); bench at the driver's step count and at 240
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {  # label, env...
  local label=$1; shift
  for steps in 20 240; do
    env "$@" timeout -k 10 150 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-call-timing > gpurun_out/occ.json 2> gpurun_out/occ.err || { echo "$label FAILED"; tail -3 gpurun_out/occ.err; exit 1; }
    python - "$label" $steps <<'PY'
import json,sys
d=json.load(open("gpurun_out/occ.json"))
print(f"{sys.argv[1]:44s} steps {sys.argv[2]:>3s} ms/frame {d['ms_per_step']:.4f} regions {[round(x,2) for x in d['config']['region_ms']]}", flush=True)
PY
  done
}
run "w4 lds stack (default)" RUSTRACER_AMD_SO=tools/dbg/w4.so
run "w4 hbm stack, 4 waves" RUSTRACER_AMD_SO=tools/dbg/w4.so RUSTRACER_GLOBAL_STACK=1
run "w5 hbm stack, 5 waves" RUSTRACER_AMD_SO=tools/dbg/w5.so RUSTRACER_GLOBAL_STACK=1 RUSTRACER_BLOCKS_PER_CU=5
run "w6 hbm stack, 6 waves" RUSTRACER_AMD_SO=tools/dbg/w6.so RUSTRACER_GLOBAL_STACK=1 RUSTRACER_BLOCKS_PER_CU=6
run "w8 hbm stack, 8 waves" RUSTRACER_AMD_SO=tools/dbg/w8.so RUSTRACER_GLOBAL_STACK=1 RUSTRACER_BLOCKS_PER_CU=8
run "w5 lds stack, 4 waves (spill cost)" RUSTRACER_AMD_SO=tools/dbg/w5.so

#!/bin/bash
# tools/occ_probe2.sh: the plain throughput body (RUSTRACER_FAST_VARIANT=1) at 4 and 5 waves per SIMD (tools/dbg/w4.so, w5.so)
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
run "w4 v1 lds stack (default)" RUSTRACER_AMD_SO=tools/dbg/w4.so RUSTRACER_FAST_VARIANT=1
run "w4 v1 hbm stack, 4 waves" RUSTRACER_AMD_SO=tools/dbg/w4.so RUSTRACER_GLOBAL_STACK=1 RUSTRACER_FAST_VARIANT=1
run "w5 v1 hbm stack, 5 waves" RUSTRACER_AMD_SO=tools/dbg/w5.so RUSTRACER_GLOBAL_STACK=1 RUSTRACER_BLOCKS_PER_CU=5 RUSTRACER_FAST_VARIANT=1
run "w5 v1 lds stack, 4 waves (spill cost)" RUSTRACER_AMD_SO=tools/dbg/w5.so RUSTRACER_FAST_VARIANT=1

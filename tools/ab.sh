#!/bin/bash
# tools/ab.sh NAME... : bench each tools/dbg/NAME.so (short run, no CPU baseline); prints ms/frame per variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for n in "$@"; do
  RUSTRACER_AMD_SO=$PWD/tools/dbg/$n.so timeout -k 10 150 python bench.py --steps ${STEPS:-120} --warmup 24 --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]; d=json.load(open(f"gpurun_out/ab_{n}.json"))
print(f"{n:12s} ms/frame {d['ms_per_step']:.4f}  launch_ms {d['roofline']['launch_ms']:.3f}  TFLOP/s {d['roofline']['achieved']:.2f}")
PY
done

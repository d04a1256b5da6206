#!/bin/bash
# tools/sweep_env.sh "VAR=val ..." ...: default-library bench at the driver's step count under extra environment
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for spec in "$@"; do
  i=$((i+1))
  env $spec timeout -k 10 150 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-call-timing ${BENCH_ARGS:-} > gpurun_out/sweep_$i.json 2> gpurun_out/sweep_$i.err || { echo "$spec FAILED"; tail -3 gpurun_out/sweep_$i.err; exit 1; }
  python - "$spec" $i <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/sweep_{sys.argv[2]}.json"))
print(f"{sys.argv[1]:60s} ms/frame {d['ms_per_step']:.4f}  TFLOP/s {d['roofline']['achieved']:.2f} regions {[round(x,2) for x in d['config']['region_ms']]}")
PY
done

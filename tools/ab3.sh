#!/bin/bash
# tools/ab3.sh "ENV=... LIB" ...: bench variants with extra environment; each arg: "<libname> [VAR=val ...]"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for spec in "$@"; do
  i=$((i+1))
  set -- $spec
  n=$1; shift
  env "$@" RUSTRACER_AMD_SO=$PWD/tools/dbg/$n.so timeout -k 10 150 python bench.py --steps ${STEPS:-192} --warmup 24 --no-cpu-baseline --no-call-timing ${BENCH_ARGS:-} > gpurun_out/ab3_$i.json 2> gpurun_out/ab3_$i.err || { echo "$spec FAILED"; tail -3 gpurun_out/ab3_$i.err; exit 1; }
  python - "$spec" $i <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab3_{sys.argv[2]}.json"))
print(f"{sys.argv[1]:50s} ms/frame {d['ms_per_step']:.4f}  launch_ms {d['roofline']['launch_ms']:.3f}  TFLOP/s {d['roofline']['achieved']:.2f}")
PY
done

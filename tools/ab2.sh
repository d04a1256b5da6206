#!/bin/bash
# tools/ab2.sh LIBNAME "bench args" ... : bench one variant under several bench.py argument sets
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
n=$1; shift
i=0
for a in "$@"; do
  i=$((i+1))
  RUSTRACER_AMD_SO=$PWD/tools/dbg/$n.so timeout -k 10 150 python bench.py --warmup 24 --no-cpu-baseline $a > gpurun_out/ab2_$i.json 2> gpurun_out/ab2_$i.err || { echo "$a FAILED"; tail -3 gpurun_out/ab2_$i.err; exit 1; }
  python - "$a" $i <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab2_{sys.argv[2]}.json"))
print(f"{sys.argv[1]:50s} ms/frame {d['ms_per_step']:.4f}  launch_ms {d['roofline']['launch_ms']:.3f}  TFLOP/s {d['roofline']['achieved']:.2f}")
PY
done

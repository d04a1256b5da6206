#!/bin/bash
# tools/ab_args.sh "bench args" ... : bench the in-tree library under several bench.py argument sets
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 150 python bench.py --warmup 32 --no-cpu-baseline $a > gpurun_out/aba_$i.json 2> gpurun_out/aba_$i.err || { echo "$a FAILED"; tail -3 gpurun_out/aba_$i.err; exit 1; }
  python - "$a" $i <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/aba_{sys.argv[2]}.json"))
print(f"{sys.argv[1]:50s} ms/frame {d['ms_per_step']:.4f}  launch_ms {d['roofline']['launch_ms']:.3f}  TFLOP/s {d['roofline']['achieved']:.2f}")
PY
done

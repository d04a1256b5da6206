#!/bin/bash
# bench for frames-in-flight x tail-team configurations on the GPU box.
set -e
mkdir -p gpurun_out
for CFG in ${CFGS:-"1:0:1" "1:0:2" "1:0:3" "1:0:4" "1:8:1" "1:8:2" "1:8:3" "2:0:3" "4:0:3"}; do
  IFS=: read T TAIL F <<< "$CFG"
  export RUSTRACER_TEAM=$T
  if [ "$TAIL" != "0" ]; then export RUSTRACER_TAIL=$TAIL; else unset RUSTRACER_TAIL; fi
  timeout -k 10 300 python bench.py --steps 60 --warmup 6 --inflight $F --no-cpu-baseline > gpurun_out/bench_if.json 2> gpurun_out/bench_if.err || { tail -20 gpurun_out/bench_if.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bench_if.json')); r=d['roofline']; print('T=$T TAIL=$TAIL F=$F', 'samples/s %.3e'%d['value'], 'ms/frame %.3f'%d['ms_per_step'], 'eff_ms %.3f'%r['kernel_ms'], 'latency_ms %.3f'%r['launch_latency_ms'], 'TFLOP/s %.2f'%r['achieved'])"
done

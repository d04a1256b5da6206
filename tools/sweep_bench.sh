#!/bin/bash
# bench.py under a list of argument/env settings: each line is "ARGS|VAR=val VAR=val ..."
mkdir -p gpurun_out
while IFS= read -r CFG; do
  [ -z "$CFG" ] && continue
  ARGS="${CFG%%|*}"; ENVS="${CFG#*|}"
  env $ENVS timeout -k 10 300 python bench.py --steps ${STEPS:-120} --warmup 12 --no-cpu-baseline $ARGS > gpurun_out/bench_env.json 2> gpurun_out/bench_env.err || { tail -20 gpurun_out/bench_env.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bench_env.json')); r=d['roofline']; print('$ARGS $ENVS', '| samples/s %.3e'%d['value'], 'ms/frame %.3f'%d['ms_per_step'], 'launch_ms %.3f'%r['launch_ms'], 'latency_ms %.3f'%r['launch_latency_ms'], 'TFLOP/s %.2f'%r['achieved'])"
done

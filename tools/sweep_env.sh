#!/bin/bash
# bench under a list of environment settings: each CFG is "F|VAR=val VAR=val ..."
mkdir -p gpurun_out
while IFS= read -r CFG; do
  [ -z "$CFG" ] && continue
  F="${CFG%%|*}"; ENVS="${CFG#*|}"
  env $ENVS timeout -k 10 300 python bench.py --steps ${STEPS:-60} --warmup 6 --inflight $F --no-cpu-baseline > gpurun_out/bench_env.json 2> gpurun_out/bench_env.err || { tail -20 gpurun_out/bench_env.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bench_env.json')); r=d['roofline']; print('F=$F $ENVS', '| samples/s %.3e'%d['value'], 'ms/frame %.3f'%d['ms_per_step'], 'latency_ms %.3f'%r['launch_latency_ms'], 'TFLOP/s %.2f'%r['achieved'])"
done

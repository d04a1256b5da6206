#!/bin/bash
# Parity + bench for each team size (RUSTRACER_TEAM) on the GPU box.
set -e
mkdir -p gpurun_out
for T in ${TEAMS:-1 2 4 8}; do
  export RUSTRACER_TEAM=$T
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "c3 or aa2 or image_pigment or degenerate or partition or solve_batch" > gpurun_out/pytest_T$T.txt 2>&1 || { tail -30 gpurun_out/pytest_T$T.txt; exit 1; }
  tail -1 gpurun_out/pytest_T$T.txt
  for IM in ${IDLES:-default}; do
    if [ "$IM" != "default" ]; then export RUSTRACER_IDLE_MIN=$IM; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_T$T.json 2> gpurun_out/bench_T$T.err
    python -c "
import json; d=json.load(open('gpurun_out/bench_T$T.json')); print('T=$T idle_min=$IM', 'samples/s %.3e'%d['value'], 'ms/frame %.3f'%d['ms_per_step'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'TFLOP/s %.2f'%d['roofline']['achieved'])"
  done
done

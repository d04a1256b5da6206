#!/bin/bash
# Parity + bench for the tail-team configurations (RUSTRACER_TEAM=1, RUSTRACER_TAIL=t) on the GPU box.
set -e
mkdir -p gpurun_out
export RUSTRACER_TEAM=1
for T in ${TAILS:-0 4 8 16}; do
  if [ "$T" != "0" ]; then export RUSTRACER_TAIL=$T; else unset RUSTRACER_TAIL; fi
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_tail$T.txt 2>&1 || { tail -30 gpurun_out/pytest_tail$T.txt; exit 1; }
  tail -1 gpurun_out/pytest_tail$T.txt
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_tail$T.json 2> gpurun_out/bench_tail$T.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_tail$T.json')); print('TAIL=$T', 'samples/s %.3e'%d['value'], 'ms/frame %.3f'%d['ms_per_step'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'TFLOP/s %.2f'%d['roofline']['achieved'])"
done

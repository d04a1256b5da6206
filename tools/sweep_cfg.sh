#!/bin/bash
# tools/sweep_cfg.sh CONFIG "VAR=val ..." ...: bench.py --config CONFIG (one frame, BENCH_ARGS for the size) under extra environment
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cfg=$1; shift
for spec in "$@"; do
  env $spec timeout -k 10 280 python bench.py --config $cfg --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/sweepc.json 2> gpurun_out/sweepc.err || { echo "$spec FAILED"; tail -3 gpurun_out/sweepc.err; exit 1; }
  python - "$cfg $spec" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sweepc.json"))
print(f"{sys.argv[1]:60s} ms/frame {d['ms_per_step']:.2f}  rays/s {d['config']['rays_per_s']:.3e} rays/frame {d['config']['rays_per_frame']:.0f}", flush=True)
PY
done

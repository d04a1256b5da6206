#!/bin/bash
# tools/hier_check.sh: after a change to the hierarchy traversal -- parity (hierarchy tests, fuzz incl. extreme seeds), then C4 / C5 frame times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q --timeout 300 --timeout-method thread \
  -k "hierarchy or random_scenes or flake or c5_random or c4_ or c5_ or mirror_chains or diagonal" > gpurun_out/hc_parity.log 2>&1 || { echo "parity FAILED"; tail -15 gpurun_out/hc_parity.log; exit 1; }
tail -1 gpurun_out/hc_parity.log
timeout -k 10 300 python -u tools/fuzz_parity.py ${FUZZ_FIRST:-1000200} ${FUZZ_N:-150} > gpurun_out/hc_fuzz.log 2>&1 || { echo "fuzz FAILED"; grep -v ": ok\|: \.\." gpurun_out/hc_fuzz.log | tail -5; exit 1; }
tail -1 gpurun_out/hc_fuzz.log
for cfg in "c4 3840 2160 1" "c5 1920 1080 4"; do
  set -- $cfg
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --config $1 --width $2 --height $3 --aa $4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/hc.json 2> gpurun_out/hc.err || { echo "$cfg FAILED"; tail -3 gpurun_out/hc.err; exit 1; }
  python - "$1 $2x$3 aa$4" <<'PY'
import json,sys
d=json.load(open("gpurun_out/hc.json"))
print(f"{sys.argv[1]:30s} ms/frame {d['ms_per_step']:.1f} rays/s {d['config']['rays_per_s']:.3e}", flush=True)
PY
  done
done

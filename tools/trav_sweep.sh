#!/bin/bash
# tools/trav_sweep.sh: C4 (3840x2160) and C5 (1920x1080, 16 spp) frames against RUSTRACER_TRAV_MIN_DONE (0 = every round waits for all rays)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for k in ${KS:-0 8 16 24 32 48}; do
  for cfg in "c4 3840 2160 1" "c5 1920 1080 4"; do
    set -- $cfg
    RUSTRACER_TRAV_MIN_DONE=$k timeout -k 10 200 python bench.py --config $1 --width $2 --height $3 --aa $4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ts.json 2> gpurun_out/ts.err || { echo "$cfg k=$k FAILED"; tail -3 gpurun_out/ts.err; exit 1; }
    python - "$1 $2x$3 aa$4 min_done=$k" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ts.json"))
print(f"{sys.argv[1]:36s} ms/frame {d['ms_per_step']:.1f} rays/s {d['config']['rays_per_s']:.3e}", flush=True)
PY
  done
done

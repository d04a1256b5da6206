#!/bin/bash
# tools/sched_probe.sh: one frame of c4 / c5 at small sizes under both schedules (bench.py --schedule passes | onepass)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in c4 c5; do
  for size in "320 180" "640 360" "1280 720" "1920 1080"; do
    set -- $size
    for sch in passes onepass; do
      timeout -k 10 280 python bench.py --config $cfg --width $1 --height $2 --aa 1 --steps 3 --warmup 1 --schedule $sch --no-cpu-baseline > gpurun_out/sp.json 2> gpurun_out/sp.err || { echo "$cfg $size $sch FAILED"; tail -3 gpurun_out/sp.err; exit 1; }
      python - "$cfg $1x$2 $sch" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sp.json"))
print(f"{sys.argv[1]:32s} ms/frame {d['ms_per_step']:.2f}", flush=True)
PY
    done
  done
done

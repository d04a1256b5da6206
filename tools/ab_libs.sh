#!/bin/bash
# tools/ab_libs.sh LIB_A LIB_B ...: same-box A/B of library builds (tools/dbg/*.so) on C4 (3840x2160) and C5 (1920x1080 x 16 spp), two rounds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
LIBS="$*"
for lib in $LIBS; do
  for cfg in "c4 3840 2160 1" "c5 1920 1080 4"; do
    read -r c w h aa <<< "$cfg"
    RUSTRACER_AMD_SO=$PWD/tools/dbg/$lib.so timeout -k 10 200 python bench.py --config $c --width $w --height $h --aa $aa --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/abl.json 2> gpurun_out/abl.err || { echo "$lib $cfg FAILED"; tail -3 gpurun_out/abl.err; exit 1; }
    python - "$lib $c ${w}x$h aa$aa" <<'PY'
import json,sys
d=json.load(open("gpurun_out/abl.json"))
print(f"{sys.argv[1]:36s} ms/frame {d['ms_per_step']:.1f}", flush=True)
PY
  done
done
done

#!/bin/bash
# tools/leaf_sweep.sh: frame time against the hierarchy's leaf size (RUSTRACER_BVH_LEAF) and, for C5, the in-flight threshold
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {  # label, env..., -- cfg w h aa
  local label=$1; shift
  timeout -k 10 200 env "$@" python bench.py --config $CFG --width $W --height $H --aa $AA --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ls.json 2> gpurun_out/ls.err || { echo "$label FAILED"; tail -3 gpurun_out/ls.err; exit 1; }
  python - "$label" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ls.json"))
print(f"{sys.argv[1]:44s} ms/frame {d['ms_per_step']:.1f}", flush=True)
PY
}
CFG=c4 W=3840 H=2160 AA=1
for l in 2 3 4 6; do run "c4 3840x2160 leaf=$l" RUSTRACER_BVH_LEAF=$l; done
run "c4 3840x2160 leaf=4 octants=1" RUSTRACER_BVH_LEAF=4 RUSTRACER_BVH_OCTANTS=1
CFG=c5 W=1920 H=1080 AA=4
for l in 3 4 6 8; do run "c5 1920x1080x16 leaf=$l" RUSTRACER_BVH_LEAF=$l; done
for k in 32 40 48 56; do run "c5 1920x1080x16 leaf=6 min_done=$k" RUSTRACER_TRAV_MIN_DONE=$k; done

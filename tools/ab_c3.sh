#!/bin/bash
# tools/ab_c3.sh LIB ...: same-box A/B of library builds (tools/dbg/*.so) on the headline: 20- and 240-step launches, the
# rt_fire_all_rays call and the lone resident frame; two rounds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LIBS="$*"
for rep in 1 2; do
for lib in $LIBS; do
  for steps in 20 240; do
    RUSTRACER_AMD_SO=$PWD/tools/dbg/$lib.so timeout -k 10 200 python bench.py --steps $steps --warmup 3 --no-cpu-baseline > gpurun_out/abc3.json 2> gpurun_out/abc3.err || { echo "$lib steps=$steps FAILED"; tail -3 gpurun_out/abc3.err; exit 1; }
    python - "$lib steps=$steps" <<'PY'
import json,sys
d=json.load(open("gpurun_out/abc3.json"))
c=d.get("call") or {}
print(f"{sys.argv[1]:28s} ms/frame {d['ms_per_step']:.4f}  lone frame {c.get('frame_ms_resident',0):.3f} call {c.get('call_ms',0):.3f}", flush=True)
PY
  done
done
done

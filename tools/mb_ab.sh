#!/bin/bash
# tools/mb_ab.sh: the open-hit mailbox experiment (make EXTRA=-DRT_MAILBOX=1 -> tools/dbg/mb.so) against the shipped body:
# parity first, then the headline launch at 20 and 240 frames, with the default kernel choice and with the plain body forced
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RUSTRACER_AMD_SO=$PWD/tools/dbg/mb.so timeout -k 10 300 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread \
  -k "drain_helpers or mirror_children or demo or headline" > gpurun_out/mb_parity.log 2>&1 || { echo "mailbox parity FAILED"; tail -5 gpurun_out/mb_parity.log; exit 1; }
tail -1 gpurun_out/mb_parity.log
for rep in 1 2; do
for lib in base3 mb; do
  for v in default 1; do
    for steps in 20 240; do
      if [ $v = default ]; then unset RUSTRACER_FAST_VARIANT; else export RUSTRACER_FAST_VARIANT=$v; fi
      RUSTRACER_AMD_SO=$PWD/tools/dbg/$lib.so timeout -k 10 200 python bench.py --steps $steps --warmup 3 --no-cpu-baseline > gpurun_out/mb.json 2> gpurun_out/mb.err || { echo "$lib v=$v steps=$steps FAILED"; tail -3 gpurun_out/mb.err; exit 1; }
      python - "$lib variant=$v steps=$steps" <<'PY'
import json,sys
d=json.load(open("gpurun_out/mb.json"))
c=d.get("call") or {}
print(f"{sys.argv[1]:36s} ms/frame {d['ms_per_step']:.4f}  lone frame {c.get('frame_ms_resident')} call {c.get('call_ms')}", flush=True)
PY
    done
  done
done
done

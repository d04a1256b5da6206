set -e
export RUSTRACER_AMD_SO=$PWD/tools/dbg/hbm.so
RUSTRACER_HBM_HELPERS=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu --timeout 300 --timeout-method thread -k "c4 or c5 or hierarchy or random_scenes or flake" > gpurun_out/h18_tests.txt 2>&1 || { tail -30 gpurun_out/h18_tests.txt; exit 1; }
tail -2 gpurun_out/h18_tests.txt
for h in "" 1 "" 1; do
if [ -n "$h" ]; then export RUSTRACER_HBM_HELPERS=1; else unset RUSTRACER_HBM_HELPERS; fi
echo "helpers=[$h]"
timeout -k 10 200 python bench.py --config c5 --width 1920 --height 1080 --aa 4 --steps 1 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 1080p aa4', d['config']['schedule'], [round(x,1) for x in d['config']['region_ms']])" | tee -a gpurun_out/h18_c5.txt
timeout -k 10 200 python bench.py --config c4 --width 1920 --height 1080 --steps 1 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 1080p', d['config']['schedule'], [round(x,1) for x in d['config']['region_ms']])" | tee -a gpurun_out/h18_c5.txt
done

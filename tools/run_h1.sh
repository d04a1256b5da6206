set -e
for v in 2 3; do
echo "== variant $v: frames per launch 1 2 4 8 12"
for f in 1 2 4 8 12; do
RUSTRACER_FAST_VARIANT=$v timeout -k 10 100 python bench.py --steps $((f*4)) --warmup $f --frames-per-launch $f --inflight 1 --schedule onepass --no-cpu-baseline --no-call-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['frames_per_launch'], d['config']['schedule'], round(d['ms_per_step'],4), round(d['roofline']['launch_ms'],3))" | tee -a gpurun_out/h14_frames.txt
done
done
timeout -k 10 900 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/h14_tests.txt 2>&1 || { tail -40 gpurun_out/h14_tests.txt; exit 1; }
tail -3 gpurun_out/h14_tests.txt

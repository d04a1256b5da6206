set -e
timeout -k 10 900 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/h24_tests.txt 2>&1 || { tail -40 gpurun_out/h24_tests.txt; exit 1; }
tail -3 gpurun_out/h24_tests.txt
BENCH_ARGS="--steps 20 --warmup 5 --no-call-timing" bash tools/prof.sh > gpurun_out/r2l_prof_steps20.txt 2>&1 || { tail -20 gpurun_out/r2l_prof_steps20.txt; exit 1; }
tail -32 gpurun_out/r2l_prof_steps20.txt
timeout -k 10 300 python bench.py > gpurun_out/r2l_bench_default.json 2> gpurun_out/r2l_bench_default.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2l_bench_steps20.json 2> gpurun_out/r2l_bench_steps20.err
python -c "
import json
for f in ('default','steps20'):
    d=json.load(open(f'gpurun_out/r2l_bench_{f}.json')); print(f, d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['frac_of_measured_nofma_ceiling'], d['roofline']['achieved_executed'], d['roofline']['frac_executed_of_measured_nofma_ceiling'], d['call']['call_ms'], d['call']['frame_ms_resident'], d['config']['region_ms'])
"

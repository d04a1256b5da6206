set -e
timeout -k 10 900 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/h11_tests.txt 2>&1 || { tail -40 gpurun_out/h11_tests.txt; exit 1; }
tail -3 gpurun_out/h11_tests.txt
timeout -k 10 200 python tools/latency_sizes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h11_sizes.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/h11_bench20.json 2> gpurun_out/h11_bench20.err
python -c "import json; d=json.load(open('gpurun_out/h11_bench20.json')); print(d['ms_per_step'], d['call'], d['roofline']['frac_of_measured_nofma_ceiling'])"

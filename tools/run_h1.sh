set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu --timeout 120 --timeout-method thread -k "c3_pathtracer_demo_small or parameter_corners or aa2_other or headline_frame or render_frames or row_tile or drain_helpers or diagonal or random_scenes" > gpurun_out/h20_tests.txt 2>&1 || { tail -30 gpurun_out/h20_tests.txt; exit 1; }
tail -2 gpurun_out/h20_tests.txt
STEPS=20 tools/ab3.sh nounit unit nounit unit | tee gpurun_out/h20_ab20.txt
STEPS=240 tools/ab3.sh nounit unit | tee gpurun_out/h20_ab240.txt

mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_place.py tests/test_gpu_srmerge.py -x -q -m gpu > gpurun_out/pl_tests.txt 2>&1
timeout 900 python tools/run_configs.py --configs 4 --reps 3 > gpurun_out/rc4.json 2> gpurun_out/rc4.err

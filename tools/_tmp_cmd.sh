mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -m gpu -k "random_probe" > gpurun_out/pc_tests.txt 2>&1

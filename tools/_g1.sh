cd $GRAFT_REPO_ROOT
UKM_FORCE_TICKET=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filetax.py tests/test_golden_vectors.py -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5

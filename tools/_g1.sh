cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filetax.py tests/test_gpu_stress.py tests/test_golden_vectors.py tests/test_gpu_properties.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5
python tools/perf_ops.py --n 1e8 --ops tax 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print({k:round(v.get('kernel_ms',0),3) for k,v in d.items() if 'scalar' not in k})"
UKM_SETOP_SRC=0 python tools/perf_ops.py --n 1e8 --ops tax 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print({k:round(v.get('kernel_ms',0),3) for k,v in d.items() if 'inter' in k and 'scalar' not in k})"

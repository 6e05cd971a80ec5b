cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo "== $t"
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/perf_ops.py --n 1e8 --ops tax 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:round(v['kernel_ms'],3) for k,v in d.items()})"
done

#!/bin/bash
# developer tool (GPU box): radix sort timings of experimental libraries.  args: TAG ... (base = the built library)
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/perf_ops.py --n 1e8 --ops sort 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$t', {k:round(v.get('call_ms',0),3) for k,v in d.items() if 'sort' in k})"
done

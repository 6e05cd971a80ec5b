#!/bin/bash
# developer tool (GPU box): BASELINE config 4 with experimental libraries.  args: TAG ...  (base = the built library)
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/run_configs.py --configs 4 --reps 4 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('$t', k[:12], {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a.endswith('_ms')})"
done

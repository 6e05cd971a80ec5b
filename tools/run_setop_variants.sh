#!/bin/bash
# developer tool (GPU box): set-op kernel timings at 2 x 1e9 keys, one line per variant.  Each argument is one variant:
# "ENV=.. [ENV=..] [lib=TAG]" (environment knobs and / or an experimental library built by tools/build_variant_any.sh)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  L=unikmer_amd/libunikmer_hip.so; envs=""
  for w in $v; do case $w in lib=*) L=unikmer_amd/libukm_exp_${w#lib=}.so;; *) envs="$envs $w";; esac; done
  env $envs UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/perf_ops.py --n 1e9 --ops setop 2>/tmp/err.txt | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:(round(v['kernel_ms'],3), round(v['call_ms'],3)) for k,v in d.items()})"
  grep -m4 "setop\]\|phases" /tmp/err.txt
done

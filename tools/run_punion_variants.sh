#!/bin/bash
# developer tool (GPU box): config 3's plain union through experimental builds of ukm_punion.hip.  args: TAG ... (base = the built library)
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  echo "== $t"
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L UKM_PUNION_DEBUG=1 python tools/c3_tax_bench.py 100 ${PER:-1e8} ${KIND:-none} 3 2>&1 | grep -E "probe  |union ms" | tail -3
done

#!/bin/bash
# developer tool (GPU box): the 2-way kernel with per-record taxids through experimental builds.  args: TAG ... (base = the built library)
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  echo "== $t"
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/tax_scaling.py ${SIZES:-3e8} 2>&1 | tail -2
done

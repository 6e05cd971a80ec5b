#!/bin/bash
# developer tool (this container): an experimental library beside the built one.
# usage: tools/build_exp.sh TAG FILE.hip [-DFLAG ...]   ->  unikmer_amd/libukm_exp_TAG.so (FILE rebuilt with the flags, the other objects as built)
set -e
cd "$(dirname "$0")/.."
tag=$1; f=$2; shift 2
o=/tmp/ukm_exp_${tag}_${f%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c unikmer_amd/csrc/$f -o $o
objs=""
for s in unikmer_amd/csrc/*.o; do
  if [ "$(basename $s)" = "${f%.hip}.o" ]; then objs="$objs $o"; else objs="$objs $s"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unikmer_amd/libukm_exp_$tag.so $objs -ldl
echo unikmer_amd/libukm_exp_$tag.so

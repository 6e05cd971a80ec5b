#!/bin/bash
# developer tool: experimental libukm_exp_TAG.so with extra -D flags for ONE source file
# usage: build_variant_any.sh <setops|sort|encode|scan|nway|kway|tax|ctx|srmerge|...> TAG [-DFOO=1 ...]
set -e
R=/root/repo; C=$R/unikmer_amd/csrc; which=$1; tag=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/ukm_$which.hip -o /tmp/var_${which}_$tag.o
objs=""
for f in ctx setops scan sort encode tax nway kway comm fold punion pfold srmerge; do
  if [ $f = $which ]; then objs="$objs /tmp/var_${which}_$tag.o"; else objs="$objs $C/ukm_$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/unikmer_amd/libukm_exp_$tag.so $objs
echo built $tag

#!/bin/bash
# developer tool: build an experimental variant of libunikmer_hip.so with extra -D flags for
# ukm_setops.hip (other objects are reused).  usage: build_variant.sh TAG [-DFOO=1 ...]
set -e
R=/root/repo
C=$R/unikmer_amd/csrc
tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/ukm_setops.hip -o /tmp/setops_$tag.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/unikmer_amd/libukm_exp_$tag.so $C/ukm_ctx.o /tmp/setops_$tag.o $C/ukm_scan.o $C/ukm_sort.o $C/ukm_encode.o $C/ukm_tax.o $C/ukm_nway.o
echo built $tag

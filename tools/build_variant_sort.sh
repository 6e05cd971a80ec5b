#!/bin/bash
# developer tool: experimental variant with extra -D flags for ukm_sort.hip
set -e
R=/root/repo; C=$R/unikmer_amd/csrc; tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/ukm_sort.hip -o /tmp/sort_$tag.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/unikmer_amd/libukm_exp_$tag.so $C/ukm_ctx.o $C/ukm_setops.o $C/ukm_scan.o /tmp/sort_$tag.o $C/ukm_encode.o $C/ukm_tax.o $C/ukm_nway.o
echo built $tag

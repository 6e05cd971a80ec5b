#!/bin/bash
# developer tool: VGPR / SGPR / LDS / spill figures of every kernel in one .hip file
# usage: kernel_regs.sh FILE.hip [-DFOO=1 ...]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$@" "$f" -o /tmp/kregs_$$.s 2>/dev/null
awk '/^[ \t]*\.amdhsa_kernel /{name=$2} /amdhsa_group_segment_fixed_size/{lds=$2} /amdhsa_next_free_vgpr/{v=$2} /amdhsa_next_free_sgpr/{s=$2} /amdhsa_private_segment_fixed_size/{sc=$2} /\.end_amdhsa_kernel/{printf "%-100s vgpr %3d sgpr %3d lds %6d scratch %d\n", substr(name,1,100), v, s, lds, sc}' /tmp/kregs_$$.s
rm -f /tmp/kregs_$$.s

#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> [pattern]  -> per-kernel VGPR / SGPR / spill / scratch / LDS / occupancy
f=$1; pat=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/_regs.o ${EXTRA_FLAGS} -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/^Function Name/{if(n)print n, l; n=$3; l=""; next} /TotalSGPRs|^VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size/{gsub(/ \[[^]]*\]/,""); l=l" | "$0} END{print n, l}' | c++filt | grep -E "$pat"

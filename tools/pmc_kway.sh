#!/bin/bash
# Run ON THE GPU BOX: PMC passes for the k-way kernel on a config-3-shaped input (every pass bounded by `timeout`)
R=${GRAFT_REPO_ROOT:-/root/repo}
FILES=${1:-100}; SIZE=${2:-1e7}; TAG=${3:-kway}
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/kway_bench.py --files $FILES --size $SIZE --reps 2"
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o p -- $CMD > $O/sq1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq2 -o p -- $CMD > $O/sq2.log 2>&1
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM -d $O/sq3 -o p -- $CMD > $O/sq3.log 2>&1
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
cd $R
for d in sq1 sq2 sq3 fetch write; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) --filter=kway_kernel; done > $O/summary.txt 2>&1
cat $O/summary.txt

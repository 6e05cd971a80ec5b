#!/bin/bash
# Run ON THE GPU BOX: PMC passes for one perf_ops op.  usage: pmc_op.sh OPS FILTER [N]
R=${GRAFT_REPO_ROOT:-/root/repo}
OPS=$1; FILT=$2; N=${3:-1e8}
O=$R/gpurun_out/pmc_$OPS
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/perf_ops.py --n $N --ops $OPS --reps 2"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o p -- $CMD > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq2 -o p -- $CMD > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
cd $R
for d in sq1 sq2 fetch write; do python tools/pmc_summary.py $(dirname $(find $O/$d -name '*counter_collection.csv' | head -1)) --filter=$FILT; done

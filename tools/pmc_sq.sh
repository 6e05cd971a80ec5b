#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats + the two SQ counter passes (no FETCH / WRITE passes) for a command.
# usage: pmc_sq.sh TAG FILTER -- CMD ...   (CMD with absolute paths: the passes run from /tmp)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FILT=$2; shift 3
O=/tmp/pmcsq_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- "$@" > $O/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o p -- "$@" > $O/sq1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq2 -o p -- "$@" > $O/sq2.log 2>&1
cd $R
grep -h "$FILT" $(find $O/trace -name '*kernel_stats.csv') | cut -c1-200
for d in sq1 sq2; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) --filter=$FILT; done
rm -rf $O

#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats + separate PMC passes for an arbitrary command.  usage: pmc_cmd.sh TAG FILTER -- CMD ...
# Raw output is deleted again (it exceeds what gpurun copies back); the summary goes to stdout.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FILT=$2; shift 3
O=/tmp/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- "$@" > $O/trace.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o p -- "$@" > $O/sq1.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq2 -o p -- "$@" > $O/sq2.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- "$@" > $O/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- "$@" > $O/write.log 2>&1
cd $R
grep -h "$FILT" $(find $O/trace -name '*kernel_stats.csv') | cut -c1-200
for d in sq1 sq2 fetch write; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) --filter=$FILT; done
rm -rf $O

#!/bin/bash
# Run ON THE GPU BOX: kernel time + VALU / LDS instruction counts of kernels matching FILTER.  usage: pmc_valu.sh FILTER -- CMD ...
R=${GRAFT_REPO_ROOT:-/root/repo}
FILT=$1; shift 2
O=/tmp/pmcv_$$; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq -o p -- "$@" > $O/sq.log 2>&1 < /dev/null
cd $R
f=$(find $O/sq -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) --filter=$FILT
rm -rf $O

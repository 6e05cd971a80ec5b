#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats + separate PMC passes for tools/run_configs.py.  usage: pmc_configs.sh CONFIGS FILTER
# (e.g. `pmc_configs.sh 3 pu_probe`): output under gpurun_out/pmc_cfg_CONFIGS/, summary printed per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; FILT=$2
O=$R/gpurun_out/pmc_cfg_$CFG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/run_configs.py --configs $CFG --reps 1"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- $CMD > $O/trace.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o p -- $CMD > $O/sq1.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq2 -o p -- $CMD > $O/sq2.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
cd $R
f0=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f0" ] && grep -h "$FILT" $f0 | cut -c1-220
for d in sq1 sq2 fetch write; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $(dirname $f) --filter=$FILT; done

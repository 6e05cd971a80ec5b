#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the
# bench command.  Outputs land in gpurun_out/prof_rNN/ ; tools/summarise_profiles.py turns them
# into the committed files under profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --cpu-sample 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
# (the PMC passes leave the taxid variant out -- UKM_BENCH_NO_TAXID=1: it lies outside the timed region and only adds launches
#  of OTHER kernels; the kernel-trace pass above runs the default command as it is)
export UKM_BENCH_NO_TAXID=1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_sq1 -o bench -- $CMD > $O/pmc_sq1.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o bench -- $CMD > $O/pmc_sq2.log 2>&1
grep -h '"metric"' $O/*.log | head -5 > $O/bench_lines.jsonl
ls -R $O | head -40

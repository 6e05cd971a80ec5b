#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats + PMC passes (own runs, bounded by `timeout`) for the round-2 kernels:
# radix sort / unique / encode / ntHash (perf_ops, 1e8) and the k-way merge (100 x 1e7 union).  Raw output under
# gpurun_out/ops_r02/ ; the per-kernel summaries are printed (and kept) as JSON lines for profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ops_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OPS="python $R/tools/perf_ops.py --n 1e8 --ops sort,unique,encode,nthash --reps 3"
KW="python $R/tools/kway_bench.py --files 100 --size 1e7 --reps 3"
run() { # tag cmd... : one rocprofv3 pass
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv "$@" > $O/$tag.log 2>&1
}
run ops_trace --stats -d $O/ops_trace -o p -- $OPS
run kw_trace --stats -d $O/kw_trace -o p -- $KW
for cmdtag in ops kw; do
  if [ $cmdtag = ops ]; then CMD=$OPS; else CMD=$KW; fi
  run ${cmdtag}_fetch --pmc FETCH_SIZE -d $O/${cmdtag}_fetch -o p -- $CMD
  run ${cmdtag}_write --pmc WRITE_SIZE -d $O/${cmdtag}_write -o p -- $CMD
  run ${cmdtag}_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/${cmdtag}_sq1 -o p -- $CMD
  run ${cmdtag}_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O/${cmdtag}_sq2 -o p -- $CMD
done
cd $R
python tools/summarise_ops.py $O > $O/summary.json 2> $O/summary.err
tail -3 $O/summary.err
python -c "
import json
d = json.load(open('$O/summary.json'))
for k, v in d.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('mean_ms', 'dispatches', 'FETCH_GB', 'WRITE_GB', 'valu_busy', 'wait_any', 'lds_conflict')})
"

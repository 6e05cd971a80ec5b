#!/bin/bash
# developer tool (GPU box): LDS bank-conflict counters of the set-op tile kernel for experimental libraries (ablations)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  if [ "$t" = base ]; then L=$R/unikmer_amd/libunikmer_hip.so; else L=$R/unikmer_amd/libukm_exp_$t.so; fi
  O=$R/gpurun_out/pmc_lds_$t; rm -rf $O; mkdir -p $O
  UKM_LIB_PATH=$L rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O -o p -- python $R/tools/perf_ops.py --n 2e8 --ops setop --reps 2 > $O/log.txt 2>&1
  python - "$O" "$t" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "setop_tile_kernel" not in n: continue
    k = n[n.index("setop_tile_kernel"):].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(sys.argv[2], k[:48], {a: "%.3e" % b for a, b in v.items()}, "ratio %.3f" % (v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
PY
done

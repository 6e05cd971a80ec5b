#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE per launch for the secondary kernels (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_ops
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/perf_ops.py --n 1e8 --ops sort,unique,encode,nthash --reps 2"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(dict)
for kind, fac in (("fetch", 2.0), ("write", 1.0)):   # FETCH_SIZE x2: gfx950 correction (MI355X_MICROARCH.md); KiB units
    f = glob.glob("$O/%s/**/p_counter_collection.csv" % kind, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "anonymous namespace" not in n: continue
        short = n.split("::")[-1].split("(")[0] if "<" not in n else n[n.index("namespace)::") + 12:].split("(")[0]
        agg[short][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        res[k][kind + "_bytes_per_launch"] = sum(d.values()) / len(d) * 1024 * fac
        res[k]["launches"] = len(d)
json.dump(res, open("$R/gpurun_out/pmc_ops/summary.json", "w"), indent=1)
for k, v in res.items(): print(k, {a: (round(b / 1e9, 4) if "bytes" in a else b) for a, b in v.items()})
PY

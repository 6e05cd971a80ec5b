#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE of a coalesced copy and of a per-lane-strip copy over the same 4 GiB (tools/fetch_calib.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
rm -rf /tmp/fc; timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o fc --output-format csv -- /tmp/fetch_calib > /tmp/fc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    key = (r["Kernel_Name"].split("(")[0], r["Grid_Size"])
    agg[key].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
N = 4 << 30
for k, v in sorted(agg.items()):
    raw = sum(x[0] for x in v) / len(v)
    ms = min(x[1] for x in v)
    print("%-18s grid %-10s FETCH_SIZE raw %.4g KiB-units -> x1024 = %.3f x the %d bytes read, x2048 = %.3f x; %.3f ms = %.2f TB/s" %
          (k[0], k[1], raw, raw * 1024 / N, N, raw * 2048 / N, ms, N / ms / 1e9))
PY

#!/bin/bash
# developer tool (GPU box): kernel trace of ONE whole library call -- every kernel of ours per call, not only the dominant
# one (round 5 found a 6 ms statistic kernel and a 5 ms declined attempt this way).
# usage: trace_call.sh TAG python tools/<bench>.py args...   (the bench should repeat the call 3 times)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o t -- "$@" > /tmp/tr_$tag.log 2>&1
grep -E "^(merge_k|union|common|inter|diff|call|files)" /tmp/tr_$tag.log | tail -6
f=$(find /tmp/tr_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::native" not in r["Name"] and "rocprim" not in r["Name"] and "hipcub" not in r["Name"]]
for r in rows[:18]:
    print("  %-80s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

"""print (name, calls, average us, total ms) of the kernels in a rocprofv3 *kernel_stats.csv that match a pattern"""
import csv, re, sys
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for row in csv.DictReader(open(sys.argv[1])):
    if pat.search(row["Name"]):
        print("%-70s calls %5s  avg %10.1f us  total %9.3f ms" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))

"""print a rocprofv3 kernel_stats.csv (names hold commas: use the csv module), optionally filtered by substrings"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:]
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    if flt and not any(f in n for f in flt):
        continue
    print(f"{float(r['AverageNs']) / 1e3:9.2f} us x {int(r['Calls']):5d} = {float(r['TotalDurationNs']) / 1e6:8.3f} ms  {n[:100]}")

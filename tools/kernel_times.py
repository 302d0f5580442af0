"""Average duration of every kernel from a rocprofv3 --kernel-trace --stats csv (kernel_stats.csv).
    python tools/kernel_times.py <kernel_stats.csv> [substring ...]"""
import csv
import sys

keys = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("bnv::", "").replace("void ", "").split("(")[0]
    if (not keys and float(r["Percentage"]) > 0.05) or any(k in name for k in keys):
        print(f"{name[:44]:44s} {int(r['Calls']):6d} calls {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):6.2f} %")

"""Summarise a rocprofv3 --kernel-trace --stats csv directory: python tools/prof_kernels.py DIR [substr...]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
pats = sys.argv[2:]
for r in csv.DictReader(open(f)):
    if not pats or any(p in r['Name'] for p in pats):
        print("%-58s calls %5s avg %9.2f us  min %9.2f  max %9.2f  total %8.3f ms" % (r['Name'][:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))

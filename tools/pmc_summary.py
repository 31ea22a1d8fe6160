"""Average PMC counter values per dispatch of the kernels named on the command line (default: warp_fused)."""
import csv, glob, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("-k=")]
pats = [a[3:] for a in sys.argv[1:] if a.startswith("-k=")] or ["warp_fused_kernel"]
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if any(p in r["Kernel_Name"] for p in pats):
                a = acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40], r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            print("%-40s %-24s avg/dispatch %16.1f  (n=%d)" % (k, c, s / n, n))

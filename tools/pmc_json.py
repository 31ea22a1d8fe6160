"""HBM traffic per dispatch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in runs of their own, as MI355X_MICROARCH.md
prescribes): python tools/pmc_json.py <fetch_dir> <write_dir> <out.json> <dispatches-per-unit> <kernel substring> [...]
FETCH_SIZE is doubled (gfx950 counts the 128-byte requests of wide streaming reads as 64 bytes), WRITE_SIZE is taken as reported; both are KiB."""
import csv, glob, json, sys, collections
fd, wd, out, per = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
pats = sys.argv[5:]


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for p in pats:
                if p in r["Kernel_Name"]:
                    a = acc[p]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs (tools/pmc_feed.sh, tools/pmc_warp.sh)",
       "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
       "kernels": {}}
total = 0.0
for p in pats:
    f_kib = fe[p][0] / max(fe[p][1], 1)
    w_kib = wr[p][0] / max(wr[p][1], 1)
    n = fe[p][1]
    traffic = (2 * f_kib + w_kib) * 1024
    res["kernels"][p] = {"dispatches": n, "FETCH_SIZE_KiB_per_dispatch": round(f_kib, 1), "WRITE_SIZE_KiB_per_dispatch": round(w_kib, 1),
                         "traffic_bytes_per_dispatch": int(traffic)}
res["dispatches_per_unit_note"] = "per-unit traffic = sum over kernels of (their dispatches in the run / units in the run) x traffic per dispatch; units in the run = %g" % per
unit = 0.0
for p in pats:
    unit += res["kernels"][p]["traffic_bytes_per_dispatch"] * res["kernels"][p]["dispatches"] / per
res["traffic_bytes_per_unit"] = int(unit)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

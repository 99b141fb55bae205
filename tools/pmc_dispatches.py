"""List counters per dispatch (in launch order) for kernels whose name contains a pattern."""
import csv, glob, collections, sys
d, pat = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            e = rows.setdefault(int(r["Dispatch_Id"]), {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k in sorted(rows):
    print(k, {a: round(b) for a, b in sorted(rows[k].items())})

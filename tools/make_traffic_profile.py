"""profiles/*_pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

usage: make_traffic_profile.py <dir with FETCH_SIZE csv> <dir with WRITE_SIZE csv> <out.json>
Counters are summed per dispatch; per kernel the value kept is the MEDIAN over its dispatches (the bench also
runs the pipeline once with the operator written out and twice from host arrays: the median is a plain step).
hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide coalesced read stream) -- an upper estimate for
kernels whose reads are not 16-byte-per-lane streams; raw_bytes_per_launch = (FETCH + WRITE) * 1024.
"""
import csv, glob, json, sys, collections

KERNELS = {"knn_cell": "knn_lane_kernel", "knn_strip_kernel": "knn_strip_kernel", "centroid": "centroid_",
           "locate_pass0": "locate_pass_kernel<true, int, true, int, true>", "locate_pass0_exact": "locate_pass_kernel<true, int, true, int, false>",
           "gather": "gather8_kernel", "cell_scatter": "cell_scatter_kernel", "target_scatter": "target_scatter_kernel"}


def per_dispatch(d, counter):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return out


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
res = {"_note": __doc__.strip().split("\n\n", 1)[1].replace("\n", " ")}
for stage, pat in KERNELS.items():
    def pick(table):
        vals = [v for name, disp in table.items() if pat in name for v in disp.values()]
        if not vals:
            return None
        vals.sort()
        return vals[len(vals) // 2]
    f, w = pick(fetch), pick(write)
    if f is None or w is None:
        continue
    res[stage] = {"FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w),
                  "hbm_bytes_per_launch": round((2 * f + w) * 1024), "raw_bytes_per_launch": round((f + w) * 1024)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))

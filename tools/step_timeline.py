"""Timeline of ONE hot-path step from a rocprofv3 kernel trace (csv): per dispatch its start relative to the
step's first kernel, its duration and the idle gap since the previous dispatch ended.  The step is the shortest
run of dispatches between two centroid kernels.   usage: step_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "centroid_bbox_kernel" in r["Kernel_Name"]]
# the SHORTEST run between two centroid kernels: a plain timed step (the bench also runs the pipeline with the operator
# written out and from host arrays, which allocate in mid-step)
spans = [(int(rows[starts[q + 1] - 1]["End_Timestamp"]) - int(rows[starts[q]]["Start_Timestamp"]), q) for q in range(len(starts) - 1)]
q = min(spans)[1]
a, b = starts[q], starts[q + 1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0; busy = 0; gaps = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    gap = s - prev_end
    print("%9.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, name))
    busy += e - s; gaps += max(gap, 0); prev_end = max(prev_end, e)
print("dispatches %d  busy %.1f us  gaps %.1f us  span %.1f us" % (len(step), busy / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))

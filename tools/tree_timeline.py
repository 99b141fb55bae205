"""Kernels of the last step of a rocprofv3 kernel trace (sqlite output), in time order.  tree_timeline.py <results.db> [min ms]"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
first = [i for i, r in enumerate(rows) if 'centroid' in r[0]][-1]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_]+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:60]
last = rows[first:]
for n, s, e, g, w in last:
    if (e - s) / 1e6 >= floor:
        print(f"{(s - last[0][1]) / 1e6:8.3f} {(e - s) / 1e6:7.3f} {g // max(w, 1):8d} {short(n)[:90]}")
print("span", (last[-1][2] - last[0][1]) / 1e6, "ms; busy", sum(e - s for _, s, e, _, _ in last) / 1e6)

"""Print the interesting numbers of bench.py JSON lines (one file per argument)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads((sys.stdin if f == '-' else open(f)).read().strip().splitlines()[-1])
    except Exception as e:
        print("==", f, "unreadable:", e); continue
    print("==", f, "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "n_gpus", d["n_gpus"], "nfailed", d.get("nfailed", d.get("nmissing")))
    for s, v in d["stages"].items():
        print("   %-14s" % s, {k: v[k] for k in v if k in ("ms", "achieved_GBps", "actual_GBps", "frac", "frac_actual", "equals_fused_values")})
    for k in ("allgather", "host_arrays", "parity_vs_cpu_sample", "speedup_vs_cpu_baseline"):
        if k in d:
            print("  ", k, d[k])

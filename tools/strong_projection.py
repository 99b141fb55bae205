#!/usr/bin/env python3
"""PROJECTION of strong scaling from ONE GPU (GPU box): for G in {2, 4, 8} every rank's share of the workload's target
set is run alone (bench.py --gpus 1 --as-rank r/G: no process group, no collective), one after the other.  Writes
{G: {per_rank_ms, max_ms, replicated_ms (centroids + grid build of the replicated source), projected points/s = all
targets / max_ms, projected efficiency against the G = 1 step of the same session}} plus the fixed cost of a step
(the pipeline on the full source mesh with 27 targets: replicated part + launches).  NOT a scaling measurement --
the all-gather and any cross-rank interference are not in it.   usage: strong_projection.py out.json [workload]"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "metric"
extra = sys.argv[3:]


def run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline"] + args + extra, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"bench.py {args} failed:\n{r.stderr[-2000:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def summary(d):
    st = d["stages"]
    return {"ms_per_step": round(d["ms_per_step"], 4), "targets": d["config"]["targets_per_gpu"],
            "replicated_ms": d["replicated_per_rank"]["ms"], "knn_query_ms": st["knn_query"]["ms"], "locate_ms": st["locate"]["ms"]}


res = {"_note": " ".join(__doc__.split()), "workload": workload}
one = run(["--as-rank", "0/1"])
res["G1"] = summary(one)
print("G=1", res["G1"], flush=True)
fixed = run(["--n-tgt", "3", "--as-rank", "0/1"])
res["fixed_cost_of_a_step"] = {"ms_per_step": round(fixed["ms_per_step"], 4), "replicated_ms": fixed["replicated_per_rank"]["ms"],
                               "note": "the full source mesh with 27 targets: centroids + grid build + every launch of the step"}
print("fixed", res["fixed_cost_of_a_step"], flush=True)
ntot = one["config"]["targets_total"]
for G in (2, 4, 8):
    ranks = [summary(run(["--as-rank", f"{r}/{G}"])) for r in range(G)]
    mx = max(r["ms_per_step"] for r in ranks)
    res[f"G{G}"] = {"per_rank": ranks, "max_ms_per_step": mx,
                    "replicated_share_of_the_slowest_rank": round(max(r["replicated_ms"] for r in ranks) / mx, 4),
                    "projected_points_per_s": ntot / (mx * 1e-3),
                    "projected_efficiency_vs_G1": round(res["G1"]["ms_per_step"] / (G * mx), 4)}
    print(f"G={G}", {k: v for k, v in res[f"G{G}"].items() if k != "per_rank"}, flush=True)
    json.dump(res, open(out, "w"), indent=1)
json.dump(res, open(out, "w"), indent=1)

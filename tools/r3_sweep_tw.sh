#!/bin/bash
# (GPU box) sweep of the lane kernel window parameters T / W on the metric workload (round 3 tuning)
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/pytest_a.log
tail -3 gpurun_out/r3/pytest_a.log
for cfg in "4 3" "4 4" "4 2" "6 4" "6 5" "5 4" "2 2" "1 1"; do set -- $cfg
  MM_KNN_LANE_T=$1 MM_KNN_LANE_W=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3/tw_$1_$2.json 2> gpurun_out/r3/tw_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3/tw_$1_$2.json")); s=d["stages"]
    print("T=$1 W=$2 step %.3f knn_cell %.3f knn_query %.3f locate %.3f nfailed %d" % (d["ms_per_step"], s["knn_cell"]["ms"], s["knn_query"]["ms"], s["locate"]["ms"], d["nfailed"]))
except Exception as e: print("T=$1 W=$2 failed", e)
PY
done

#!/bin/bash
# time of the K = 8 strip kernel when it ends after phase n of the first round (MM_KNN_DBG_STOP)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/phase_sweep.txt
for st in 1 2 3 4 5 6 7 8 0; do
  MM_KNN_DBG_STOP=$st timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print($st, d['stages']['knn_cell']['ms'])" >> gpurun_out/phase_sweep.txt || exit 1
done

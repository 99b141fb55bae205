#!/bin/bash
# (GPU box) the round's regression gate: GPU test suite, default bench, randomised checkers.
#   tools/r3_check.sh <tag> [fuzz cases]
TAG=${1:-x}; N=${2:-300}
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3/pytest_$TAG.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench_$TAG.json 2> gpurun_out/r3/bench_$TAG.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3/bench_$TAG.json")); s=d["stages"]
    print("bench: step %.3f ms | centroid %.3f build %.3f knn_query %.3f (lane %.3f) locate %.3f | nfailed %d" % (d["ms_per_step"], s["centroid"]["ms"], s["knn_build"]["ms"], s["knn_query"]["ms"], s["knn_cell"]["ms"], s["locate"]["ms"], d["nfailed"]))
except Exception as e: print("bench failed", e)
PY
timeout -k 10 600 python tools/fuzz_knn.py $N 7001 > gpurun_out/r3/fuzz_knn_$TAG.log 2>&1; echo "fuzz_knn rc=$?"; tail -1 gpurun_out/r3/fuzz_knn_$TAG.log
MM_KNN_KERNEL=lane timeout -k 10 600 python tools/fuzz_knn.py $N 7002 > gpurun_out/r3/fuzz_knn_lane_$TAG.log 2>&1; echo "fuzz_knn(lane forced) rc=$?"; tail -1 gpurun_out/r3/fuzz_knn_lane_$TAG.log
timeout -k 10 600 python tools/fuzz_pipeline.py $N 7003 > gpurun_out/r3/fuzz_pipe_$TAG.log 2>&1; echo "fuzz_pipeline rc=$?"; tail -1 gpurun_out/r3/fuzz_pipe_$TAG.log

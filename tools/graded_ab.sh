#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in "3 0.1 9" "3 0.1 70" "3 0.05 600"; do
  set -- $cfg
  echo "== refine=$1 share=$2 cells=$3" >> gpurun_out/graded_ab.txt
  MM_KNN_REFINE=$1 MM_KNN_REFINE_SHARE=$2 MM_KNN_REFINE_CELLS=$3 MM_KNN_DEBUG=1 timeout -k 10 300 python3 tools/bench_knn_graded.py >> gpurun_out/graded_ab.txt 2>&1 || exit 1
done

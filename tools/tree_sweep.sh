#!/bin/bash
# Round 4, tuning of the kNN tree's window choice on the graded 10M mesh: hand-overs and stage times per setting
# (cell_min tile_max of the first pass, then of the second).
mkdir -p gpurun_out/tree
for setting in "0 0 0 0" "2 700 4 500" "3 640 8 500"; do
  set -- $setting
  MM_TREE_CELL_MIN=$1 MM_TREE_TILE_MAX=$2 MM_TREE_CELL_MIN2=$3 MM_TREE_TILE_MAX2=$4 MM_KNN_DEBUG=1 timeout -k 10 120 python tools/bench_graded_mesh.py 216 ${POWER:-1.5} > gpurun_out/tree/sw.json 2> gpurun_out/tree/sw.err
  echo "pass 1: cell_min=$1 tile_max=$2, pass 2: $3 $4; rc=$? $(cat gpurun_out/tree/sw.json)"
  grep "\[mm_knn\] tree" gpurun_out/tree/sw.err | sort | uniq -c | cut -c1-200
done

#!/bin/bash
# Round 4, tuning: thin layers per cell layer (T) and first-window half-width (W) of the tree's lane kernel, graded 10M mesh.
mkdir -p gpurun_out/tree
for tw in "6 5" "6 4" "8 6" "8 7" "10 8" "10 7" "4 3"; do
  set -- $tw
  MM_KNN_TREE=1 MM_TREE_T=$1 MM_TREE_W=$2 timeout -k 10 120 python tools/bench_graded_mesh.py 216 1.5 > gpurun_out/tree/sw.json 2> gpurun_out/tree/sw.err
  echo "T=$1 W=$2 rc=$? $(cat gpurun_out/tree/sw.json)"
done

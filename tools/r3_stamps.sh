#!/bin/bash
# (GPU box) diagnostic build with phase stamps in the lane kernel, on the box's own copy of the tree
mkdir -p gpurun_out/r3
touch multimesh_amd/csrc/mm_knn.hip
make -C multimesh_amd/csrc -j16 EXTRA=-DMM_LANE_STAMPS > gpurun_out/r3/stamps_make.log 2>&1 || { tail -5 gpurun_out/r3/stamps_make.log; exit 1; }
for cfg in "${@:-6 4}"; do set -- $cfg
  echo "== T=$1 W=$2"; MM_KNN_LANE_T=$1 MM_KNN_LANE_W=$2 timeout -k 10 200 python tools/lane_stamps.py 2>&1 | tail -11
done

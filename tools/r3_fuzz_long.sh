#!/bin/bash
# (GPU box) the long randomised runs of the round
mkdir -p gpurun_out/r3
( timeout -k 10 1000 python tools/fuzz_knn.py 3000 31001 > gpurun_out/r3/fz_knn.log 2>&1; echo "fuzz_knn rc=$?"; tail -1 gpurun_out/r3/fz_knn.log )
( MM_KNN_KERNEL=lane timeout -k 10 1000 python tools/fuzz_knn.py 3000 31002 > gpurun_out/r3/fz_knn_lane.log 2>&1; echo "fuzz_knn lane rc=$?"; tail -1 gpurun_out/r3/fz_knn_lane.log )
( MM_KNN_LANE_W=2 timeout -k 10 1000 python tools/fuzz_knn.py 1500 31003 > gpurun_out/r3/fz_knn_w2.log 2>&1; echo "fuzz_knn W=2 rc=$?"; tail -1 gpurun_out/r3/fz_knn_w2.log )
( timeout -k 10 1000 python tools/fuzz_pipeline.py 2000 31004 > gpurun_out/r3/fz_pipe.log 2>&1; echo "fuzz_pipeline rc=$?"; tail -1 gpurun_out/r3/fz_pipe.log )
( timeout -k 10 600 python tools/fuzz_pipeline.py 20 31005 -1 big > gpurun_out/r3/fz_pipe_big.log 2>&1; echo "fuzz_pipeline big rc=$?"; tail -1 gpurun_out/r3/fz_pipe_big.log )
( timeout -k 10 1000 python tools/fuzz_gll.py 1000 31006 > gpurun_out/r3/fz_gll.log 2>&1; echo "fuzz_gll rc=$?"; tail -1 gpurun_out/r3/fz_gll.log )
( timeout -k 10 600 python tools/fuzz_unique.py 300 31007 > gpurun_out/r3/fz_unique.log 2>&1; echo "fuzz_unique rc=$?"; tail -1 gpurun_out/r3/fz_unique.log )

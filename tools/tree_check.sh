#!/bin/bash
# Round 4, the kNN tree: fuzz with the tree forced, the graded tests, the graded bench, a kernel trace of one power.
# usage (on the GPU box): tools/tree_check.sh [power for the trace]
mkdir -p gpurun_out/tree
MM_KNN_TREE=1 timeout -k 10 200 python tools/fuzz_knn.py ${FUZZ:-150} 777 > gpurun_out/tree/fuzz1.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/tree/fuzz1.log
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "graded or two_densities or knn" > gpurun_out/tree/pytest1.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/tree/pytest1.log
MM_KNN_DEBUG=1 timeout -k 10 200 python tools/bench_graded_mesh.py 216 1.5 2.2 > gpurun_out/tree/graded.json 2> gpurun_out/tree/graded.err; echo "bench rc=$?"; cat gpurun_out/tree/graded.json; grep "\[mm_knn\] tree" gpurun_out/tree/graded.err | sort | uniq -c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-1.5}
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d gpurun_out/tree/prof -o p -- python tools/bench_graded_mesh.py 216 $P > gpurun_out/tree/prof.log 2>&1; echo prof rc=$?

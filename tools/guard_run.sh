#!/bin/bash
# (GPU box) the randomised checkers and the GPU tests with GUARDED allocations: MM_GUARD_ALLOC=1 maps every
# device allocation of the library -- the Python layer's input and output arrays included -- so that it ends
# at the end of its mapping, with unmapped addresses behind it; an access past an array faults at once.
#   tools/guard_run.sh <tag> [cases]
TAG=${1:-x}; N=${2:-300}
mkdir -p gpurun_out/guard
export MM_GUARD_ALLOC=1
timeout -k 10 500 python tools/fuzz_knn.py $N 8101 > gpurun_out/guard/fuzz_knn_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_knn rc=$rc"; tail -1 gpurun_out/guard/fuzz_knn_$TAG.log
[ $rc -eq 0 ] || exit $rc
MM_KNN_KERNEL=lane timeout -k 10 500 python tools/fuzz_knn.py $N 8102 > gpurun_out/guard/fuzz_knn_lane_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_knn (lane forced) rc=$rc"; tail -1 gpurun_out/guard/fuzz_knn_lane_$TAG.log
[ $rc -eq 0 ] || exit $rc
MM_KNN_TREE=1 timeout -k 10 500 python tools/fuzz_knn.py $N 8107 > gpurun_out/guard/fuzz_knn_tree_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_knn (tree forced) rc=$rc"; tail -1 gpurun_out/guard/fuzz_knn_tree_$TAG.log
[ $rc -eq 0 ] || exit $rc
MM_KNN_TREE=1 FP_MODE=tol timeout -k 10 500 python tools/fuzz_pipeline.py $N 8108 > gpurun_out/guard/fuzz_pipe_tree_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_pipeline (tree forced, MM_FP_TOL) rc=$rc"; tail -1 gpurun_out/guard/fuzz_pipe_tree_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/fuzz_pipeline.py $N 8103 > gpurun_out/guard/fuzz_pipe_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_pipeline rc=$rc"; tail -1 gpurun_out/guard/fuzz_pipe_$TAG.log
[ $rc -eq 0 ] || exit $rc
FP_MODE=tol timeout -k 10 500 python tools/fuzz_pipeline.py $N 8106 > gpurun_out/guard/fuzz_pipe_tol_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_pipeline (MM_FP_TOL) rc=$rc"; tail -1 gpurun_out/guard/fuzz_pipe_tol_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/fuzz_gll.py $N 8104 > gpurun_out/guard/fuzz_gll_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_gll rc=$rc"; tail -1 gpurun_out/guard/fuzz_gll_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/fuzz_unique.py $N 8105 > gpurun_out/guard/fuzz_unique_$TAG.log 2>&1; rc=$?; echo "guarded fuzz_unique rc=$rc"; tail -1 gpurun_out/guard/fuzz_unique_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/guard/pytest_$TAG.log 2>&1; rc=$?; echo "guarded pytest rc=$rc"; tail -3 gpurun_out/guard/pytest_$TAG.log
exit $rc

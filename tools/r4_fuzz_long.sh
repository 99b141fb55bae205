#!/bin/bash
# (GPU box) the long randomised runs of round 4: fuzz_pipeline.py in MM_FP_TOL (node ids exact, weights / values to the
# stated tolerance) and in MM_FP_EXACT, fuzz_knn.py default and with the lane kernel forced.  Progress goes to
# gpurun_out/r4_fuzz/ (a silent GPU run is taken to be hung).   tools/r4_fuzz_long.sh [cases per pipeline batch]
N=${1:-1000}
O=gpurun_out/r4_fuzz; mkdir -p $O
for seed in 41001 41002 41003; do
  FP_MODE=tol timeout -k 10 1100 python tools/fuzz_pipeline.py $N $seed > $O/pipeline_tol_$seed.log 2>&1; echo "tol $seed rc=$? $(tail -1 $O/pipeline_tol_$seed.log)" | tee -a $O/summary.log
done

#!/bin/bash
# tuning: rebuild the library with different strip shapes on the GPU box and time the kNN stage
# usage: tune_strip.sh "Z CAP [extra flags] [per_cell]" ...
set -e
for v in "$@"; do
  set -- $v
  touch multimesh_amd/csrc/mm_knn.hip
  make -C multimesh_amd/csrc EXTRA="-DMM_STRIP_Z=$1 -DMM_STRIP_CAP=$2 $3" > gpurun_out/tune_build_$1_$2.log 2>&1
  tag=$1_$2_${4:-8}
  MM_KNN_PER_CELL=${4:-8} MM_KNN_DEBUG=1 timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/tune_$tag.json 2> gpurun_out/tune_$tag.err
  echo "Z=$1 cap=$2 per_cell=${4:-8} done"
done

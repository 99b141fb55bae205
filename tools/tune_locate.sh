#!/bin/bash
# tuning: rebuild with different Newton caps for the compacting passes and time the locate stage
set -e
for v in "$@"; do
  touch multimesh_amd/csrc/mm_locate_hex8.hip
  make -C multimesh_amd/csrc EXTRA="-DMM_PASS_ITERS=$v" > gpurun_out/tunel_build_$v.log 2>&1
  timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/tunel_$v.json 2> gpurun_out/tunel_$v.err
  echo "cap=$v done"
done

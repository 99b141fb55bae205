#!/bin/bash
# (GPU box) locate experiments: counts of rounds / solves, and the (unsafe) z-box skip as an upper bound
mkdir -p gpurun_out/r3
touch multimesh_amd/csrc/mm_locate_hex8.hip
make -C multimesh_amd/csrc -j16 EXTRA="-DMM_LOCATE_COUNT" > /tmp/v.log 2>&1 || { tail -5 /tmp/v.log; exit 1; }
MM_LOCATE_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep mm_locate | tail -2
for z in 5 50; do
  bash tools/r3_variant.sh mm_locate_hex8.hip "-DMM_EXP_ZBOX=$z"
done
touch multimesh_amd/csrc/mm_locate_hex8.hip
make -C multimesh_amd/csrc -j16 EXTRA="-DMM_LOCATE_COUNT -DMM_EXP_ZBOX=5" > /tmp/v.log 2>&1 || { tail -5 /tmp/v.log; exit 1; }
MM_LOCATE_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep mm_locate | tail -2

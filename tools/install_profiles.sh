#!/bin/bash
# gpurun_out/refresh/* (made by tools/refresh_profiles.sh on the GPU box) -> profiles/<round>_*:  install_profiles.sh r03
R=${1:?round tag, e.g. r03}; O=gpurun_out/refresh
for f in bench.json bench_under_rocprof.json kernel_stats.csv pmc_traffic.json knn_counters.json gll_counters.json step_timeline.txt \
         bench_cfg3.json bench_cfg4_shard0.json bench_cfg4_shard7.json bench_cfg5.json gll_cfg5_kernel_stats.csv \
         bench_rccl_world1.json bench_rehearsal_2ranks_strong.json bench_rehearsal_2ranks_cfg5.json \
         bench_exact.json mem_counters_tol.json knn_counters_exact.json cpu_full.json strong_projection.json; do
  [ -s $O/$f ] && cp $O/$f profiles/${R}_$f || echo "missing: $f"
done

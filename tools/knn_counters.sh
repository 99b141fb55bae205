#!/bin/bash
# SQ / TCP / TCC / LDS counters of the bench's kernels (run on the GPU box; the VALU instruction classes of
# bench.py's roofline_valu come from the SQ_INSTS_VALU_* group): one rocprofv3 pass per
# counter group, each with --kernel-trace only, summarised per kernel into $1 (default
# gpurun_out/counters/knn_counters.json) by tools/make_counter_profile.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/counters; rm -rf $O; mkdir -p $O
OUT=${1:-$O/knn_counters.json}
# MM_COUNTER_ARGS: another workload (e.g. "bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline" ->
# profiles/*_gll_counters.json); MM_COUNTER_VALU_ONLY=1: only the instruction-count groups roofline_valu needs
ARGS=${MM_COUNTER_ARGS:-"bench.py --steps 2 --warmup 1 --no-cpu-baseline"}
i=0
for group in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ -n "$MM_COUNTER_VALU_ONLY" ]; then case "$group" in SQ_WAVES*|SQ_INSTS_VALU_ADD*) ;; *) continue;; esac; fi
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/pass$i -o p -- python3 $ARGS > $O/pass$i.log 2>&1 \
    || echo "pass $i ($group) failed: see $O/pass$i.log"
done
python3 tools/make_counter_profile.py $O $OUT

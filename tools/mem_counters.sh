#!/bin/bash
# TA / TCP / TD / UTCL1 counters of the bench's kernels (GPU box): where a gather-heavy kernel waits -- address
# processing, tag lookups, L2 returns, address translation.  One rocprofv3 pass per group (--kernel-trace only).
#   tools/mem_counters.sh [out.json]      MM_COUNTER_ARGS overrides the command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/memcounters; rm -rf $O; mkdir -p $O
OUT=${1:-$O/mem_counters.json}
ARGS=${MM_COUNTER_ARGS:-"bench.py --steps 2 --warmup 1 --no-cpu-baseline"}
i=0
# (small groups: a block has few counter slots -- six TA counters in one pass abort rocprofv3 with "exceeds the
# capabilities of the hardware" and leave it hanging; every pass runs under its own timeout)
for group in \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  echo "pass $i: $group" >> $O/progress.log
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/pass$i -o p -- python3 $ARGS > $O/pass$i.log 2>&1 \
    || echo "pass $i ($group) failed: see $O/pass$i.log" | tee -a $O/progress.log
done
python3 tools/make_counter_profile.py $O $OUT > /dev/null
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("locate_pass_kernel", "knn_lane_kernel", "gather8_kernel", "centroid_bbox_kernel"):
    if k in d:
        print(k)
        for a, b in sorted(d[k].items()):
            if "run_total" not in a:
                print("   %-45s %s" % (a, b))
PY

#!/bin/bash
# (GPU box) rebuild with extra flags and print cfg5's step and its GLL kernels' average durations (rocprofv3 kernel trace):
#   tools/r4_variant_cfg5.sh "<EXTRA flags>"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
X=$1
touch multimesh_amd/csrc/mm_locate_gll.hip
make -C multimesh_amd/csrc -j16 EXTRA="$X" > /tmp/variant_make.log 2>&1 || { tail -5 /tmp/variant_make.log; exit 1; }
rm -rf /tmp/v5; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v5 -o p -- python3 bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/v5.json 2>/dev/null
python3 - "$X" <<'PY'
import csv, json, sys
d = json.loads(open('/tmp/v5.json').read().strip().splitlines()[-1])
rows = {r['Name']: r for r in csv.DictReader(open('/tmp/v5/p_kernel_stats.csv'))}
def avg(pat):
    for n, r in rows.items():
        if pat in n:
            return float(r['AverageNs']) / 1e6, int(r['Calls'])
    return (0, 0)
print('EXTRA=%s: step %.3f locate %.3f | first_pass %.3f  pass %.3f x%d  values %.3f' % (sys.argv[1], d['ms_per_step'], d['stages']['locate']['ms'], avg('first_pass_kernel')[0], avg('locate_gll_pass_kernel')[0], avg('locate_gll_pass_kernel')[1] // 7, avg('gll_values_kernel')[0]))
PY

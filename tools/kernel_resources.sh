#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel of one translation unit (compiler view):
#   tools/kernel_resources.sh mm_knn.hip [name filter] [extra flags]
cd "$(dirname "$0")/../multimesh_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
  -Wno-unused-function -Rpass-analysis=kernel-resource-usage $3 -c "$1" -o /tmp/kres_$$.o 2>&1 | python3 -c '
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:\S+: )?\s*(.*?) \[-Rpass-analysis", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
import subprocess
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if flt and flt not in dem: continue
    print("%-58s VGPR %-4s AGPR %-3s spill v%-3s s%-4s scratch %-4s LDS %-6s occ %s" % (dem[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$2"
rm -f /tmp/kres_$$.o

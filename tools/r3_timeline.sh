#!/bin/bash
# (GPU box) per-dispatch timeline of one step under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3}/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/err.log || { tail -3 $O/err.log; exit 1; }
python3 tools/step_timeline.py $O > gpurun_out/${1:-r3}/step_timeline.txt; cat gpurun_out/${1:-r3}/step_timeline.txt

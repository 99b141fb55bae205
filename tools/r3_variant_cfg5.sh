#!/bin/bash
# (GPU box) rebuild one translation unit with extra flags and run the cfg5 bench:  r3_variant_cfg5.sh <file> "<flags>"
F=$1; X=$2
touch multimesh_amd/csrc/$F
make -C multimesh_amd/csrc -j16 EXTRA="$X" > /tmp/variant_make.log 2>&1 || { tail -5 /tmp/variant_make.log; exit 1; }
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('EXTRA=$X: step %.3f | unique %.3f knn_query %.3f locate %.3f' % (d['ms_per_step'], s['unique_points']['ms'], s['knn_query']['ms'], s['locate']['ms']))"

#!/bin/bash
# (GPU box) one bench run, stage table on one line:  tools/stage_line.sh <tag> [bench args...]
TAG=$1; shift
python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('$TAG: step %.3f | centroid %.3f build %.3f knn_query %.3f (lane %.3f) locate %.3f nfailed %d' % (d['ms_per_step'], s['centroid']['ms'], s['knn_build']['ms'], s['knn_query']['ms'], s['knn_cell']['ms'], s['locate']['ms'], d['nfailed']))"

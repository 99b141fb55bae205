#!/bin/bash
# (GPU box) sweep of the search grid's density and the lane kernel's strip length / thin layers
for cfg in "8 7 6 5" "6 7 6 5" "6 9 5 4" "5 10 5 4" "10 6 6 5" "12 5 8 6" "8 6 8 6" "8 5 8 7" "10 5 8 6"; do set -- $cfg
  MM_KNN_PER_CELL=$1 MM_KNN_LANE_Z=$2 MM_KNN_LANE_T=$3 MM_KNN_LANE_W=$4 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('per_cell=$1 Z=$2 T=$3 W=$4: step %.3f | build %.3f lane %.3f knn_query %.3f locate %.3f nfailed %d' % (d['ms_per_step'], s['knn_build']['ms'], s['knn_cell']['ms'], s['knn_query']['ms'], s['locate']['ms'], d['nfailed']))"
done

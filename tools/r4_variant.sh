#!/bin/bash
# (GPU box) rebuild the library with extra compile flags on the box's own copy of the tree and run the bench in
# MM_FP_TOL:   tools/r4_variant.sh "<file to touch>" "<EXTRA flags>" [bench args...]
F=$1; X=$2; shift 2
touch multimesh_amd/csrc/$F
make -C multimesh_amd/csrc -j16 EXTRA="$X" > /tmp/variant_make.log 2>&1 || { tail -5 /tmp/variant_make.log; exit 1; }
MM_FP_MODE=${MM_FP_MODE-tol} python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('EXTRA=$X: step %.3f | lane %.3f knn_query %.3f locate %.3f (pass %.3f) nfailed %d' % (d['ms_per_step'], s['knn_cell']['ms'], s['knn_query']['ms'], s['locate']['ms'], s['locate_pass0']['ms'], d['nfailed']))"

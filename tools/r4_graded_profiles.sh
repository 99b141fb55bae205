#!/bin/bash
# Round 4 (GPU box): the graded-mesh and graded-cloud benches with the stack of density levels (MM_KNN_TREE=0), with the
# tree forced (=1) and as shipped (unset), and a kernel timeline of one tree-served pass.  Output: gpurun_out/graded_r04/.
OUT=gpurun_out/graded_r04; mkdir -p $OUT
for mode in 0 1 default; do
  if [ $mode = default ]; then unset MM_KNN_TREE; else export MM_KNN_TREE=$mode; fi
  MM_KNN_DEBUG=1 timeout -k 10 200 python tools/bench_graded_mesh.py 216 1.0 1.5 2.2 > $OUT/mesh_$mode.json 2> $OUT/mesh_$mode.err; echo "mesh $mode rc=$? $(cat $OUT/mesh_$mode.json)"
  grep -c "\[mm_knn\] tree" $OUT/mesh_$mode.err
  timeout -k 10 200 python tools/bench_knn_graded.py > $OUT/clouds_$mode.json 2> $OUT/clouds_$mode.err; echo "clouds $mode rc=$? $(cat $OUT/clouds_$mode.json)"
done
export MM_KNN_TREE=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python tools/bench_graded_mesh.py 216 1.5 > $OUT/prof.log 2>&1; echo prof rc=$?
python tools/tree_timeline.py $OUT/prof/p_results.db 0.02 > $OUT/timeline_p1.5_tree.txt; tail -3 $OUT/timeline_p1.5_tree.txt

#!/bin/bash
# Re-create the judged artefacts under profiles/ from the current build (run on the GPU box):
#   kernel-trace summary of the bench command, the bench line under the profiler, the two counter
#   passes the roofline's `traffic` comes from (each in its own run, with --kernel-trace only), and the
#   SQ / TCP / TCC counter groups of tools/knn_counters.sh.  Outputs in gpurun_out/refresh/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
python3 tools/make_traffic_profile.py $O/fetch $O/write $O/pmc_traffic.json > /dev/null || exit 1
cp $O/stats/p_kernel_stats.csv $O/kernel_stats.csv
bash tools/knn_counters.sh $O/counters.json > $O/counters.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard0.json 2>/dev/null
python3 bench.py --workload cfg4 --cfg4-shard 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard7.json 2>/dev/null
ls $O

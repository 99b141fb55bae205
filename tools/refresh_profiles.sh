#!/bin/bash
# Re-create the judged artefacts under profiles/ from the current build (run on the GPU box; outputs in
# gpurun_out/refresh/, copied to profiles/rNN_* by tools/install_profiles.sh):
#   kernel-trace summary of the bench command, the bench line under the profiler, the two counter passes the
#   roofline's `traffic` comes from (each in its own run, with --kernel-trace only), the SQ / TCP / TCC counter
#   groups of tools/knn_counters.sh (VALU instruction classes included), the per-dispatch timeline of one step,
#   the other workloads (cfg3, cfg4 shards, cfg5 with its own kernel summary), the RCCL path at world size 1 and
#   a 2-rank strong-scaling rehearsal on the one GPU.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
R=${1:-r04}   # round tag: the fresh counter profiles are put under profiles/ (of the box's copy) before the plain
              # bench lines are made, so that their roofline_valu reads THIS build's instruction counts
step() { echo "[refresh] $*"; }
step kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
cp $O/stats/p_kernel_stats.csv $O/kernel_stats.csv
python3 tools/step_timeline.py $O/stats > $O/step_timeline.txt
step pmc traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
python3 tools/make_traffic_profile.py $O/fetch $O/write $O/pmc_traffic.json > /dev/null || exit 1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json   # (the bench lines' `traffic` reads it)
step counters
bash tools/knn_counters.sh $O/knn_counters.json > $O/counters.log 2>&1
step memory-side counters
bash tools/mem_counters.sh $O/mem_counters_tol.json > $O/mem_counters.log 2>&1
[ -s $O/mem_counters_tol.json ] && cp $O/mem_counters_tol.json profiles/${R}_mem_counters_tol.json
MM_COUNTER_ARGS="bench.py --fp-mode exact --steps 2 --warmup 1 --no-cpu-baseline" MM_COUNTER_VALU_ONLY=1 \
  bash tools/knn_counters.sh $O/knn_counters_exact.json > $O/counters_exact.log 2>&1
MM_COUNTER_ARGS="bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline" MM_COUNTER_VALU_ONLY=1 \
  bash tools/knn_counters.sh $O/gll_counters.json > $O/gll_counters.log 2>&1
[ -s $O/knn_counters.json ] && cp $O/knn_counters.json profiles/${R}_knn_counters.json
[ -s $O/gll_counters.json ] && cp $O/gll_counters.json profiles/${R}_gll_counters.json
step cfg5 kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o p -- python3 bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_under_rocprof.json 2> $O/stats5.err && cp $O/stats5/p_kernel_stats.csv $O/gll_cfg5_kernel_stats.csv
step plain bench lines
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --fp-mode exact --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_exact.json 2>/dev/null
step full-size cpu baseline
python3 bench.py --steps 3 --warmup 1 --cpu-sample-stride 1 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
json.dump({'_note': 'the reference CPU path on ALL 10,077,696 targets of the metric workload (no extrapolation): bench.py --cpu-sample-stride 1', 'cpu_baseline': d['cpu_baseline'], 'parity_vs_cpu_sample': d['parity_vs_cpu_sample'], 'parity_detail': d['parity_detail'], 'gpu_value': d['value'], 'speedup_vs_cpu_baseline': d['speedup_vs_cpu_baseline']}, open('$O/cpu_full.json', 'w'), indent=1)"
step strong-scaling projection
python3 tools/strong_projection.py $O/strong_projection.json > $O/strong_projection.log 2>&1
python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard0.json 2>/dev/null
python3 bench.py --workload cfg4 --cfg4-shard 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard7.json 2>/dev/null
python3 bench.py --workload cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.json 2>/dev/null
step rccl world 1 / rehearsal
MM_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/rccl.err
MM_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_rehearsal_2ranks_strong.json 2> $O/rehearse.err
MM_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rehearsal_2ranks_cfg5.json 2>> $O/rehearse.err
ls $O

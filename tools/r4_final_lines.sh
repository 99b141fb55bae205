#!/bin/bash
# Round 4 (GPU box): the plain bench lines of tools/refresh_profiles.sh once more (after a change that touches only what
# the lines SAY, e.g. a kernel's name), into gpurun_out/refresh/ beside the rest of that script's output.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/refresh; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
cp $O/stats/p_kernel_stats.csv $O/kernel_stats.csv
python3 tools/step_timeline.py $O/stats > $O/step_timeline.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --fp-mode exact --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_exact.json 2>/dev/null
python3 tools/strong_projection.py $O/strong_projection.json > $O/strong_projection.log 2>&1
python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard0.json 2>/dev/null
python3 bench.py --workload cfg4 --cfg4-shard 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_shard7.json 2>/dev/null
python3 bench.py --workload cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.json 2>/dev/null
MM_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/rccl.err
MM_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_rehearsal_2ranks_strong.json 2> $O/rehearse.err
tail -2 $O/strong_projection.log

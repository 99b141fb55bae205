#!/usr/bin/env python3
"""Benchmark of the mesh-to-mesh interpolation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload metric|cfg2|cfg3|cfg4]

With N > 1 and no WORLD_SIZE in the environment this process is only a launcher: it starts N
fresh ranks (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ...``) BEFORE anything touches a GPU, passes their output through and exits with their
code (non-zero when the node has fewer than N devices).  Started by ``torch.distributed.run``
itself (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank: one process per GPU, RCCL over xGMI.

A "step" is one pass of the whole hot path (reference scripts/cli.py:62-100) over one batch of
synthetic targets with every input already resident in HBM: element centroids -> search-grid
build -> k nearest centroids -> hex8 Newton location -> weighted gather (-> one RCCL all-gather of
the interpolated field when N > 1).

Workloads:
  metric  BASELINE.json's metric configuration: 10M -> 10M nodes (216^3 jittered hex meshes),
          1 scalar field, k = 20.  Weak scaling: every rank interpolates its own 10M-node target
          mesh (jitter seed 7 + rank) from the replicated source mesh.
  cfg3    the same with the 3-component vector field;  cfg2: 1M -> 1M.
  cfg4    100.5M-node target mesh (465^3, seed 7) cut into 8 contiguous shards
          (``shard_bounds(465^3, 8, s)``, ~12.57M targets = a 58-plane slab each) over the replicated
          216^3 source.  Rank r interpolates shard r (8 ranks = the whole mesh; fewer ranks = the
          first N shards; 1 rank: shard ``--cfg4-shard``, default 0), so per-GPU work is fixed.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     HBM roofline of the dominant single kernel (algorithmic bytes / its measured duration),
  "stages":       the same accounting for every stage, with the bytes the implementation really moves
                  ("actual_bytes": HBM-unique, narrowed dtypes) beside SURVEY section 8(d)'s figure,
  "gather_reapply": the stand-alone A9 gather on the operator of this very run (reference
                  interpolator.py:724-740: operator cached, only the gather re-runs per field),
  "allgather_ms": one blocking all-gather of the field, timed by itself (N > 1),
  "cpu_baseline": the reference's CPU path timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from multimesh_amd import synth  # noqa: E402
from multimesh_amd.distributed import shard_bounds  # noqa: E402
from multimesh_amd.helpers import STAGES  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); a device-to-device copy reaches ~5 TB/s (roofline.measured_copy_GBps)
CFG4_SHARDS = 8
LAZY_K = 8              # candidates the kNN stage delivers up front (mm_set_lazy_lists)


def algorithmic_bytes(n_targets, n_elem, n_nodes, k, ncomp):
    """algorithmic bytes per launch of each stage (reference dtypes: int64 ids, fp64).

    knn_cell / locate_pass0 are single kernels inside the knn_query / locate stages and are
    priced with the whole stage's algorithmic bytes (they do the stage's work; the other
    kernels of the stage are its bookkeeping)."""
    knn = n_targets * (24 + 8 * k) + n_elem * 24
    loc = n_targets * (24 + 8 * k + 64 + 192 + 128)
    return {
        "centroid": n_elem * (8 * 8 + 8 * 24 + 24),
        "knn_build": n_elem * 24 * 2,                       # read centroids, write them cell-sorted
        "knn_query": knn,
        "locate": loc,
        "gather": n_targets * (128 + 72 * ncomp),
        "knn_cell": knn,
        "locate_pass0": loc,
    }


def actual_bytes(n_targets, n_elem, n_nodes, k, ncomp, fused_gather):
    """Bytes the implementation has to move through HBM at least once per launch:
    every array counted ONCE however often the kernels re-read it through L2 (node coordinates are
    shared by 8 elements, gathered field values by 8 targets), in the dtypes actually stored
    (int32 candidate rows of LAZY_K entries, 32-byte sorted records)."""
    kq = min(k, LAZY_K)
    knn_cell = n_targets * (32 + 4 * kq) + n_elem * 32        # target records in, int32 rows out, source records once
    loc = n_targets * (24 + 4 * kq) + n_elem * 64 + n_nodes * 24   # points + candidate rows, mesh once
    loc += n_targets * 8 * ncomp + n_nodes * 8 * ncomp if fused_gather else n_targets * 128
    return {
        "centroid": n_elem * 64 + n_nodes * 24 + n_elem * 24,
        "knn_build": n_elem * (24 + 8 + 8 + 24 + 32),        # centroids twice, {cell, rank} out and in, records out
        "knn_query": knn_cell + n_targets * (24 + 8 + 8 + 24 + 32),   # + the counting sort of the targets
        "locate": loc,
        "gather": n_targets * (128 + 8 * ncomp) + n_nodes * 8 * ncomp,
        "knn_cell": knn_cell,
        "locate_pass0": loc,
    }


#: stages that are ONE kernel launch each (candidates for the "dominant kernel" roofline)
SINGLE_KERNEL_STAGES = ("centroid", "knn_cell", "locate_pass0", "gather")
KERNEL_OF_STAGE = {"centroid": "centroid_bbox_kernel", "knn_cell": "kNN tile kernel (see config.knn_kernel)",
                   "locate_pass0": "locate_pass_kernel<true, int> (first pass)", "gather": "gather8_kernel<true>"}


def measured_traffic(kernel_stage):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json,
    FETCH_SIZE doubled for wide streaming reads as MI355X_MICROARCH.md prescribes); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
        return data.get(kernel_stage, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(pa, ca, pb, fields, k, sample_stride):
    """The reference CPU path (cKDTree + compiled reference C + NumPy) on a bounded sample."""
    from oracle import oracle as O
    from scipy.spatial import cKDTree

    use_ref = O.have_reference()
    t0 = time.perf_counter()
    cen = O.ref_centroid(ca, pa) if use_ref else O.centroid(ca, pa)
    t_cen = time.perf_counter() - t0
    t0 = time.perf_counter()
    tree = cKDTree(cen, balanced_tree=False)                 # reference scripts/cli.py:66
    t_build = time.perf_counter() - t0
    sample = np.ascontiguousarray(pb[::sample_stride])
    t0 = time.perf_counter()
    _, nn = tree.query(sample, k=k)                          # no workers= -> 1 core, as the reference
    t_query = time.perf_counter() - t0
    conn = synth.reorder_hex8(ca)
    t0 = time.perf_counter()
    if use_ref:
        enc, w, nf = O.ref_locate_hex8(nn, conn, pa, sample)
    else:
        enc, w, nf = O.locate_hex8(nn, conn, pa, sample)
    t_locate = time.perf_counter() - t0
    t0 = time.perf_counter()
    vals = O.gather_numpy(fields, enc, w)
    t_gather = time.perf_counter() - t0
    per_point = (t_query + t_locate + t_gather) / len(sample)
    t_full = t_cen + t_build + per_point * len(pb)

    # "Best-effort CPU" beside it, so the speed-up is not flattered: NOT the
    # reference's behaviour -- cKDTree.query(workers=-1) and the locate restatement on one thread per
    # core (the C call releases the GIL; the reference's triLinearInterpolator is serial).
    from concurrent.futures import ThreadPoolExecutor

    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    _, nn_mt = tree.query(sample, k=k, workers=-1)
    t_query_mt = time.perf_counter() - t0
    chunks = np.array_split(np.arange(len(sample)), ncores * 4)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncores) as pool:
        parts = list(pool.map(lambda c: O.locate_hex8(nn_mt[c], conn, pa, sample[c]), chunks))
    t_locate_mt = time.perf_counter() - t0
    enc_mt = np.concatenate([p[0] for p in parts])
    same = bool(np.array_equal(enc_mt, enc))
    t_full_mt = t_cen + t_build + (t_query_mt + t_locate_mt + t_gather) / len(sample) * len(pb)
    all_cores = {"value": len(pb) / t_full_mt, "unit": "points/s", "cores": ncores,
                 "note": (f"not the reference's behaviour: cKDTree.query(workers=-1) {t_query_mt:.2f}s + locate "
                          f"restatement on {ncores} threads {t_locate_mt:.2f}s on the same sample; tree build "
                          f"({t_build:.2f}s) and gather stay serial; same node ids as the serial run: {same}")}
    return {
        "value": len(pb) / t_full,
        "unit": "points/s",
        "cores": 1,
        "kind": "reference" if use_ref else "port",
        "sample": (f"EXTRAPOLATED from a 1/{sample_stride} sample: full {len(cen)}-element source (centroid {t_cen:.2f}s + "
                   f"cKDTree build {t_build:.2f}s) + every {sample_stride}th target ({len(sample)} points: query "
                   f"{t_query:.2f}s, locate {t_locate:.2f}s, gather {t_gather:.3f}s), per-point cost extrapolated to "
                   f"{len(pb)} targets; host has {os.cpu_count()} cores, path is single-threaded like the reference"),
        "nfailed": int(nf),
        "all_cores": all_cores,
    }, (sample_stride, enc, w, vals)


def hbm_copy_gbps(torch, dev, nbytes=1 << 31, reps=10):
    """Device-to-device copy rate (read + write bytes / time): what this card's HBM delivers to the
    simplest streaming kernel, reported beside the nominal 8 TB/s the fractions are priced against."""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="metric", choices=sorted(synth.CONFIGS))
    ap.add_argument("--n-src", type=int, default=0, help="override nodes per side of the source mesh")
    ap.add_argument("--n-tgt", type=int, default=0, help="override nodes per side of the target mesh")
    ap.add_argument("--ncomp", type=int, default=0)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--cfg4-shard", type=int, default=0, help="cfg4 at one rank: which of the 8 shards")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-stride", type=int, default=0)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------
# launcher: N > 1 without a torchrun environment
# ------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Start ``args.gpus`` ranks as CHILD processes and return their exit code.  Nothing in this
    process initialises the GPU (``torch.cuda.device_count()`` only counts devices), and nothing is
    exec'ed over it: the ranks are fresh interpreters started by torch.distributed.run."""
    import torch

    rehearse = os.environ.get("MM_BENCH_REHEARSE") == "1"    # ranks share GPU 0, gloo carries the collectives
    have = torch.cuda.device_count()
    if have < args.gpus and not rehearse:
        print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    rehearse = os.environ.get("MM_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # MM_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-gather) even at world size 1
    use_dist = world > 1 or (os.environ.get("MM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if use_dist:
            dist.barrier()

    def all_gather(t_all, t_out, async_op=False):
        """The one collective of the path: equal, padded blocks -> [world*chunk, C]."""
        if rehearse:     # gloo has no device all-gather: staged through the host (control flow only)
            parts = [torch.empty(t_out.shape, dtype=t_out.dtype) for _ in range(world)]
            dist.all_gather(parts, t_out.cpu())
            t_all.copy_(torch.cat(parts))
            return None
        return dist.all_gather_into_tensor(t_all, t_out, async_op=async_op)

    cfg = dict(synth.CONFIGS[args.workload])
    n_src = args.n_src or cfg["n_src"]
    n_tgt = args.n_tgt or cfg["n_tgt"]
    ncomp = args.ncomp or cfg["ncomp"]
    k = args.k

    # ---- synthetic inputs (host), then resident in HBM before any timing ----
    pa, ca = synth.hex_mesh(n_src, seed=1)
    sharded = args.workload == "cfg4"
    if sharded:
        # rank r owns shard r of the 8 the 100.5M-node target mesh is cut into (never a cube of its own)
        shard = rank if world > 1 else args.cfg4_shard
        if not 0 <= shard < CFG4_SHARDS or world > CFG4_SHARDS:
            if rank == 0:
                print("bench.py: cfg4 has 8 shards", file=sys.stderr)
            return 2
        lo, hi = shard_bounds(n_tgt ** 3, CFG4_SHARDS, shard)
        def tgt_rows(r, a, b):
            """rows [a, b) of rank r's block of targets"""
            base = shard_bounds(n_tgt ** 3, CFG4_SHARDS, r if world > 1 else args.cfg4_shard)[0]
            return synth.hex_mesh_rows(n_tgt, base + a, base + b, seed=7)

        pb = synth.hex_mesh_rows(n_tgt, lo, hi, seed=7)
        chunk = -(-n_tgt ** 3 // CFG4_SHARDS)          # ceil: the last shard is 7 rows short, blocks are padded
        target_desc = (f"shard {shard} of {CFG4_SHARDS} of the {n_tgt}^3 = {n_tgt ** 3}-node target mesh (rows {lo}..{hi}: "
                       f"a {(hi - 1) // n_tgt ** 2 - lo // n_tgt ** 2 + 1}-plane slab), rank r <-> shard r")
    else:
        def tgt_rows(r, a, b):
            return synth.hex_mesh_rows(n_tgt, a, b, seed=7 + r)

        pb, _ = synth.hex_mesh(n_tgt, seed=7 + rank)
        chunk = pb.shape[0]
        target_desc = f"every rank its own {n_tgt}^3-node target mesh (jitter seed 7 + rank)"
    fields = synth.vector_field(pa)[:ncomp]
    t_nodes = torch.from_numpy(pa).to(dev)
    t_conn = torch.from_numpy(ca).to(dev)
    t_pts = torch.from_numpy(pb).to(dev)
    t_fields = torch.from_numpy(fields).to(dev)
    n_local = pb.shape[0]
    # Two sets of output buffers: with more than one rank the all-gather of step s runs on RCCL's own
    # stream while step s+1 computes into the other set (the gather only reads its own step's block).
    nbuf = 2 if use_dist else 1
    t_outs = [torch.zeros((chunk, ncomp), dtype=torch.float64, device=dev) for _ in range(nbuf)]
    t_alls = [torch.empty((world * chunk, ncomp), dtype=torch.float64, device=dev) if use_dist else None
              for _ in range(nbuf)]
    t_out, t_all = t_outs[0], t_alls[0]
    pending = [None] * nbuf
    step_no = [0]

    from multimesh_amd.device import Context

    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(dev_index, stream=stream)
    ctx.set_profiling(True)

    stage_ms = {s: 0.0 for s in STAGES}
    nfailed_total = 0

    def step(record):
        nonlocal nfailed_total, t_out, t_all
        b = step_no[0] % nbuf
        step_no[0] += 1
        if pending[b] is not None:
            pending[b].wait()          # the gather that last read this buffer set (two steps ago)
            pending[b] = None
        t_out, t_all = t_outs[b], t_alls[b]
        _, nf = ctx.interpolate_hex8(t_nodes, t_conn, t_pts, t_fields, nelem_to_search=k, out=t_out[:n_local])
        if use_dist:
            # asynchronous: the gather overlaps the next step's kernels
            pending[b] = all_gather(t_all, t_out, async_op=True)
        if record:
            nfailed_total += nf
            for s, v in ctx.last_timings().items():
                stage_ms[s] += v

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step(False)
    drain()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    drain()                            # every gather of the timed steps has completed ...
    torch.cuda.synchronize()           # ... and so has everything else on the device
    barrier()
    elapsed = time.perf_counter() - t0

    def all_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = all_max(elapsed)
    if use_dist:
        nf_t = torch.tensor([nfailed_total], dtype=torch.int64, device="cpu" if rehearse else dev)
        dist.all_reduce(nf_t)
        nfailed_total = int(nf_t.item())

    # ---- the collective by itself: one extra step with a BLOCKING all-gather, timed on the stream ----
    allgather = None
    if use_dist:
        ms = []
        for _ in range(3):
            torch.cuda.synchronize()
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            all_gather(t_all, t_out)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        ag_ms = all_max(sorted(ms)[1])
        nbytes_in = (world - 1) * chunk * ncomp * 8
        allgather = {"ms": round(ag_ms, 4), "bytes_received_per_gpu": nbytes_in,
                     "GBps_per_gpu": round(nbytes_in / (ag_ms * 1e-3) / 1e9, 1) if ag_ms > 0 else None,
                     "note": "median of 3 blocking all_gather_into_tensor calls (max over ranks); in the timed steps "
                             "the gather is asynchronous and overlaps the next step's kernels"}
        # the gathered field = what one rank computes alone: rank 0 re-interpolates a sample of the LAST
        # rank's block on its own GPU and compares bit for bit
        if rank == 0:
            last = world - 1
            n_last = (shard_bounds(n_tgt ** 3, CFG4_SHARDS, last)[1] - shard_bounds(n_tgt ** 3, CFG4_SHARDS, last)[0]
                      if sharded else chunk)
            ns = min(200_000, n_last)
            sample = torch.from_numpy(tgt_rows(last, n_last - ns, n_last)).to(dev)
            ref, _ = ctx.interpolate_hex8(t_nodes, t_conn, sample, t_fields, nelem_to_search=k)
            ref = torch.from_numpy(ref.numpy()).to(dev)
            got = t_all[last * chunk + n_last - ns:last * chunk + n_last]
            allgather["gathered_equals_single_rank_on_sample"] = bool(torch.equal(got, ref))
            allgather["sample"] = f"last {ns} targets of rank {last}'s block"
            assert torch.equal(t_all[:n_local], t_out[:n_local]), "all-gather did not return this rank's block"

    rc = 0
    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed / steps * 1e3
        total_targets = n_local * world if not sharded else sum(
            shard_bounds(n_tgt ** 3, CFG4_SHARDS, r)[1] - shard_bounds(n_tgt ** 3, CFG4_SHARDS, r)[0]
            for r in (range(world) if world > 1 else [args.cfg4_shard]))
        value = total_targets * steps / elapsed
        n_elem, n_nodes = ca.shape[0], pa.shape[0]
        fused = stage_ms["gather"] == 0.0
        abytes = algorithmic_bytes(n_local, n_elem, n_nodes, k, ncomp)
        rbytes = actual_bytes(n_local, n_elem, n_nodes, k, ncomp, fused)
        stages = {}
        if fused:
            # values-only call: the weighted sum (A9) is formed inside the locate kernels at the
            # point of acceptance, so the locate stage carries the gather's algorithmic bytes too
            abytes["locate"] += abytes["gather"]
            abytes["locate_pass0"] += abytes["gather"]

        def account(ms, alg, act):
            gb_alg = alg / (ms * 1e-3) / 1e9
            gb_act = act / (ms * 1e-3) / 1e9
            d = {"ms": round(ms, 4), "algorithmic_bytes": alg, "actual_bytes": act,
                 "achieved_GBps": round(gb_alg, 1), "actual_GBps": round(gb_act, 1),
                 "frac_actual": round(gb_act / HBM_PEAK_GBPS, 4)}
            if gb_alg <= HBM_PEAK_GBPS:
                d["frac"] = round(gb_alg / HBM_PEAK_GBPS, 4)
            else:
                # SURVEY §8(d) prices re-reads that are L2 hits (a node's coordinates once per element,
                # a field value once per target): not a fraction of HBM peak
                d["frac"] = None
                d["note"] = "section 8(d) bytes include L2-resident re-reads; see frac_actual"
            return d

        for s in STAGES:
            ms = stage_ms[s] / steps
            if ms == 0.0:
                stages[s] = {"ms": 0.0, "fused_into": "locate"}
                continue
            stages[s] = account(ms, abytes[s], rbytes[s])
        dominant = max(SINGLE_KERNEL_STAGES, key=lambda s: stage_ms[s])
        d = stages[dominant]
        roofline = {"bound": "hbm", "kernel": KERNEL_OF_STAGE[dominant], "stage": dominant,
                    "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": d["frac"], "ms": d["ms"], "algorithmic_bytes": abytes[dominant],
                    "actual_bytes": rbytes[dominant], "frac_actual_bytes": d["frac_actual"],
                    "traffic": measured_traffic(dominant),
                    "timing": "hipEvents on the context's stream around this kernel's launches, averaged over the timed steps"}
        e2e_bytes = 952 + 72 * (ncomp - 1)          # 184 + 568 + 200 at C = 1, k = 20 (+72 per component)
        e2e_bytes += 8 * (k - 20) * 2
        line = {
            "metric": "interpolated points/sec, 10M->10M 3D mesh, 1 scalar field",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: hex8 3D {n_nodes} source nodes -> {n_local} targets per GPU (n_src={n_src}, "
                                   f"n_tgt={n_tgt} per side, jittered unit cube), {ncomp} field component(s), k={k}; targets: "
                                   + target_desc,
                       "source_nodes": n_nodes, "source_elements": n_elem, "targets_per_gpu": n_local,
                       "targets_total": total_targets,
                       "candidate_lists": f"evaluated lazily: the {LAZY_K} nearest centroids up front, the full "
                                          f"k={k} list only for targets that exhaust them; every output is "
                                          "bit-identical to the eager evaluation (mm_set_lazy_lists(0))",
                       "knn_kernel": os.environ.get("MM_KNN_KERNEL", "default"),
                       "parallelism": (f"targets sharded x{world}, source replicated, 1 all-gather per step (asynchronous: "
                                       "overlaps the next step's kernels)" + (" [REHEARSAL: ranks share GPU 0, gloo]" if rehearse else ""))
                       if world > 1 else "single GPU"},
            "nfailed": nfailed_total,
            "roofline": roofline,
            "roofline_end_to_end": {"algorithmic_bytes_per_target": e2e_bytes,
                                    "achieved_GBps_per_gpu": round(e2e_bytes * n_local / (ms_per_step * 1e-3) / 1e9, 1),
                                    "frac": round(e2e_bytes * n_local / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "stages": stages,
        }
        if allgather:
            line["allgather_ms"] = allgather["ms"]
            line["allgather"] = allgather
            if allgather.get("gathered_equals_single_rank_on_sample") is False:
                rc = 1
        roofline["measured_copy_GBps"] = round(hbm_copy_gbps(torch, dev), 1)

        # ---- stand-alone A9: the operator of THIS run re-applied to the field (the stored_array pattern) ----
        vals_op, enc, w, _ = ctx.interpolate_hex8(t_nodes, t_conn, t_pts, t_fields, nelem_to_search=k, want_operator=True)
        reps = max(args.steps, 5)
        g_out = ctx.gather(t_fields, enc, w)
        ctx.gather(t_fields, enc, w).free()
        g_ms = 0.0
        for _ in range(reps):
            o = ctx.gather(t_fields, enc, w)
            g_ms += ctx.last_timings()["gather"]
            o.free()
        g_ms /= reps
        ga = account(g_ms, n_local * (128 + 72 * ncomp), n_local * (128 + 8 * ncomp) + n_nodes * 8 * ncomp)
        ga["kernel"] = "gather8_kernel<true>"
        ga["equals_fused_values"] = bool(np.array_equal(g_out.numpy(), t_out[:n_local].cpu().numpy()))
        ga["traffic"] = measured_traffic("gather")
        ga["note"] = ("mm_gather on the (node ids, weights) rows of this run's targets: section 8(d) prices 64 B of gathered "
                      "field values per target and component, which are L2 hits (each node serves ~8 targets); actual_bytes "
                      "counts ids + weights + output + the field once")
        line["gather_reapply"] = ga
        stages["gather_reapply"] = ga
        del vals_op, enc, w, g_out

        if world == 1:
            # PCIe-inclusive rate (never `value`): the same step fed from HOST arrays, i.e. what the
            # legacy host-pointer boundary costs: H2D of mesh + targets + field, D2H of the result
            out_h = np.zeros((n_local, ncomp))          # the reference's callers pass zero-initialised outputs (cli.py:77-78)
            ctx.interpolate_hex8_host(pa, ca, pb, fields, nelem_to_search=k, out=out_h)      # warm-up: device copies allocated
            t0 = time.perf_counter()
            vals_h, _ = ctx.interpolate_hex8_host(pa, ca, pb, fields, nelem_to_search=k, out=out_h)
            t_host = time.perf_counter() - t0
            nbytes = pa.nbytes + ca.nbytes + pb.nbytes + fields.nbytes + out_h.nbytes
            line["host_arrays"] = {"ms": t_host * 1e3, "points_per_s": n_local / t_host, "bytes_over_pcie": nbytes,
                                   "link_GBps_if_transfers_alone": round(nbytes / t_host / 1e9, 1),
                                   "equals_resident_run": bool(np.array_equal(vals_h, t_out[:n_local].cpu().numpy())),
                                   "note": "mm_interpolate_hex8_host: one step fed from pageable NumPy arrays (H2D on a second "
                                           "stream beside the kernels + D2H), device copies cached in the context; not `value`"}
            del vals_h
        if world == 1 and not args.no_cpu_baseline:
            stride = args.cpu_sample_stride or max(1, n_local // 2_500_000)   # ~12 s of single-threaded CPU work
            base, (stride, enc_c, w_c, vals_c) = cpu_baseline(pa, ca, pb, fields, k, stride)
            line["cpu_baseline"] = base
            line["speedup_vs_cpu_baseline"] = value / base["value"]
            # parity gate on the sample the CPU just computed
            got = t_out[:n_local].cpu().numpy()[::stride]
            line["parity_vs_cpu_sample"] = bool(np.array_equal(got, vals_c))
            if not line["parity_vs_cpu_sample"]:
                rc = 1
        print(json.dumps(line), flush=True)

    ctx.close()
    if use_dist:
        barrier()
        dist.destroy_process_group()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())

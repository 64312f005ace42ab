#!/usr/bin/env python3
"""bench.py - headline measurement of the MI355X kNN / ICP hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`: one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): synthetic 1 000 000-point target uniform in
[0,1024)^3 (seed 20260101), replicated on every GPU; 1 000 000 queries per GPU from the same cube
(seed 20260102 + rank), k = 16.  One step = one pclmi_knn_dev call over the rank's 1M queries:
query ordering (Morton-cell counting sort) + the traversal kernel, inputs and outputs resident in
HBM.  Queries are sharded with no data-path collective, so scaling is "weak".

The same line carries
  roofline      HIP-event time of the traversal kernel (pclmi_timing_*, events recorded on the
                stream the kernel runs on) against the algorithmic bytes of SURVEY 8d:
                16 B query + 8 B x k results per query + 16 B per target point = 160 MB / launch;
  cpu_baseline  the CPU oracle's FLANN-style kd-tree (oracle/, OpenMP over queries) on the host
                cores of this box, same inputs, rank 0 at N = 1 only;
  icp           ICP iterations / s on BASELINE configs[3] (1M source vs 1M target, 30 iterations,
                source sharded over the ranks, one all-reduce of 24 fp64 sums per iteration; the
                loop is device-resident: step kernel, all-reduce and solve kernel are enqueued on
                one stream, the host reads the result once).
torch is used for device memory, streams and torch.distributed only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TARGET = 1_000_000
N_QUERY = 1_000_000
K = 16
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SEED_T, SEED_Q = 20260101, 20260102


def uniform_cloud(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3), dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def surface_cloud(n, seed):
    """SURVEY 8d surface-like variant: points on 64 random planes inside [0,1024)^3 + sigma=0.5 noise."""
    rng = np.random.default_rng(seed)
    planes = 64
    origin = rng.random((planes, 3)) * 1024
    nrm = rng.normal(size=(planes, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    u = np.cross(nrm, np.array([0.3, 0.5, 0.8]))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(nrm, u)
    pid = rng.integers(0, planes, n)
    ab = (rng.random((n, 2)) - 0.5) * 600
    pts = origin[pid] + ab[:, :1] * u[pid] + ab[:, 1:] * v[pid] + rng.normal(0, 0.5, (n, 3))
    return pts.astype(np.float32)


def rigid(angle_deg, axis, trans):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    th = np.deg2rad(angle_deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = trans
    return T


def pmc_traffic(which="knn16"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*<which>*_pmc.json, produced by tools/pmc_summary.py / tools/pmc_icp.sh; FETCH_SIZE doubled per
    the gfx950 note of the micro-architecture guide); None when no summary is committed."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_pmc.json" % which))):
        try:
            d = json.load(open(p))
            best = d.get("hbm_bytes_per_launch")
            if best is None and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                best = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        except Exception:
            pass
    return best


def pmc_valu_issue(which, kernel_ms, n_simd=1024, clock_ghz=2.4):
    """Fraction of the launch's VALU issue slots the kernel used: SQ_INSTS_VALU of the committed PMC summary x 4
    cycles per wave64 instruction / (SIMDs x kernel cycles at the 2.4 GHz peak engine clock).  None without a summary."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_pmc.json" % which))):
        try:
            v = json.load(open(p)).get("SQ_INSTS_VALU")
            if v:
                return 4.0 * v / (n_simd * kernel_ms * 1e-3 * clock_ghz * 1e9)
        except Exception:
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--icp-iters", type=int, default=30)
    ap.add_argument("--no-icp", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pcl_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a one-GPU box (never used by the driver): every rank on device 0,
    # collectives over gloo instead of RCCL (RCCL refuses two ranks on one GPU).
    rehearsal = os.environ.get("PCLMI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available() or capi.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: libpclmi has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    capi.load_library().pclmi_set_device(local)

    verbose = rehearsal or os.environ.get("PCLMI_BENCH_VERBOSE") == "1"
    if verbose:  # a stuck rank says where: stack dump after PCLMI_BENCH_WATCHDOG seconds (default 150), then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PCLMI_BENCH_WATCHDOG", "150")), exit=True)

    def log(msg):
        if verbose:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- kNN headline -------------------------------------------------------------------------
    log("process group up (world %d), building the target tree" % world)
    target = uniform_cloud(N_TARGET, SEED_T, 0, 1024)
    t0 = time.time()
    tree = capi.Tree(target)  # replicated on every GPU
    build_s = time.time() - t0
    q_host = uniform_cloud(N_QUERY, SEED_Q + rank, 0, 1024)
    q = torch.from_numpy(np.concatenate([q_host, np.ones((N_QUERY, 1), np.float32)], axis=1)).to(dev)  # PointXYZ rows
    idx = torch.empty((N_QUERY, K), dtype=torch.int32, device=dev)
    d2 = torch.empty((N_QUERY, K), dtype=torch.float32, device=dev)
    # an explicit (non-null) torch stream: the library enqueues on it, torch and RCCL order against it
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    torch.cuda.synchronize()

    def step():
        tree.knn_dev(q.data_ptr(), N_QUERY, 16, K, idx.data_ptr(), d2.data_ptr(), stream)

    log("tree built in %.2f s, warm-up" % build_s)
    for _ in range(args.warmup):
        step()
    # HIP events around the dominant kernel only, live over the timed region (two event records per step)
    capi.timing_enable(2 << capi.TIMER_KNN)
    capi.timing_read(capi.TIMER_KNN, reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    log("timed region done")
    elapsed = max_over_ranks(time.perf_counter() - t0)
    k_ms, k_n = capi.timing_read(capi.TIMER_KNN, reset=True)
    # the ordering pass (counting sort by Morton cell) is timed in a short pass of its own, outside the timed region
    capi.timing_enable(2 << capi.TIMER_ORDER)
    capi.timing_read(capi.TIMER_ORDER, reset=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    o_ms, o_n = capi.timing_read(capi.TIMER_ORDER, reset=True)
    capi.timing_enable(False)
    value = world * N_QUERY * args.steps / elapsed
    kern_ms = k_ms / max(k_n, 1)
    algo_bytes = 16 * N_QUERY + 8 * K * N_QUERY + 16 * N_TARGET
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    traffic = pmc_traffic()

    # size-independent sanity of the timed outputs (full parity lives in tests/): ascending rows,
    # first neighbour recomputed from the coordinates
    torch.cuda.synchronize()
    d2h = d2[:4096].cpu().numpy()
    idh = idx[:4096].cpu().numpy()
    assert np.all(np.diff(d2h, axis=1) >= 0) and idh.min() >= 0 and idh.max() < N_TARGET
    diff = q_host[:4096] - target[idh[:, 0]]
    rec = ((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).astype(np.float32)
    assert np.array_equal(rec, d2h[:, 0]), "kNN output failed the distance re-computation check"

    out = {
        "metric": "kNN queries/sec (k=16, 1M-pt target)",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "configs[1]: 1M-point uniform target, 1M queries per GPU, nearestKSearch k=16",
                   "n_target": N_TARGET, "n_queries_per_gpu": N_QUERY, "k": K,
                   "parallelism": "target replicated, queries sharded x%d" % world,
                   "tree_build_s": round(build_s, 3)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "knn_union_ix_kernel<16,40,1>", "kernel_ms": kern_ms, "launches": int(k_n),
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "order_ms": o_ms / max(o_n, 1),
                     # the traversal is VALU-issue-bound, not HBM-bound: the binding resource beside
                     # the nominal HBM fraction
                     "valu_issue_frac": pmc_valu_issue("knn16", kern_ms)},
    }

    # ---- ICP iterations / s (configs[3]) -------------------------------------------------------
    if not args.no_icp:
        log("kNN done: %.3e queries/s; ICP next" % value)
        out["icp"] = bench_icp(args, capi, torch, dist, world, rank, dev, barrier, max_over_ranks, log)

    # ---- CPU baseline (rank 0, N = 1 only) -----------------------------------------------------
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(target, q_host, idx, d2)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_icp(args, capi, torch, dist, world, rank, dev, barrier, max_over_ranks, log=lambda m: None):
    n = 1_000_000
    tgt = surface_cloud(n, SEED_T + 10)
    # rigid motion to recover: 1 degree about (1,1,1) through the cloud centre + (2,-1,0.5)
    T0 = rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
    c = np.array([512.0, 512.0, 512.0])
    T0[:3, 3] += c - T0[:3, :3] @ c
    rng = np.random.default_rng(SEED_Q + 10)
    src_all = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
    Ti = np.linalg.inv(T0)
    src_all = (src_all.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    per = (n + world - 1) // world
    src = src_all[rank * per:(rank + 1) * per]
    tree = capi.Tree(tgt)
    sums = torch.zeros(capi.ICP_NSUMS, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    thr = 25.0
    iters = args.icp_iters

    def run():
        """One align of `iters` iterations, device-resident: step kernel -> (all-reduce) -> solve kernel,
        enqueued back to back on one stream; the host reads the state once at the end."""
        ctx = capi.IcpShard(tree, src)  # upload + Morton ordering of the shard: setup, not an iteration
        barrier()
        t0 = time.perf_counter()
        if world == 1:
            ctx.run_dev(thr, iters, stream)
        else:
            for _ in range(iters):
                ctx.step_dev(thr, sums.data_ptr(), stream)
                dist.all_reduce(sums)  # the ONE exchange of an iteration: 24 fp64 sums over xGMI (RCCL)
                ctx.solve_dev(sums.data_ptr(), stream)
        rc, T_final, st = ctx.state()  # synchronises the stream
        barrier()
        dt = time.perf_counter() - t0
        assert rc == 0 and st.iterations == iters, (rc, st.iterations)
        ctx.end(want_aligned=False)
        return dt, T_final, st.last_correspondences

    log("ICP clouds and tree ready, warm-up align")
    run()  # warm-up pass (allocations, first-touch, RCCL channel set-up)
    log("ICP warm-up done, timed align")
    dt, T_final, npairs = run()  # the timed align carries no instrumentation (events cost ~7 us per iteration)
    elapsed = max_over_ranks(dt)
    log("ICP timed align done, instrumented align")
    capi.timing_enable(2 << capi.TIMER_ICP_STEP)  # a third align, with HIP events around every step kernel
    capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
    run()
    s_ms, s_n = capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
    capi.timing_enable(False)
    err = float(np.abs(T_final - T0).max())
    step_ms = s_ms / max(s_n, 1)
    algo = 52 * len(src) + 16 * n  # 52 B / source point + first touch of the target
    return {"metric": "ICP iterations/sec (1M source vs 1M target, point-to-point, RANSAC off)",
            "value": iters / elapsed, "unit": "iterations/s", "iterations": iters, "scaling": "strong",
            "ms_per_iteration": elapsed / iters * 1e3, "n_source_total": n, "n_source_per_gpu": len(src),
            "last_pairs_global": npairs, "max_abs_err_vs_true_transform": err,
            "step_kernel_ms": step_ms, "step_kernel_ms_from": "a separate align with HIP events around each step",
            "roofline": {"bound": "hbm", "achieved": algo / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": algo / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("icp_step") if world == 1 else None,
                         "kernel": "icp_step_kernel<true>", "algorithmic_bytes_per_launch": algo}}


def cpu_baseline(target, q_host, idx, d2):
    """Times the oracle's FLANN-style kd-tree (CPU restatement, kind = "port") on this box's cores."""
    from oracle import pcl_oracle as orc
    cores = min(16, len(os.sched_getaffinity(0)))
    tree = orc.Tree(target)
    reps = 6  # ~10 core-seconds of CPU work on 16 cores
    nq = len(q_host)
    orc.Tree.knn(tree, q_host[:20000], K, nthreads=cores)  # thread-pool warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        oi, od = tree.knn(q_host, K, nthreads=cores)
    dt = time.perf_counter() - t0
    # the timed GPU outputs equal the CPU baseline's (bit-exact) on a sample
    gi, gd = idx[:20000].cpu().numpy(), d2[:20000].cpu().numpy()
    parity = bool(np.array_equal(gi, oi[:20000]) and np.array_equal(gd.view(np.int32), od[:20000].view(np.int32)))
    t1 = time.perf_counter()
    tree.knn(q_host[:200_000], K, nthreads=1)
    dt1 = time.perf_counter() - t1
    return {"value": reps * nq / dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": "all %d queries x %d passes, FLANN-style kd-tree (leaf 15), OpenMP over queries" % (nq, reps),
            "single_thread_value": 200_000 / dt1, "gpu_matches_cpu_bitexact_on_20000": parity}


if __name__ == "__main__":
    main()

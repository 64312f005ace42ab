"""Steady-state ICP iteration on one GPU (development helper): configs[3] clouds, converged transform, (30 - 10
iterations) / 20 of whole run_dev calls, for the shards of an N-way split (WORLDS=1,8)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from pcl_amd import capi
tgt, src_all, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
for world in [int(v) for v in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world)
    ctx.run_dev(25.0, 30)
    rc, T, st = ctx.state()
    ctx.end(want_aligned=False)
    best = {}
    for rep in range(3):
        for it in (10, 30):
            ctx = capi.IcpShard(tree, src_all, guess=T, rank=world // 2, world=world)
            ctx.run_dev(25.0, 2)  # the first iterations from the converged transform: neighbours found, bounds tight
            ctx.state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run_dev(25.0, it)
            ctx.state()
            dt = time.perf_counter() - t0
            ctx.end(want_aligned=False)
            best[it] = min(best.get(it, 1e9), dt)
    print("world %d: %7d points: steady iteration %.2f us" % (world, ctx.shard_size if hasattr(ctx, "shard_size") else 0,
                                                               (best[30] - best[10]) / 20 * 1e6))

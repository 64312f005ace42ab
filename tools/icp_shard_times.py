"""Per-rank compute side of the sharded ICP loop on ONE GPU (development helper): for world sizes 1, 2, 4, 8 (and
1024: a 977-point shard = the fixed cost of an iteration) one shard of bench.py's configs[3] source runs the
device-resident loop (ONE kernel per iteration: step, reduction and solve; no exchange) against the full 1M-point
target.  Gives the compute-only ceiling of the strong scaling that the 8-GPU run adds its exchange to.
SHARD=hilbert (default): the rank's contiguous run of the Hilbert-ordered source (pclmi_icp_begin_sharded);
SHARD=slice: a slice of the unsorted source (round 1's split)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from pcl_amd import capi

n = 1_000_000
iters = 30
tgt = bench.surface_cloud(n, bench.SEED_T + 10)
T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
c = np.array([512.0, 512.0, 512.0])
T0[:3, 3] += c - T0[:3, :3] @ c
rng = np.random.default_rng(bench.SEED_Q + 10)
src_all = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
Ti = np.linalg.inv(T0)
src_all = (src_all.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
tree = capi.Tree(tgt)
hilbert = os.environ.get("SHARD", "hilbert") == "hilbert"
for world in [int(v) for v in os.environ.get("WORLDS", "1,2,4,8,1024").split(",")]:
    per = (n + world - 1) // world
    src = src_all[:per]
    best = None
    for rep in range(3):
        ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world) if hilbert else capi.IcpShard(tree, src)
        torch.cuda.synchronize()
        capi.timing_enable(os.environ.get("NO_EVENTS") != "1")
        capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        t0 = time.perf_counter()
        ctx.run_dev(25.0, iters)
        rc, T, st = ctx.state()
        dt = time.perf_counter() - t0
        s_ms, s_n = capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        capi.timing_enable(False)
        ctx.end(want_aligned=False)
        if best is None or dt < best[0]:
            best = (dt, s_ms / max(s_n, 1))
    print("world %d (%s): %7d source points per rank: %.4f ms / iteration (step kernel incl. tail %.4f ms)" %
          (world, "hilbert run" if hilbert else "slice", per, best[0] / iters * 1e3, best[1]), flush=True)

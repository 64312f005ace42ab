"""Where an ICP iteration's time goes inside its ONE kernel (development helper): device wall-clock stamps of the step
launch - workgroup 0 started, last workgroup arrived, sums ready, solve done - for shards of bench.py's configs[3].
usage: icp_tail_ticks.py [world ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from pcl_amd import capi

n = 1_000_000
tgt = bench.surface_cloud(n, bench.SEED_T + 10)
T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
c = np.array([512.0, 512.0, 512.0])
T0[:3, 3] += c - T0[:3, :3] @ c
rng = np.random.default_rng(bench.SEED_Q + 10)
src_all = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
Ti = np.linalg.inv(T0)
src_all = (src_all.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
tree = capi.Tree(tgt)
lib = capi.load_library()
lib.pclmi_debug_icp_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
for world in [int(v) for v in (sys.argv[1:] or ["1", "8", "1024"])]:
    ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world)
    rows = []
    for it in range(30):
        ctx.run_dev(25.0, 1)
        t = (C.c_uint64 * 4)()
        lib.pclmi_debug_icp_ticks(ctx.h, t)
        rows.append([(t[i] - t[0]) / 100.0 for i in (1, 2, 3)])
    ctx.end(want_aligned=False)
    r = np.array(rows)
    print("world %d (%d points): us from workgroup 0's start to: last arrival / sums ready / solve done" % (world, ctx.shard_size))
    print("   iteration 1: %.1f / %.1f / %.1f" % tuple(r[0]))
    print("   iterations 2-5 mean: %.1f / %.1f / %.1f" % tuple(r[1:5].mean(0)))
    print("   iterations 11-30 mean: %.1f / %.1f / %.1f" % tuple(r[10:].mean(0)))

"""Step-kernel time of every ICP iteration (development helper): usage icp_iter_times.py [world] [iterations]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from pcl_amd import capi

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
src = src_all[:(n + world - 1) // world]
for rep in range(2):
    ctx = capi.IcpShard(tree, src)
    torch.cuda.synchronize()
    capi.timing_enable(True)
    out = []
    for it in range(iters):
        capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        ctx.run_dev(25.0, 1)
        rc, T, st = ctx.state()
        ms, cnt = capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        out.append(ms / max(cnt, 1) * 1e3)
    capi.timing_enable(False)
    ctx.end(want_aligned=False)
print("%s world %d (%d points): step kernel us per iteration:" % (os.environ.get("PCLMI_ICP_WALK", "bottomup"), world, len(src)),
      " ".join("%.0f" % v for v in out))

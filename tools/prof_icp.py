"""A few device-resident ICP iterations on bench.py's configs[3] data, for rocprofv3 (development helper).
usage: prof_icp.py [iterations] [n] [converged]   (converged: start from the true transform - steady iterations only)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from pcl_amd import capi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
tgt = bench.surface_cloud(n, bench.SEED_T + 10)
T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
c = np.array([512.0, 512.0, 512.0])
T0[:3, 3] += c - T0[:3, :3] @ c
rng = np.random.default_rng(bench.SEED_Q + 10)
src = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
Ti = np.linalg.inv(T0)
src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
tree = capi.Tree(tgt)
ctx = capi.IcpShard(tree, src, guess=T0 if len(sys.argv) > 3 and sys.argv[3] == "converged" else None)
ctx.run_dev(25.0, iters)
rc, T, st = ctx.state()
print("iterations", st.iterations, "pairs", st.last_correspondences, "err", float(np.abs(T - T0).max()))
ctx.end(want_aligned=False)

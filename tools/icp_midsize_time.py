"""Default-parameter ICP (RANSAC rejection on) and RANSAC-off ICP on mid-size clouds - the sizes of downsampled depth
frames, where an iteration is launch- and latency-bound (development helper)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from pcl_amd import capi
for n in (397, 10_000, 50_000, 200_000):
    tgt = bench.surface_cloud(n, 7) if n > 1000 else None
    if tgt is None:
        from tests import util
        src, tgt = util.bun0(), util.bun4()
        thr, rth = 0.05, 0.05
    else:
        T0 = bench.rigid(1.0, (1, 1, 1), (1.0, -0.5, 0.25))
        rng = np.random.default_rng(8)
        src = tgt[rng.permutation(n)[: n * 9 // 10]] + rng.normal(0, 0.1, (n * 9 // 10, 3)).astype(np.float32)
        Ti = np.linalg.inv(T0)
        src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        thr, rth = 10.0 * (1e6 / n) ** 0.5, 0.5 * (1e6 / n) ** 0.5
    tree = capi.Tree(tgt)
    for ransac in (0, 1000):
        res = {}
        for iters in (10, 30, 10, 30):
            p = capi.icp_default_params()
            p.max_iterations, p.transformation_epsilon, p.euclidean_fitness_epsilon = iters, 0.0, -1.0
            p.max_correspondence_distance, p.ransac_iterations, p.ransac_outlier_threshold = thr, ransac, rth
            t0 = time.perf_counter()
            rc, T, _, st = capi.icp_align(tree, src, p)
            res[iters] = time.perf_counter() - t0
        print("n = %d, ransac %d: 30-iteration align %.2f ms, iterations 11-30 %.1f us each (rc %d, inliers %d of %d)"
              % (n, ransac, res[30] * 1e3, (res[30] - res[10]) / 20 * 1e6, rc, st.last_inliers, st.last_correspondences), flush=True)

"""RANSAC-on ICP per-iteration time on configs[3]'s clouds (development helper; bench.py reports the same figure)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pcl_amd import capi

tgt, src, T_true = bench.icp_clouds()
tree = capi.Tree(tgt)
res = {}
for iters in (1, 5, 10, 30, 1, 5, 10, 30):
    p = capi.icp_default_params()
    p.max_iterations = iters
    p.transformation_epsilon = 0.0
    p.euclidean_fitness_epsilon = -1.0
    p.max_correspondence_distance = 10.0
    p.ransac_iterations = 1000
    p.ransac_outlier_threshold = 0.5
    t0 = time.perf_counter()
    rc, T, _, st = capi.icp_align(tree, src, p)
    res[iters] = time.perf_counter() - t0
    print("iterations %d: %.2f ms (rc %d, inliers %d)" % (iters, res[iters] * 1e3, rc, st.last_inliers), flush=True)
print("per iteration, 2..5: %.3f ms; 11..30 (converged): %.3f ms" % ((res[5] - res[1]) / 4 * 1e3, (res[30] - res[10]) / 20 * 1e3))

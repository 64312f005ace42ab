"""Whole-call time of pclmi_icp_align on configs[3]'s clouds (host pointers in, transform and aligned cloud out):
what a caller of IterativeClosestPoint::align pays besides the iterations (development helper)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pcl_amd import capi

tgt, src, T_true = bench.icp_clouds()
t0 = time.perf_counter()
tree = capi.Tree(tgt)
print("tree: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
for ransac in (0, 1000):
    for iters in (1, 30, 1, 30):
        p = capi.icp_default_params()
        p.max_iterations = iters
        p.transformation_epsilon = 0.0
        p.euclidean_fitness_epsilon = -1.0
        p.max_correspondence_distance = 10.0
        p.ransac_iterations = ransac
        p.ransac_outlier_threshold = 0.5
        t0 = time.perf_counter()
        rc, T, _, st = capi.icp_align(tree, src, p)
        print("ransac %d, %d iterations: %.2f ms (rc %d)" % (ransac, iters, (time.perf_counter() - t0) * 1e3, rc), flush=True)

"""Two default-parameter (RANSAC on) 30-iteration aligns of configs[3]'s clouds, for rocprofv3 --kernel-trace
(development helper: the timeline of one iteration was read off its kernel trace)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from pcl_amd import capi
tgt, src, T_true = bench.icp_clouds()
tree = capi.Tree(tgt)
p = capi.icp_default_params(); p.max_iterations = 30; p.transformation_epsilon = 0.0; p.euclidean_fitness_epsilon = -1.0
p.max_correspondence_distance = 10.0; p.ransac_iterations = 1000; p.ransac_outlier_threshold = 0.5
for _ in range(2):
    t0 = time.perf_counter(); rc, T, _, st = capi.icp_align(tree, src, p); print((time.perf_counter() - t0) * 1e3, rc)

"""Point-to-plane and RANSAC-on per-iteration times on configs[3]'s clouds (development helper for A/B runs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from pcl_amd import capi
tgt, src, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
tn = capi.normals(tree, tgt, k=16, viewpoint=(512.0, 512.0, 4096.0))[:, :3].copy()
def run(kind, iters):
    p = capi.icp_default_params()
    p.max_iterations, p.transformation_epsilon, p.euclidean_fitness_epsilon = iters, 0.0, -1.0
    p.max_correspondence_distance = 25.0
    p.ransac_iterations = 1000 if kind == "ransac" else 0
    p.ransac_outlier_threshold = 0.5
    t0 = time.perf_counter()
    if kind == "plane":
        rc, T, _, st = capi.icp_align_point_to_plane(tree, src, tn, p)
    else:
        rc, T, _, st = capi.icp_align(tree, src, p)
    return time.perf_counter() - t0
for kind in sys.argv[1:] or ("plane", "svd", "ransac"):
    best = {}
    for rep in range(6):
        for it in (10, 30):
            best[it] = min(best.get(it, 1e9), run(kind, it))
    print("%s: %.4f ms per iteration (iterations 11-30)" % (kind, (best[30] - best[10]) / 20 * 1e3), flush=True)

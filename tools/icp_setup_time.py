"""What a whole pclmi_icp_align call costs around its iterations (development helper): configs[3]'s clouds, one
iteration against thirty, RANSAC off and on, and the context creation alone."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from pcl_amd import capi
tgt, src_all, T0 = bench.icp_clouds()
for n in (1_000_000, 100_000):
    tree = capi.Tree(tgt[:n]); src = src_all[:n]
    capi.IcpShard(tree, src).end(want_aligned=False)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx = capi.IcpShard(tree, src); best = min(best, time.perf_counter() - t0); ctx.end(want_aligned=False)
    line = "n = %7d: context creation %.2f ms;" % (n, best * 1e3)
    for ransac in (0, 1000):
        for it in (1, 30):
            p = capi.icp_default_params()
            p.max_iterations, p.max_correspondence_distance, p.ransac_iterations, p.ransac_outlier_threshold = it, 25.0, ransac, 0.5
            b = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); capi.icp_align(tree, src, p); b = min(b, time.perf_counter() - t0)
            line += " align(ransac %d, %d it) %.2f ms;" % (ransac, it, b * 1e3)
    print(line)
    tree.close()

"""Tree build timing: device builder from device memory / from host memory, host builder (development helper).

usage: python tools/bench_tree_build.py [n ...]   (default 1e6 1e7)
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from pcl_amd import capi

sizes = [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 10_000_000]
for n in sizes:
    pts = bench.surface_cloud(n, 20260111) if n <= 2_000_000 else (np.random.default_rng(1).random((n, 3), dtype=np.float32) * 1000).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    torch.cuda.synchronize()
    best = {}
    for rep in range(3):
        t0 = time.time()
        t = capi.Tree.from_device(d.data_ptr(), n, 12)
        best["device builder, cloud in HBM"] = min(best.get("device builder, cloud in HBM", 1e9), time.time() - t0)
        t.close()
        t0 = time.time()
        t = capi.Tree(pts)
        best["device builder, cloud on the host"] = min(best.get("device builder, cloud on the host", 1e9), time.time() - t0)
        t.close()
    os.environ["PCLMI_TREE_BUILD"] = "host"
    t0 = time.time()
    t = capi.Tree(pts)
    best["host builder (tree_build.cpp) + upload"] = time.time() - t0
    t.close()
    os.environ.pop("PCLMI_TREE_BUILD")
    for k, v in best.items():
        print("n=%d  %-40s %8.2f ms" % (n, k, v * 1e3), flush=True)

"""Device-resident surface-normal timing: 1M-point surface-like cloud, normals of every point (development helper)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import ctypes as C
import bench
from pcl_amd import capi

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
pts = bench.surface_cloud(n, 20260111)
tree = capi.Tree(pts)
lib = capi.load_library()
dq = torch.from_numpy(pts).cuda()
out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
vp = (C.c_float * 3)(512.0, 512.0, 5000.0)
for k, radius in ((16, 0.0), (32, 0.0), (0, 12.0)):
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        rc = lib.pclmi_normals_dev(tree.h, dq.data_ptr(), n, 12, k, radius, vp, out.data_ptr(), None)
        torch.cuda.synchronize(); dt = time.time() - t0
        assert rc == 0, lib.pclmi_last_error()
    print("k=%d radius=%g: %.3f ms  %.3e normals/s" % (k, radius, dt * 1e3, n / dt), flush=True)

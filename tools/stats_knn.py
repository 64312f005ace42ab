"""Traversal statistics from the -DPCLMI_STATS build (development helper)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import numpy as np
from pcl_amd import capi
capi.load_library(str(capi.LIB_PATH).replace("libpclmi.so", "libpclmi_stats.so"))
lib = capi.load_library()
n = int(float(sys.argv[1])); nq = int(float(sys.argv[2])); k = int(sys.argv[3])
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
dq = torch.from_numpy(q).cuda()
di = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
out = (C.c_ulonglong * 16)()
lib.pclmi_debug_stats(out, 1)
tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
torch.cuda.synchronize()
lib.pclmi_debug_stats(out, 1)
waves = (nq + 63) // 64
names = ["inner", "bucket_pop", "bucket_scan", "cand_anypass", "pass_lane0", "compactions", "-", "-", "cyc_prepass", "cyc_bucket_load", "cyc_compact", "cyc_bucket_scan(incl compact)", "cyc_inner", "cyc_total_traversal"]
for i, nm in enumerate(names):
    print("%-32s %12d  per wave %.1f" % (nm, out[i], out[i] / waves))

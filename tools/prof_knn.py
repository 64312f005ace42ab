"""A few device-resident kNN (or radius) calls for rocprofv3 (development helper).
usage: prof_knn.py n_target n_queries k reps [radius]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi
if os.environ.get("PCLMI_LIB"):
    capi.load_library(os.environ["PCLMI_LIB"])

n = int(float(sys.argv[1])); nq = int(float(sys.argv[2])); k = int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
dq = torch.from_numpy(q).cuda()
di = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
for it in range(reps):
    tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
torch.cuda.synchronize()
print("done")

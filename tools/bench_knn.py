"""Quick device-resident kNN timing (development helper, not the contract bench)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import numpy as np
import torch
from pcl_amd import capi
if os.environ.get('PCLMI_LIB'):
    capi.load_library(os.environ['PCLMI_LIB'])

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16]
rng = np.random.default_rng(20260101)
pts = (rng.random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
rng = np.random.default_rng(20260102)
q = (rng.random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
t0 = time.time(); tree = capi.Tree(pts); print("build %.3fs" % (time.time() - t0), flush=True)
dq = torch.from_numpy(q).cuda()
for k in ks:
    di = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t0
        print("k=%d nq=%d: %.3f ms  %.3e q/s" % (k, nq, dt * 1e3, nq / dt), flush=True)

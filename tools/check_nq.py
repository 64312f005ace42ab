import sys, time
import numpy as np, torch
from pcl_amd import capi
from oracle import pcl_oracle as orc
n = 1_000_000
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
ot = orc.Tree(pts)
for nq in [int(float(v)) for v in sys.argv[1:]]:
    q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
    dq = torch.from_numpy(q).cuda(); k = 16
    di = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t0
    sel = np.random.default_rng(1).integers(0, nq, 3000)
    oi, od = ot.knn(q[sel], k, nthreads=8)
    gi = di.cpu().numpy()[sel]; gd = dd.cpu().numpy()[sel]
    print("nq=%d %.3f ms %.3e q/s  parity=%s" % (nq, dt*1e3, nq/dt, np.array_equal(gi, oi) and np.array_equal(gd, od)), flush=True)

"""k > 64: the tiled K = 64 search against the per-lane heap kernel (PCLMI_BIGK=heap), 1M-point target (development)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pcl_amd import capi
pts = (np.random.default_rng(20260101).random((1_000_000, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
for nq in [int(x) for x in os.environ.get("NQS", "1200,100000,1000000").split(",")]:
    q = (np.random.default_rng(3).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
    dq = torch.from_numpy(q).cuda()
    for k in (64, 128, 256, 512):
        if nq * k > 300_000_000:
            continue
        di = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("%s nq=%d k=%d: %.3f ms  %.3e q/s  %.1f ps/neighbour" % (os.environ.get("PCLMI_BIGK", "tiled"), nq, k, ms, nq / ms * 1e3, ms * 1e9 / (nq * k)), flush=True)

"""Kernel time of the per-lane and the union k-NN kernel around the sparse/dense crossover (development helper).
usage: PCLMI_KNN=lane|union crossover_time.py [k]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1_000_000
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
for nq in (50_000, 100_000, 150_000, 200_000, 250_000, 350_000):
    q = (np.random.default_rng(7).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
    dq = torch.from_numpy(q).cuda()
    di = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for _ in range(5):
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
    torch.cuda.synchronize()
    capi.timing_enable(True)
    capi.timing_read(capi.TIMER_KNN, reset=True)
    capi.timing_read(capi.TIMER_ORDER, reset=True)
    for _ in range(30):
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
    torch.cuda.synchronize()
    ms, cnt = capi.timing_read(capi.TIMER_KNN, reset=True)
    mo, co = capi.timing_read(capi.TIMER_ORDER, reset=True)
    capi.timing_enable(False)
    print("PCLMI_KNN=%s k=%d nq=%d: kernel %.1f us, order %.1f us" % (os.environ.get("PCLMI_KNN", "-"), k, nq, 1e3 * ms / max(cnt, 1), 1e3 * mo / max(co, 1)), flush=True)

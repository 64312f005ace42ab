"""Ordering-pass and kernel time of device-resident k-NN calls (library HIP events); development helper.
usage: [PCLMI_TILE=..] order_time.py [k]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = nq = 1_000_000
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
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
print("PCLMI_TILE=%s: order %.4f ms, kernel %.4f ms" % (os.environ.get("PCLMI_TILE", "-"), mo / max(co, 1), ms / max(cnt, 1)))

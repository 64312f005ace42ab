"""Per-wave traversal statistics of the k-NN kernel from the instrumented library variant
(make -C pcl_amd/csrc variant NAME=stats EXTRA=-DPCLMI_STATS).  usage: knn_stats.py [k]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = capi.load_library(os.environ.get("PCLMI_LIB") or os.path.join(os.path.dirname(capi.LIB_PATH), "libpclmi_stats.so"))
n = nq = 1_000_000
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
dq = torch.from_numpy(q).cuda()
di = torch.empty((nq, k), dtype=torch.int32, device="cuda")
dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
out = (C.c_ulonglong * 16)()
lib.pclmi_debug_knn_stats(out, 1)
tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
torch.cuda.synchronize()
lib.pclmi_debug_knn_stats(out, 0)
w = max(out[6], 1)
print("waves %d; per wave: leaves %.1f, candidates some lane appended %.1f, chain executions %.1f (of %.1f candidates = %.0f%%), compactions %.2f, "
      "inner expansions %.1f" % (out[6], out[0] / w, out[1] / w, out[2] / w, 16 * out[0] / w,
                                 100 * out[2] / max(16 * out[0], 1), out[3] / w, out[4] / w))
print("buckets popped %.1f per wave; clock64 ticks per wave: pre-pass %.0f, walk %.0f, epilogue %.0f" %
      (out[5] / w, out[8] / w, out[9] / w, out[10] / w))

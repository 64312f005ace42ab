"""A few device-resident radius searches (bench.py's configs[2] workload: r for ~32 neighbours, sorted CSR output)
for rocprofv3 (development helper).  usage: prof_radius.py [n_target] [n_queries] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
r = float((32 * 3 / (4 * np.pi * n / 1024.0 ** 3)) ** (1 / 3))
tree = capi.Tree(pts)
lib = capi.load_library()
dq = torch.from_numpy(q).cuda()
off = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
pi, pd, tot = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
for _ in range(reps):
    rc = lib.pclmi_radius_dev(tree.h, dq.data_ptr(), nq, 12, r, 0, 1, off.data_ptr(), C.byref(pi), C.byref(pd),
                              C.byref(tot), None)
    assert rc == 0, lib.pclmi_last_error()
    lib.pclmi_free_dev(pi)
    lib.pclmi_free_dev(pd)
torch.cuda.synchronize()
print("neighbours per query", tot.value / nq)

"""Device-resident radius search timing at BASELINE config C3 (development helper)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi
import ctypes as C

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
mean_m = float(sys.argv[3]) if len(sys.argv) > 3 else 32
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
r = (mean_m * 3 / (4 * np.pi * n / 1024.0 ** 3)) ** (1 / 3)
tree = capi.Tree(pts)
lib = capi.load_library()
dq = torch.from_numpy(q).cuda()
off = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
capi.timing_enable(True)
for srt in (1, 0):
    for it in range(4):
        pi, pd, tot = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        torch.cuda.synchronize(); t0 = time.time()
        rc = lib.pclmi_radius_dev(tree.h, dq.data_ptr(), nq, 12, float(r), 0, srt, off.data_ptr(), C.byref(pi), C.byref(pd), C.byref(tot), None)
        torch.cuda.synchronize(); dt = time.time() - t0
        assert rc == 0, lib.pclmi_last_error()
        lib.pclmi_free_dev(pi); lib.pclmi_free_dev(pd)
        ms, nl = capi.timing_read(capi.TIMER_RADIUS)
        print("sorted=%d r=%.2f total=%d (%.1f/query): %.3f ms  %.3e q/s   traversal passes %.3f ms" % (srt, r, tot.value, tot.value / nq, dt * 1e3, nq / dt, ms), flush=True)

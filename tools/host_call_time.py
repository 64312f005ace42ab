"""Whole-call latency of the host-pointer entry points on small batches (development helper)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pcl_amd import capi
n = 1_000_000
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
for nq in (1, 64, 1024, 16384):
    q = (np.random.default_rng(7).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
    for _ in range(5): tree.radius(q, 20.168)
    t0 = time.perf_counter()
    for _ in range(50): off, idx, d2 = tree.radius(q, 20.168)
    dt = (time.perf_counter() - t0) / 50
    print("radius host call nq=%d: %.1f us, mean neighbours %.1f" % (nq, dt * 1e6, off[-1] / nq), flush=True)
    for _ in range(5): tree.knn(q, 16)
    t0 = time.perf_counter()
    for _ in range(50): tree.knn(q, 16)
    dt = (time.perf_counter() - t0) / 50
    print("knn16 host call nq=%d: %.1f us" % (nq, dt * 1e6), flush=True)

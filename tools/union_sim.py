"""numpy / scipy model of the union walk's cost driver: how many buckets the UNION of a wavefront's 64 search balls
touches under a given query order (development tool; nothing here is product or oracle code).  It predicted the
effect of the tube order ("Query ordering") and rejected leaf-aligned orders before any kernel changed:
  uniform 1M x 1M, k = 16: Hilbert 18-bit cells 47.4 buckets per wavefront, queries sorted by home bucket 55.1, perfect
  cubes of 64 queries 36.9, tube order 37.3 (side x 0.8: 39.0, x 1.25: 39.3, x 0.5 / x 2: 53); one query's own ball: 7.9.
usage: union_sim.py [k] [uniform|surface]"""
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
n = nq = 1_000_000
if kind == "uniform":
    pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
    q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
else:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    pts = bench.surface_cloud(n, 7)
    q = pts.copy()
L = 16
buckets = []


def build(idx):  # bucket kd-tree: widest extent, balanced by bucket count; DFS numbering = left-to-right bucket order
    m = (len(idx) + L - 1) // L
    if m <= 1:
        buckets.append(idx)
        return
    p = pts[idx]
    d = int(np.argmax(p.max(0) - p.min(0)))
    nl = min(((m + 1) // 2) * L, len(idx) - 1)
    part = np.argpartition(p[:, d], nl - 1)
    build(idx[part[:nl]])
    build(idx[part[nl:]])


sys.setrecursionlimit(10000)
t0 = time.time()
build(np.arange(n))
blo = np.array([pts[b].min(0) for b in buckets])
bhi = np.array([pts[b].max(0) for b in buckets])
bucket_of = np.empty(n, np.int64)
for i, b in enumerate(buckets):
    bucket_of[b] = i
lo, hi = pts.min(0), pts.max(0)
ext = hi - lo
print("%d buckets, fill %.3f (%.1f s)" % (len(buckets), (bhi - blo).prod(1).sum() / ext.prod(), time.time() - t0), flush=True)
dk, ik = cKDTree(pts).query(q, k=K, workers=8)
tau = dk[:, -1].astype(np.float64) ** 2  # the exact k-th distance: a lower bound of what the walk's bound ever is
home = bucket_of[ik[:, 0]]


def hilbert(ix, iy, iz, bits=10):  # Skilling's axes-to-transpose, as knn_kernels.hip curve_index
    X = [ix.astype(np.uint32).copy(), iy.astype(np.uint32).copy(), iz.astype(np.uint32).copy()]
    Q = np.uint32(1 << (bits - 1))
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            X[0] = np.where(m, X[0], X[0] ^ t)
            X[i] = np.where(m, X[i], X[i] ^ t)
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = np.uint32(1 << (bits - 1))
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t

    def spread(v):
        v = v.astype(np.uint64)
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return r
    return (spread(X[0]) << np.uint64(2)) | (spread(X[1]) << np.uint64(1)) | spread(X[2])


btree = cKDTree((blo + bhi) / 2)
bext = np.linalg.norm((bhi - blo) / 2, axis=1).max()


def union_count(order, nw=300, seed=1):
    rng = np.random.default_rng(seed)
    tot = own = 0
    for w in rng.integers(0, nq // 64, nw):
        ids = order[w * 64:(w + 1) * 64]
        qq, tt = q[ids].astype(np.float64), tau[ids]
        c = (qq.min(0) + qq.max(0)) / 2
        cand = np.array(btree.query_ball_point(c, np.linalg.norm(qq - c, axis=1).max() + np.sqrt(tt.max()) + bext))
        d = np.maximum(np.maximum(blo[cand][None] - qq[:, None, :], qq[:, None, :] - bhi[cand][None]), 0)
        need = (d ** 2).sum(2) <= tt[:, None]
        tot += need.any(0).sum()
        own += need.sum(1).mean()
    return tot / nw, own / nw


cell = np.clip(((q - lo) / ext * 1024).astype(np.int64), 0, 1023)
h = hilbert(cell[:, 0], cell[:, 1], cell[:, 2])
for bits in (18, 21, 24):
    print("hilbert, %d-bit cells: %.1f buckets per wavefront (one query's own: %.1f)" %
          ((bits,) + union_count(np.argsort(h >> np.uint64(30 - bits), kind="stable"))), flush=True)
print("sorted by home bucket: %.1f" % union_count(np.argsort(home, kind="stable"))[0], flush=True)
side = (64.0 * ext.prod() / nq) ** (1.0 / 3.0)
for scale in (0.5, 0.8, 1.0, 1.25, 2.0):
    s = side * scale
    gy, gz = max(1, int(round(ext[1] / s))), max(1, int(round(ext[2] / s)))
    ty = np.clip(((q[:, 1] - lo[1]) / ext[1] * gy).astype(np.int64), 0, gy - 1)
    tz = np.clip(((q[:, 2] - lo[2]) / ext[2] * gz).astype(np.int64), 0, gz - 1)
    tube = tz * gy + np.where(tz & 1, gy - 1 - ty, ty)
    xq = np.clip(((q[:, 0] - lo[0]) / ext[0] * 1024).astype(np.int64), 0, 1023)
    key = tube * 1024 + np.where(tube & 1, 1023 - xq, xq)
    print("tube order, side x %.2f (%d x %d tubes): %.1f" % (scale, gy, gz, union_count(np.argsort(key, kind="stable"))[0]),
          flush=True)
